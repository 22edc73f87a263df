// Checks the addressing of global_load_lds_dwordx4 (gfx950) as render_bwd.hip uses it: lane l of the wave, if active, writes
// its 16 bytes at  LDS base + 16 l;  inactive lanes write nothing.  hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* __restrict__ src, const int* __restrict__ ids, float4* out) {
    __shared__ float4 buf[2][64];
    const int lane = threadIdx.x;
    buf[0][lane] = make_float4(-1, -1, -1, -1);
    buf[1][lane] = make_float4(-2, -2, -2, -2);
    __syncthreads();
    const int id = ids[lane];
    if (lane < 32 && (lane % 3) != 1) {
        __builtin_amdgcn_global_load_lds(src + id, &buf[0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + id + 1, &buf[1][0], 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xF70);  // vmcnt(0)
    __syncthreads();
    out[lane] = buf[0][lane];
    out[64 + lane] = buf[1][lane];
}
int main() {
    float4 h[512]; int ids[64];
    for (int i = 0; i < 512; i++) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    for (int l = 0; l < 64; l++) ids[l] = (l * 37) % 400;
    float4 *d, *o; int* di;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 128 * sizeof(float4)); hipMalloc(&di, sizeof ids);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(di, ids, sizeof ids, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, di, o);
    float4 r[128]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const bool act = l < 32 && (l % 3) != 1;
        const float e0 = act ? (float)ids[l] : -1.f, e1 = act ? (float)(ids[l] + 1) : -2.f;
        if (r[l].x != e0 || r[64 + l].x != e1 || (act && r[l].w != e0 + 0.75f)) { bad++; if (bad < 6) printf("lane %d: got %g %g want %g %g\n", l, r[l].x, r[64 + l].x, e0, e1); }
    }
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "OK: lane l writes base + 16 l, inactive lanes nothing", bad);
    return bad != 0;
}
