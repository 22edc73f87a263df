// Throughput of VALU work mixed with fp32 or bf16 MFMAs at 4 waves/SIMD (tools only): does the MFMA time
// add to the VALU time of the co-resident waves, or hide behind it?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int NF, int NB>
__global__ __launch_bounds__(64, 4) void mix(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    const float fa = threadIdx.x, fb = 1.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
#pragma unroll
        for (int k = 0; k < NF; k++) {
            if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s + acc0[0] + acc1[1];
}
template <int NV, int NF, int NB>
void run(float* d, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 16, iters = 2000;
    mix<NV, NF, NB><<<blocks, 64>>>(d, iters);
    hipEventRecord(e0);
    mix<NV, NF, NB><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one iteration of all 4 waves of a SIMD, in clocks at 2.4 GHz
    printf("%-28s %8.3f ms  = %7.1f clk per iteration per SIMD (4 waves)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 16 * 64 * 4);
    run<48, 0, 0>(d, "48 VALU");
    run<0, 4, 0>(d, "4 f32 MFMA");
    run<48, 4, 0>(d, "48 VALU + 4 f32 MFMA");
    run<0, 0, 4>(d, "4 bf16 MFMA");
    run<48, 0, 4>(d, "48 VALU + 4 bf16 MFMA");
    run<48, 0, 2>(d, "48 VALU + 2 bf16 MFMA");
    run<96, 4, 0>(d, "96 VALU + 4 f32 MFMA");
    run<96, 0, 0>(d, "96 VALU");
    return 0;
}
