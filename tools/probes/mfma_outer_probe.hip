// Probe of v_mfma_f32_32x32x1_2b_f32 on gfx950: which lanes' A and B meet in each (lane, register) of the result, checked against
// the layout render_fwd.hip (OUTER) assumes:  block b = register / 16;  A comes from lane 32 b + i, B from lane 32 b + j with
// j = lane % 32 and i = 8 (r / 4) + 4 (lane / 32) + r % 4, r = register % 16.   hipcc --offload-arch=gfx950 -o probe ... && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x32 acc;
    for (int r = 0; r < 32; r++) acc[r] = 0.f;
    // A = 1 + lane, B = 128 (1 + lane): D = A B = (1 + la) * 128 * (1 + lb): both factors below 2^7 -- exact, and decodable
    acc = __builtin_amdgcn_mfma_f32_32x32x1f32((float)(1 + l), 128.f * (float)(1 + l), acc, 0, 0, 0);
    for (int r = 0; r < 32; r++) out[l * 32 + r] = acc[r];
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 32 * 4);
    probe<<<1, 64>>>(d);
    static float h[64 * 32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 32; r++) {
            const int v = (int)h[l * 32 + r];
            const int lb = v / 128 % 65536;  // v = (1 + la) * 128 * (1 + lb): try all la
            int fa = -1, fb = -1;
            for (int a = 0; a < 64 && fa < 0; a++)
                for (int b = 0; b < 64; b++)
                    if ((1 + a) * 128 * (1 + b) == v && (fa < 0)) {
                        // ambiguous factorisations exist: prefer the assumed one if it matches
                        const int blk = r / 16, rr = r % 16;
                        const int ia = 32 * blk + 8 * (rr / 4) + 4 * (l / 32) + rr % 4, ib = 32 * blk + l % 32;
                        if ((1 + ia) * 128 * (1 + ib) == v) { fa = ia; fb = ib; }
                        else { fa = a; fb = b; }
                        break;
                    }
            (void)lb;
            const int blk = r / 16, rr = r % 16;
            const int ia = 32 * blk + 8 * (rr / 4) + 4 * (l / 32) + rr % 4, ib = 32 * blk + l % 32;
            if ((1 + ia) * 128 * (1 + ib) != v) {
                if (bad < 20) printf("lane %2d reg %2d: got %d = A[%d] * B[%d], assumed A[%d] * B[%d]\n", l, r, v, fa, fb, ia, ib);
                bad++;
            }
        }
    printf(bad ? "LAYOUT MISMATCH in %d places\n" : "layout as assumed (0 mismatches)\n", bad);
    for (int l = 0; l < 64; l += 21) {
        printf("lane %2d:", l);
        for (int r = 0; r < 32; r += 5) printf(" r%d=%d", r, (int)h[l * 32 + r]);
        printf("\n");
    }
    return bad != 0;
}
