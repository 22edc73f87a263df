// Probe of the v_mfma_f32_4x4x1_16b_f32 operand / result layout on gfx950 (tools only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    // A = 1000 + lane, B = lane: D = A*B reveals which lanes' A and B meet in each (lane, reg)
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1000 + l), (float)(l), acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    probe<<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; l++) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; r++) {
            // factor: find (a_lane, b_lane) with (1000+a)*b == h
            int fa = -1, fb = -1;
            for (int a = 0; a < 64 && fa < 0; a++)
                for (int b = 0; b < 64; b++)
                    if ((float)(1000 + a) * (float)b == h[l * 4 + r] && (b != 0 || h[l*4+r]==0)) { fa = a; fb = b; break; }
            printf("  r%d=A[%d]*B[%d]", r, fa, fb);
        }
        printf("\n");
    }
    return 0;
}
