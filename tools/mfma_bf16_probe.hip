// Checks the operand / result layout assumed for v_mfma_f32_16x16x32_bf16 on gfx950 (tools only):
//   A: lane l holds A[row = l % 16][k = 8 (l / 16) + i], i = 0..7
//   B: lane l holds B[k = 8 (l / 16) + i][col = l % 16]
//   D: lane l holds D[row = 4 (l / 16) + r][col = l % 16], r = 0..3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, kq = l >> 4, mm = l & 15;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) {
        a[i] = (__bf16)A[mm * 32 + 8 * kq + i];
        b[i] = (__bf16)B[(8 * kq + i) * 16 + mm];
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * kq + r) * 16 + mm] = acc[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 512; i++) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { float s = 0; for (int k = 0; k < 32; k++) s += hA[r * 32 + k] * hB[k * 16 + c]; ref[r * 16 + c] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("mfma_f32_16x16x32_bf16 layout check: %d mismatches of 256\n", bad);
    return bad != 0;
}
