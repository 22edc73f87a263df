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
// wave-specialised: even workgroups (single waves) only VALU, odd ones only bf16 MFMA -- do the two pipes overlap ACROSS waves?
template <int NV, int NB>
__global__ __launch_bounds__(64, 4) void split(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    if (blockIdx.x & 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < NB; k += 4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < NV; k++) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s + acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
template <int NV, int NB>
void run_split(float* d, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 16, iters = 2000;
    split<NV, NB><<<blocks, 64>>>(d, iters);
    hipEventRecord(e0);
    split<NV, NB><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  = %7.1f clk per iteration per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}
// transcendental throughput: NE v_exp_f32 per iteration on 8 independent chains, 4 waves per SIMD
template <int NE, int OP>
__global__ __launch_bounds__(64, 4) void trans(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i * 0.1f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NE; k++) {
            if (OP == 0) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
            else if (OP == 1) v[k & 7] = __builtin_amdgcn_rcpf(v[k & 7]);
            else v[k & 7] = __builtin_amdgcn_logf(v[k & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NE, int OP>
void run_trans(float* d, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 16, iters = 2000;
    trans<NE, OP><<<blocks, 64>>>(d, iters);
    hipEventRecord(e0);
    trans<NE, OP><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  = %7.1f clk per instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / iters / (4.0 * NE));
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
    run_trans<48, 0>(d, "48 v_exp_f32 x 4 waves");
    run_trans<48, 1>(d, "48 v_rcp_f32 x 4 waves");
    run_trans<48, 2>(d, "48 v_log_f32 x 4 waves");
    run_split<96, 0>(d, "split: 2 waves x 96 VALU, 2 waves idle");
    run_split<0, 8>(d, "split: 2 waves idle, 2 waves x 8 bf16 MFMA");
    run_split<96, 8>(d, "split: 2 waves x 96 VALU | 2 waves x 8 bf16 MFMA");
    run_split<96, 16>(d, "split: 2 waves x 96 VALU | 2 waves x 16 bf16 MFMA");
    run_split<192, 16>(d, "split: 2 waves x 192 VALU | 2 waves x 16 bf16 MFMA");
    return 0;
}
