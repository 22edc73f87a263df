// Fused multi-tensor Adam step for the Gaussian parameter groups.
//
// Reference: scene/gaussian_model.py:163-253 builds torch.optim.Adam(lr=0.0, eps=1e-15) over up to
// seven named groups (xyz, f_dc, f_rest, semantics, opacity, scaling, rotation) and train.py:193
// calls optimizer.step(); gui/main.py:480-513 (clear_noralative_gs_grad) zeroes the gradient rows of
// masked Gaussians before the step.  torch runs that as ~10 foreach kernels per state tensor; here
// ONE launch walks every group: 16 bytes/element in (param, grad, exp_avg, exp_avg_sq), 12 out --
// the HBM floor of the update -- with the optional per-Gaussian mask applied on the fly.
//
// Arithmetic follows torch's (non-capturable, foreach) Adam on the GPU, op for op in fp32:
//     m   = m + (g - m) * (1 - beta1)                 _foreach_lerp_
//     v   = v * beta2 + ((1 - beta2) * g) * g         _foreach_mul_, _foreach_addcmul_
//     den = sqrt(v) / sqrt(1 - beta2^t) + eps         _foreach_sqrt, _foreach_div_, _foreach_add_
//     p   = p + (-lr / (1 - beta1^t)) * (m / den)     _foreach_addcdiv_
// (scalars are formed in double on the host and rounded to fp32 once, as torch does).  This TU is
// compiled with -ffp-contract=off so that the spelling is the arithmetic.
#include "common.h"

namespace goi {

namespace {

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_VEC = 4;
constexpr int ADAM_BLOCK_ELEMS = ADAM_THREADS * ADAM_VEC;

struct AdamTable {
    GoiAdamGroup g[GOI_ADAM_MAX_GROUPS];
    unsigned int block_end[GOI_ADAM_MAX_GROUPS];  // exclusive prefix of blocks per group
    int n;
    float one_minus_beta1, beta2, one_minus_beta2, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size_neg, float bc2_sqrt,
                                         const AdamTable& t) {
    m = m + (g - m) * t.one_minus_beta1;
    v = v * t.beta2 + (t.one_minus_beta2 * g) * g;
    const float den = sqrtf(v) / bc2_sqrt + t.eps;
    p = p + step_size_neg * (m / den);
}

// skip_flag (device word, may be NULL): a non-zero value makes the whole launch a no-op -- parameters and moments stay
// as they are.  The rasterizer sets such a word for a view whose speculative forward was truncated (goi_raster_truncated_flag):
// that view is skipped on the device, with no host round trip.
__global__ __launch_bounds__(ADAM_THREADS) void adam_step_k(const AdamTable t, const uint8_t* __restrict__ nograd_mask,
                                                            const uint32_t* __restrict__ skip_flag) {
    if (skip_flag && *skip_flag) return;
    int gi = 0;
#pragma unroll
    for (int i = 0; i < GOI_ADAM_MAX_GROUPS - 1; i++)
        if (i < t.n - 1 && blockIdx.x >= t.block_end[i]) gi = i + 1;
    const GoiAdamGroup grp = t.g[gi];
    const unsigned int b0 = gi ? t.block_end[gi - 1] : 0u;
    const long long e0 = ((long long)(blockIdx.x - b0) * ADAM_THREADS + threadIdx.x) * ADAM_VEC;
    if (e0 >= grp.numel) return;
    const float ssn = -grp.step_size;
    if (e0 + ADAM_VEC <= grp.numel) {
        float4 p = *reinterpret_cast<const float4*>(grp.param + e0);
        float4 g = *reinterpret_cast<const float4*>(grp.grad + e0);
        float4 m = *reinterpret_cast<const float4*>(grp.exp_avg + e0);
        float4 v = *reinterpret_cast<const float4*>(grp.exp_avg_sq + e0);
        if (nograd_mask) {
            if (nograd_mask[(e0 + 0) / grp.row_len]) g.x = 0.f;
            if (nograd_mask[(e0 + 1) / grp.row_len]) g.y = 0.f;
            if (nograd_mask[(e0 + 2) / grp.row_len]) g.z = 0.f;
            if (nograd_mask[(e0 + 3) / grp.row_len]) g.w = 0.f;
        }
        adam_one(p.x, g.x, m.x, v.x, ssn, grp.bc2_sqrt, t);
        adam_one(p.y, g.y, m.y, v.y, ssn, grp.bc2_sqrt, t);
        adam_one(p.z, g.z, m.z, v.z, ssn, grp.bc2_sqrt, t);
        adam_one(p.w, g.w, m.w, v.w, ssn, grp.bc2_sqrt, t);
        *reinterpret_cast<float4*>(grp.param + e0) = p;
        *reinterpret_cast<float4*>(grp.exp_avg + e0) = m;
        *reinterpret_cast<float4*>(grp.exp_avg_sq + e0) = v;
    } else {
        for (long long e = e0; e < grp.numel; e++) {
            float p = grp.param[e], g = grp.grad[e], m = grp.exp_avg[e], v = grp.exp_avg_sq[e];
            if (nograd_mask && nograd_mask[e / grp.row_len]) g = 0.f;
            adam_one(p, g, m, v, ssn, grp.bc2_sqrt, t);
            grp.param[e] = p;
            grp.exp_avg[e] = m;
            grp.exp_avg_sq[e] = v;
        }
    }
}

}  // namespace

int launch_adam_step(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                     const uint8_t* nograd_mask, const uint32_t* skip_flag, hipStream_t s) {
    AdamTable t;
    unsigned int blocks = 0;
    t.n = 0;
    for (int i = 0; i < n_groups; i++) {
        if (groups[i].numel <= 0) continue;
        t.g[t.n] = groups[i];
        blocks += (unsigned int)((groups[i].numel + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS);
        t.block_end[t.n] = blocks;
        t.n++;
    }
    if (t.n == 0) return 0;
    for (int i = t.n; i < GOI_ADAM_MAX_GROUPS; i++) {
        t.g[i] = t.g[0];
        t.block_end[i] = blocks;
    }
    // the host forms 1 - beta in double and rounds once, like the Python scalars torch passes down
    t.one_minus_beta1 = (float)(1.0 - beta1);
    t.beta2 = (float)beta2;
    t.one_minus_beta2 = (float)(1.0 - beta2);
    t.eps = (float)eps;
    adam_step_k<<<dim3(blocks), dim3(ADAM_THREADS), 0, s>>>(t, nograd_mask, skip_flag);
    return 0;
}

}  // namespace goi
