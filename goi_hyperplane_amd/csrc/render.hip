// Tile blend kernels for gfx950: forward compositing of RGB + S-dim semantic feature + depth +
// alpha, its backward, and trace.  One 256-thread workgroup (4 waves) per 16x16 tile, one pixel
// per lane; each WAVE owns an 8x8 pixel quadrant.
//
// Behaviour follows the reference's renderCUDA (forward: cuda_rasterizer/forward.cu:261-386,
// backward: cuda_rasterizer/backward.cu:415-625, trace: forward.cu:422-551): same per-pixel
// front-to-back order, same guards (power > 0, alpha < 1/255, T(1-alpha) < 1e-4, 0.99 clamp),
// same outputs.  The execution design is this library's own, for CDNA4:
//   * a batch of Gaussians is staged ONCE per workgroup into LDS including its feature row
//     (rgb, depth, semantics), so the inner loop only issues broadcast LDS reads;
//   * at staging time every Gaussian is tested against each wave's 8x8 quadrant with its exact
//     contribution box (GaussRec hx/hy); the four 64-bit hit masks go to LDS and each wave walks
//     only the set bits of its own masks (scalar bit scan), so Gaussians that provably cannot
//     reach alpha >= 1/255 in a quadrant cost that wave nothing;
//   * early termination is per wave (64-bit ballot) with one workgroup vote per batch;
//   * backward: the per-channel "accum_rec" recurrences collapse to ONE scalar recurrence on the
//     dot product <feature, dL/dpixel> (same mathematics), per-Gaussian partial
//     gradients are reduced across the 64 lanes with DPP, across the 4 waves in LDS, and leave
//     the workgroup as one global atomic per (tile, Gaussian, quantity) instead of one per
//     (pixel, Gaussian, quantity).
#include "common.h"

namespace goi {

namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 0.0001f;
constexpr float kAlphaMax = 0.99f;

// alpha evaluation shared by forward, backward and trace so that all three agree on which
// (pixel, Gaussian) pairs contribute.
struct PairEval {
    float dx, dy, power, G, alpha;
    bool hit;
};
__device__ __forceinline__ PairEval eval_pair(float gx_, float gy_, float ca, float cb, float cc, float o, float pxf,
                                              float pyf) {
    PairEval e;
    e.dx = gx_ - pxf;
    e.dy = gy_ - pyf;
    e.power = -0.5f * (ca * e.dx * e.dx + cc * e.dy * e.dy) - cb * e.dx * e.dy;
    e.G = __expf(e.power);
    e.alpha = fminf(kAlphaMax, o * e.G);
    e.hit = (e.power <= 0.0f) && (e.alpha >= kAlphaMin);
    return e;
}

__device__ __forceinline__ bool box_hits_quadrant(float x, float y, float hx, float hy, float X0, float Y0) {
    return (hx >= 0.f) && (x - hx <= X0 + 7.f) && (x + hx >= X0) && (y - hy <= Y0 + 7.f) && (y + hy >= Y0);
}

// 64-lane sum, result valid in lane 63 (DPP row ops + row broadcasts; gfx9 wave64 idiom).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    int x;
#define GOI_DPP_ADD(ctrl, rmask)                                                                  \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false);               \
    v += __int_as_float(x);
    GOI_DPP_ADD(0x111, 0xF)  // row_shr:1
    GOI_DPP_ADD(0x112, 0xF)  // row_shr:2
    GOI_DPP_ADD(0x114, 0xF)  // row_shr:4
    GOI_DPP_ADD(0x118, 0xF)  // row_shr:8   -> lane 15 of each row holds the row sum
    GOI_DPP_ADD(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    GOI_DPP_ADD(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef GOI_DPP_ADD
    return v;
}

template <int S4>
struct Cfg {
    static constexpr int NF4 = 1 + S4;        // float4 words of staged features: (r,g,b,depth) + semantics
    static constexpr int NSEM = 4 * S4;       // padded semantic channels
    static constexpr int NQ = 10 + 4 * S4;    // backward partial sums per Gaussian
};

struct TileGeom {
    int tile, tx, ty, w, lane, px, py;
    bool inside;
    float pxf, pyf;
};
__device__ __forceinline__ TileGeom tile_geom(int W, int H, int gx) {
    TileGeom t;
    t.tile = blockIdx.x;
    t.tx = t.tile % gx;
    t.ty = t.tile / gx;
    t.w = threadIdx.x >> 6;
    t.lane = threadIdx.x & 63;
    t.px = t.tx * TILE + (t.w & 1) * 8 + (t.lane & 7);
    t.py = t.ty * TILE + (t.w >> 1) * 8 + (t.lane >> 3);
    t.inside = t.px < W && t.py < H;
    t.pxf = (float)t.px;
    t.pyf = (float)t.py;
    return t;
}

// ------------------------------------------------------------------------------------------------
// forward (TRACE = false) and trace (TRACE = true)
// ------------------------------------------------------------------------------------------------
template <int S4, bool TRACE>
__global__ __launch_bounds__(256) void render_fwd_k(const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list, int W, int H, int gx, int S,
                                                    const GaussRec* __restrict__ rec, const float* __restrict__ semantics,
                                                    const float* __restrict__ bg, float* __restrict__ out_color,
                                                    float* __restrict__ out_sem, float* __restrict__ out_depth,
                                                    float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
                                                    const float* __restrict__ img_sem, float* __restrict__ gau_sem,
                                                    int* __restrict__ num_gsem) {
    constexpr int NF4 = TRACE ? 1 : Cfg<S4>::NF4;
    constexpr int NSEM = TRACE ? 0 : Cfg<S4>::NSEM;
    __shared__ float4 s_geo[256];
    __shared__ float2 s_geo2[256];
    __shared__ float4 s_feat[256 * NF4];
    __shared__ unsigned long long s_mask[4][4];
    __shared__ uint32_t s_id[TRACE ? 256 : 1];

    const TileGeom t = tile_geom(W, H, gx);
    const uint2 range = ranges[t.tile];
    const int len = (int)(range.y - range.x);
    const int rounds = (len + 255) / 256;
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    const float X0 = (float)(t.tx * TILE), Y0 = (float)(t.ty * TILE);
    const uint64_t lt = (1ull << t.lane) - 1ull;
    (void)lt;

    bool done = !t.inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C[4] = {0.f, 0.f, 0.f, 0.f};  // r, g, b, depth
    float Cs[NSEM > 0 ? NSEM : 1];
#pragma unroll
    for (int i = 0; i < NSEM; i++) Cs[i] = 0.f;

    for (int b = 0; b < rounds; b++) {
        const int wave_done = __all(done) ? 1 : 0;
        if (__syncthreads_and(wave_done)) break;

        // ---- stage one batch: geometry, feature row, per-quadrant hit masks
        {
            const int k = b * 256 + (int)threadIdx.x;
            bool h0 = false, h1 = false, h2 = false, h3 = false;
            if (k < len) {
                const uint32_t id = point_list[range.x + k];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id);
                const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
                s_geo[threadIdx.x] = q0;
                s_geo2[threadIdx.x] = make_float2(q1.x, q1.y);
                s_feat[threadIdx.x * NF4] = make_float4(q1.w, q2.x, q2.y, q1.z);
                if constexpr (TRACE) {
                    s_id[threadIdx.x] = id;
                } else {
                    const float* srow = semantics + (size_t)id * S;
                    if ((S & 3) == 0) {
#pragma unroll
                        for (int i = 0; i < S4; i++)
                            s_feat[threadIdx.x * NF4 + 1 + i] = reinterpret_cast<const float4*>(srow)[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < S4; i++) {
                            float4 v;
                            v.x = (4 * i + 0 < S) ? srow[4 * i + 0] : 0.f;
                            v.y = (4 * i + 1 < S) ? srow[4 * i + 1] : 0.f;
                            v.z = (4 * i + 2 < S) ? srow[4 * i + 2] : 0.f;
                            v.w = (4 * i + 3 < S) ? srow[4 * i + 3] : 0.f;
                            s_feat[threadIdx.x * NF4 + 1 + i] = v;
                        }
                    }
                }
                h0 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0);
                h1 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0);
                h2 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0 + 8.f);
                h3 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0 + 8.f);
            }
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if (t.lane == 0) {
                s_mask[0][t.w] = m0;
                s_mask[1][t.w] = m1;
                s_mask[2][t.w] = m2;
                s_mask[3][t.w] = m3;
            }
        }
        __syncthreads();

        // ---- consume: this wave walks the set bits of its own quadrant's masks, front to back
        if (!wave_done) {
            for (int sw = 0; sw < 4; sw++) {
                unsigned long long m = s_mask[t.w][sw];
                m = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                    (unsigned int)__builtin_amdgcn_readfirstlane((int)(m & 0xFFFFFFFFull));
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const int gi = sw * 64 + j;
                    const float4 g = s_geo[gi];
                    const float2 g2 = s_geo2[gi];
                    const PairEval e = eval_pair(g.x, g.y, g.z, g.w, g2.x, g2.y, t.pxf, t.pyf);
                    bool c = !done && e.hit;
                    const float test_T = T * (1.f - e.alpha);
                    if (c && test_T < kTMin) {
                        done = true;
                        c = false;
                    }
                    if (__any(c)) {
                        const float wgt = c ? e.alpha * T : 0.f;
                        const float4 f0 = s_feat[gi * NF4];
                        C[0] += f0.x * wgt;
                        C[1] += f0.y * wgt;
                        C[2] += f0.z * wgt;
                        C[3] += f0.w * wgt;
                        if constexpr (!TRACE) {
#pragma unroll
                            for (int i = 0; i < S4; i++) {
                                const float4 f = s_feat[gi * NF4 + 1 + i];
                                Cs[4 * i + 0] += f.x * wgt;
                                Cs[4 * i + 1] += f.y * wgt;
                                Cs[4 * i + 2] += f.z * wgt;
                                Cs[4 * i + 3] += f.w * wgt;
                            }
                        } else {
                            if (c && (double)e.alpha > 0.005) {
                                const uint32_t id = s_id[gi];
                                for (int ch = 0; ch < S; ch++)
                                    atomicAdd(&gau_sem[(size_t)id * S + ch], img_sem[ch * HW + pix_id]);
                                atomicAdd(&num_gsem[id], S);
                            }
                        }
                        if (c) {
                            T = test_T;
                            last_contributor = (uint32_t)(b * 256 + gi + 1);
                        }
                    }
                    if (__all(done)) {
                        m = 0;
                        sw = 4;
                    }
                }
            }
        }
    }

    if (t.inside) {
        n_contrib[pix_id] = last_contributor;
        out_color[0 * HW + pix_id] = C[0] + T * bg[0];
        out_color[1 * HW + pix_id] = C[1] + T * bg[1];
        out_color[2 * HW + pix_id] = C[2] + T * bg[2];
        if constexpr (!TRACE) {
#pragma unroll
            for (int ch = 0; ch < NSEM; ch++)
                if (ch < S) out_sem[ch * HW + pix_id] = Cs[ch];
            out_alpha[pix_id] = 1.f - T;
            out_depth[pix_id] = C[3];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int BWD_BATCH = 128;  // Gaussians staged per round (2 staging waves)

template <int S4>
__global__ __launch_bounds__(256) void render_bwd_k(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int S,
    const GaussRec* __restrict__ rec, const float* __restrict__ semantics, const float* __restrict__ bg,
    const float* __restrict__ out_alpha, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
    const float* __restrict__ dL_dpixsem, const float* __restrict__ dL_dpixdepth, const float* __restrict__ dL_dalphas,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_dsemantic, float* __restrict__ dL_ddepths) {
    constexpr int NF4 = Cfg<S4>::NF4;
    constexpr int NSEM = Cfg<S4>::NSEM;
    constexpr int NQ = Cfg<S4>::NQ;
    __shared__ float4 s_geo[BWD_BATCH];
    __shared__ float2 s_geo2[BWD_BATCH];
    __shared__ float4 s_feat[BWD_BATCH * NF4];
    __shared__ uint32_t s_id[BWD_BATCH];
    __shared__ float s_acc[BWD_BATCH * NQ];
    __shared__ uint32_t s_touched[BWD_BATCH];
    __shared__ unsigned long long s_mask[4][BWD_BATCH / 64];

    const TileGeom t = tile_geom(W, H, gx);
    const uint2 range = ranges[t.tile];
    const int len = (int)(range.y - range.x);
    if (len == 0) return;
    const int rounds = (len + BWD_BATCH - 1) / BWD_BATCH;
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    const float X0 = (float)(t.tx * TILE), Y0 = (float)(t.ty * TILE);

    const float T_final = t.inside ? (1.f - out_alpha[pix_id]) : 0.f;
    float T = T_final;
    const int last_contributor = t.inside ? (int)n_contrib[pix_id] : 0;
    float dLc[4];  // dL/d(r,g,b,depth)
    float dLs[NSEM];
    float dLa = 0.f;
    if (t.inside) {
        dLc[0] = dL_dpix[0 * HW + pix_id];
        dLc[1] = dL_dpix[1 * HW + pix_id];
        dLc[2] = dL_dpix[2 * HW + pix_id];
        dLc[3] = dL_dpixdepth[pix_id];
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLs[ch] = ch < S ? dL_dpixsem[ch * HW + pix_id] : 0.f;
        dLa = dL_dalphas[pix_id];
    } else {
        dLc[0] = dLc[1] = dLc[2] = dLc[3] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLs[ch] = 0.f;
    }
    const float bg_dot = bg[0] * dLc[0] + bg[1] * dLc[1] + bg[2] * dLc[2];
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    float R = 0.f;  // <accumulated colour behind the current Gaussian, dL/dpixel> (scalar recurrence)

    for (int i = threadIdx.x; i < BWD_BATCH * NQ; i += 256) s_acc[i] = 0.f;
    if (threadIdx.x < BWD_BATCH) s_touched[threadIdx.x] = 0;

    // wave-level skip: nothing to do for this wave once every lane is past its last contributor
    for (int b = 0; b < rounds; b++) {
        __syncthreads();  // previous flush finished; s_acc is zero
        // ---- stage (back to front): slot j holds list position len-1-(b*BATCH+j)
        if (threadIdx.x < BWD_BATCH) {
            const int k = b * BWD_BATCH + (int)threadIdx.x;
            bool h0 = false, h1 = false, h2 = false, h3 = false;
            if (k < len) {
                const uint32_t id = point_list[range.y - 1 - k];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id);
                const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
                s_geo[threadIdx.x] = q0;
                s_geo2[threadIdx.x] = make_float2(q1.x, q1.y);
                s_feat[threadIdx.x * NF4] = make_float4(q1.w, q2.x, q2.y, q1.z);
                s_id[threadIdx.x] = id;
                const float* srow = semantics + (size_t)id * S;
                if ((S & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < S4; i++)
                        s_feat[threadIdx.x * NF4 + 1 + i] = reinterpret_cast<const float4*>(srow)[i];
                } else {
#pragma unroll
                    for (int i = 0; i < S4; i++) {
                        float4 v;
                        v.x = (4 * i + 0 < S) ? srow[4 * i + 0] : 0.f;
                        v.y = (4 * i + 1 < S) ? srow[4 * i + 1] : 0.f;
                        v.z = (4 * i + 2 < S) ? srow[4 * i + 2] : 0.f;
                        v.w = (4 * i + 3 < S) ? srow[4 * i + 3] : 0.f;
                        s_feat[threadIdx.x * NF4 + 1 + i] = v;
                    }
                }
                h0 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0);
                h1 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0);
                h2 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0 + 8.f);
                h3 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0 + 8.f);
            }
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if (t.lane == 0) {
                s_mask[0][t.w] = m0;
                s_mask[1][t.w] = m1;
                s_mask[2][t.w] = m2;
                s_mask[3][t.w] = m3;
            }
        }
        __syncthreads();

        // ---- consume
        for (int sw = 0; sw < BWD_BATCH / 64; sw++) {
            unsigned long long m = s_mask[t.w][sw];
            m = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                (unsigned int)__builtin_amdgcn_readfirstlane((int)(m & 0xFFFFFFFFull));
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int gi = sw * 64 + j;
                const int pos0 = len - 1 - (b * BWD_BATCH + gi);  // 0-based list position
                const float4 g = s_geo[gi];
                const float2 g2 = s_geo2[gi];
                const PairEval e = eval_pair(g.x, g.y, g.z, g.w, g2.x, g2.y, t.pxf, t.pyf);
                const bool c = (pos0 < last_contributor) && e.hit;
                if (!__any(c)) continue;

                const float one_m_a = 1.f - e.alpha;
                const float Tn = T / one_m_a;
                const float wgt = c ? e.alpha * Tn : 0.f;
                const float4 f0 = s_feat[gi * NF4];
                float dotv = f0.x * dLc[0] + f0.y * dLc[1] + f0.z * dLc[2] + f0.w * dLc[3] + dLa;
                float part[NQ];
                part[0] = wgt * dLc[0];
                part[1] = wgt * dLc[1];
                part[2] = wgt * dLc[2];
                part[3] = wgt * dLc[3];
#pragma unroll
                for (int i = 0; i < S4; i++) {
                    const float4 f = s_feat[gi * NF4 + 1 + i];
                    dotv += f.x * dLs[4 * i + 0] + f.y * dLs[4 * i + 1] + f.z * dLs[4 * i + 2] + f.w * dLs[4 * i + 3];
                    part[4 + 4 * i + 0] = wgt * dLs[4 * i + 0];
                    part[4 + 4 * i + 1] = wgt * dLs[4 * i + 1];
                    part[4 + 4 * i + 2] = wgt * dLs[4 * i + 2];
                    part[4 + 4 * i + 3] = wgt * dLs[4 * i + 3];
                }
                float dL_dopa = (dotv - R) * Tn + (-T_final / one_m_a) * bg_dot;
                if (c) {
                    R = e.alpha * dotv + one_m_a * R;
                    T = Tn;
                } else {
                    dL_dopa = 0.f;
                }
                const float Gm = c ? e.G : 0.f;  // keeps inf/NaN of non-contributing lanes out of the sums
                const float dL_dG = g2.y * dL_dopa;
                const float gdx = Gm * e.dx, gdy = Gm * e.dy;
                const float dG_ddelx = -gdx * g.z - gdy * g.w;
                const float dG_ddely = -gdy * g2.x - gdx * g.w;
                part[4 + NSEM + 0] = dL_dG * dG_ddelx * ddelx_dx;
                part[4 + NSEM + 1] = dL_dG * dG_ddely * ddely_dy;
                part[4 + NSEM + 2] = -0.5f * gdx * e.dx * dL_dG;
                part[4 + NSEM + 3] = -0.5f * gdx * e.dy * dL_dG;
                part[4 + NSEM + 4] = -0.5f * gdy * e.dy * dL_dG;
                part[4 + NSEM + 5] = Gm * dL_dopa;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float s = wave_sum_to_lane63(part[q]);
                    if (t.lane == 63) atomicAdd(&s_acc[gi * NQ + q], s);
                }
                if (t.lane == 63) s_touched[gi] = 1;
            }
        }
        __syncthreads();

        // ---- flush this batch: one global atomic per (Gaussian, quantity) that received anything
        for (int i = threadIdx.x; i < BWD_BATCH * NQ; i += 256) {
            const int gi = i / NQ, q = i - gi * NQ;
            if (s_touched[gi]) {
                const float v = s_acc[i];
                s_acc[i] = 0.f;
                const uint32_t id = s_id[gi];
                float* dst;
                if (q < 3)
                    dst = dL_dcolor + (size_t)id * 3 + q;
                else if (q == 3)
                    dst = dL_ddepths + id;
                else if (q < 4 + NSEM) {
                    const int ch = q - 4;
                    dst = (ch < S) ? dL_dsemantic + (size_t)id * S + ch : nullptr;
                } else if (q < 4 + NSEM + 2)
                    dst = dL_dmean2D + (size_t)id * 3 + (q - 4 - NSEM);
                else if (q < 4 + NSEM + 5) {
                    const int cidx = q - 4 - NSEM - 2;  // 0,1,2 -> x,y,w of the float4
                    dst = dL_dconic + (size_t)id * 4 + (cidx == 2 ? 3 : cidx);
                } else
                    dst = dL_dopacity + id;
                if (dst) atomicAdd(dst, v);
            }
        }
        __syncthreads();
        if (threadIdx.x < BWD_BATCH) s_touched[threadIdx.x] = 0;
    }
}

template <int S4>
void launch_fwd_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                   float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_fwd_k<S4, false><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec,
                                                              sc.semantics, sc.bg, out_color, out_sem, out_depth,
                                                              out_alpha, im.n_contrib, nullptr, nullptr, nullptr);
}

template <int S4>
void launch_bwd_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                   const float* out_alpha, const float* dL_dpix, const float* dL_dsem, const float* dL_ddepth,
                   const float* dL_dalpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                   float* dL_dsemantic, float* dL_ddepths, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_bwd_k<S4><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec, sc.semantics,
                                                       sc.bg, out_alpha, im.n_contrib, dL_dpix, dL_dsem, dL_ddepth,
                                                       dL_dalpha, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                                       dL_dsemantic, dL_ddepths);
}

}  // namespace

#define GOI_DISPATCH_S4(S, CALL)                     \
    switch (((S) + 3) / 4) {                         \
        case 1: CALL(1); break;                      \
        case 2: CALL(2); break;                      \
        case 3: CALL(3); break;                      \
        case 4: CALL(4); break;                      \
        case 5: CALL(5); break;                      \
        case 6: CALL(6); break;                      \
        case 7: CALL(7); break;                      \
        default: CALL(8); break;                     \
    }

void launch_render_fwd(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s) {
#define GOI_CALL(N) launch_fwd_s4<N>(sc, g, im, point_list, out_color, out_sem, out_depth, out_alpha, s)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

void launch_render_bwd(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       const float* out_alpha, const float* dL_dpix, const float* dL_dsem, const float* dL_ddepth,
                       const float* dL_dalpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolor, float* dL_dsemantic, float* dL_ddepths, hipStream_t s) {
#define GOI_CALL(N)                                                                                              \
    launch_bwd_s4<N>(sc, g, im, point_list, out_alpha, dL_dpix, dL_dsem, dL_ddepth, dL_dalpha, dL_dmean2D, dL_dconic, \
                     dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepths, s)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

void launch_trace_fwd(const GoiRasterScene& sc, const float* img_sem, const GeomView& g, const ImageView& im,
                      const uint32_t* point_list, float* out_color, float* gau_sem, int* num_gsem, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_fwd_k<1, true><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec, nullptr,
                                                            sc.bg, out_color, nullptr, nullptr, nullptr, im.n_contrib,
                                                            img_sem, gau_sem, num_gsem);
}

}  // namespace goi
