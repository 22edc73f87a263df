// Backward tile blend for gfx950, WORKGROUP-PER-TILE variant (4 waves in lock step per batch, partial
// sums of the four quadrants combined in LDS before they leave as global atomics: ~3.7x fewer L2
// atomic operations than the wave-per-quadrant kernel of render_bwd.hip, at the price of five
// workgroup barriers per 64-Gaussian batch).  Restates the reference's backward renderCUDA
// (cuda_rasterizer/backward.cu:415-625): per pixel, back-to-front over the tile's sorted list
// starting at the forward's last contributor, same guards, T recovered as T_final / prod(1-alpha),
// gradients w.r.t. colour, semantics, depth, 2D mean (NDC units), conic (a, b, c) and opacity with
// the 0.99 clamp ignored.  What is different is HOW it is computed on CDNA4:
//
//  1. The workgroup starts at the tile's LAST contributor (max over its pixels of n_contrib), not at
//     the end of the list: the saturated tail of a tile list is never staged.
//  2. The per-channel "accum_rec" recurrences (backward.cu:557,571,583,589) collapse into ONE scalar
//     recurrence per pixel: with d_i = <feature_i, dL/dpixel> (+ depth and alpha terms),
//         dL/dalpha_i = (d_i - R_i) * T_i - T_final/(1-alpha_i) * <bg, dL/dcolour>,
//         R_{i-1} = alpha_i * d_i + (1-alpha_i) * R_i          (R = <accum_rec, dL/dpixel>).
//  3. The per-Gaussian sums over the 64 pixels of a wave are MATRIX PRODUCTS and run on the matrix
//     cores in exact fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain):
//         dL/dfeature[j][ch] = sum_pix w[pix][j] * dL/dpixel[pix][ch],      w = alpha * T
//         moments[j][m]      = sum_pix h[pix][j] * basis[pix][m],           h = G * dL/dalpha
//     with basis = (1, u, v, u^2, uv, v^2) in tile-centred pixel coordinates; the 2D-mean, conic
//     and opacity gradients are exact linear combinations of the six moments (expanded around the
//     Gaussian's centre at flush time).  Per wave, 16 contributing Gaussians form a group; their w
//     and h columns are transposed through LDS into the MFMA A-operand layout.
//  4. Partial sums of the four waves meet in LDS; one global atomic per (tile, Gaussian, quantity)
//     leaves the workgroup (the reference issues one per (pixel, Gaussian, quantity)).
#include "blend_common.h"

namespace goi {

namespace {


constexpr int BATCH = 64;    // Gaussians staged per round
constexpr int GROUP = 16;    // contributing Gaussians per MFMA group (the M of 16x16x4)
constexpr int TSTRIDE = 66;  // row stride (floats) of the transposition buffers: conflict-free A reads

template <int S4>
struct BwdTileCfg {
    static constexpr int NF4 = 1 + S4;            // staged float4 words per Gaussian
    static constexpr int NCH = 4 + 4 * S4;        // (r,g,b,depth) + padded semantic channels
    static constexpr int NB = (NCH + 15) / 16;    // 16-column MFMA blocks for the feature gradient
    static constexpr int NQ = NCH + 6;            // + six moments
};

template <int S4>
__global__ __launch_bounds__(256) void render_bwd_tile_k(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int S,
    const GaussRec* __restrict__ rec, const float* __restrict__ semantics, const float* __restrict__ bg,
    const float* __restrict__ out_alpha, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
    const float* __restrict__ dL_dpixsem, const float* __restrict__ dL_dpixdepth, const float* __restrict__ dL_dalphas,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_dsemantic, float* __restrict__ dL_ddepths,
    const uint32_t* __restrict__ counters) {
    if (counters[COUNTER_OVF]) return;  // truncated frame: the accumulators stay at the zeros they were cleared to
    using Cfg = BwdTileCfg<S4>;
    constexpr int NF4 = Cfg::NF4, NCH = Cfg::NCH, NB = Cfg::NB, NQ = Cfg::NQ;
    __shared__ float4 s_geo[BATCH];            // x, y, conic a, b
    __shared__ float4 s_geo2[BATCH];           // conic c, opacity, hx, hy
    __shared__ float4 s_feat[BATCH * NF4];     // (r,g,b,depth), semantics
    __shared__ uint32_t s_id[BATCH];
    __shared__ float s_acc[BATCH * NQ];
    __shared__ uint32_t s_touched[BATCH];
    __shared__ float s_wt[4][GROUP * TSTRIDE];  // per wave: w columns, [slot][pixel]
    __shared__ float s_ht[4][GROUP * TSTRIDE];  // per wave: h columns
    __shared__ int s_gid[4][GROUP];
    __shared__ int s_red[4];

    const TileGeom t = tile_geom(W, H, gx);
    const uint2 range = ranges[t.tile];
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    const int last_contributor = t.inside ? (int)n_contrib[pix_id] : 0;

    // ---- tile-wide and wave-wide last contributor
    int wmax = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (t.lane == 0) s_red[t.w] = wmax;
    __syncthreads();
    const int n_proc = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    if (n_proc == 0) return;  // nothing was composited in this tile
    const int rounds = (n_proc + BATCH - 1) / BATCH;

    // ---- per-pixel upstream gradients, channel order (r, g, b, depth, sem0..)
    const float T_final = t.inside ? (1.f - out_alpha[pix_id]) : 0.f;
    float T = T_final;
    float dLch[NCH];
    float dLa = 0.f;
    if (t.inside) {
        dLch[0] = dL_dpix ? dL_dpix[0 * HW + pix_id] : 0.f;  // an absent upstream gradient (NULL) is zero
        dLch[1] = dL_dpix ? dL_dpix[1 * HW + pix_id] : 0.f;
        dLch[2] = dL_dpix ? dL_dpix[2 * HW + pix_id] : 0.f;
        dLch[3] = dL_dpixdepth ? dL_dpixdepth[pix_id] : 0.f;
#pragma unroll
        for (int ch = 0; ch < 4 * S4; ch++) dLch[4 + ch] = (dL_dpixsem && ch < S) ? dL_dpixsem[ch * HW + pix_id] : 0.f;
        dLa = dL_dalphas ? dL_dalphas[pix_id] : 0.f;
    } else {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) dLch[ch] = 0.f;
    }
    const float bg_dot = bg[0] * dLch[0] + bg[1] * dLch[1] + bg[2] * dLch[2];
    float R = 0.f;

    // ---- MFMA B operands (fixed for the whole kernel), built once through LDS:
    //      bfrag[nb][s] = dL[pixel 4s + (lane>>4)][channel 16 nb + (lane&15)]
    float bfrag[NB][16];
    {
        float* stage = &s_wt[t.w][0];  // 64 x NCH floats <= GROUP*TSTRIDE? no: use both buffers
        float* stage2 = &s_ht[t.w][0];
        // pixel-major rows of 16 channels per block; blocks 0.. live in s_wt, s_ht alternately
        static_assert(64 * 16 <= GROUP * TSTRIDE, "staging region too small");
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            float* buf = (nb & 1) ? stage2 : stage;
            if (nb >= 2) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int ch = nb * 16 + c;
                buf[t.lane * 16 + c] = ch < NCH ? dLch[ch] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; s++) bfrag[nb][s] = buf[(4 * s + (t.lane >> 4)) * 16 + (t.lane & 15)];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // moment basis B operand, generated on the fly: lane (kq = lane>>4, m = lane&15) needs
    // basis_m(pixel 4s + kq); tile-centred coordinates u, v in [-7.5, 7.5]
    const int kq = t.lane >> 4, mm = t.lane & 15;
    const float u_even = (float)(kq + 8 * (t.w & 1)) - 7.5f;  // pixel column of 4s+kq is 4(s&1)+kq
    const float v_base = (float)(8 * (t.w >> 1)) - 7.5f;      // pixel row is s>>1
    const float k1 = mm == 0 ? 1.f : 0.f, ku = mm == 1 ? 1.f : 0.f, kv = mm == 2 ? 1.f : 0.f;
    const float kuu = mm == 3 ? 1.f : 0.f, kuv = mm == 4 ? 1.f : 0.f, kvv = mm == 5 ? 1.f : 0.f;

    for (int i = threadIdx.x; i < BATCH * NQ; i += 256) s_acc[i] = 0.f;

    const float QX0 = (float)(t.tx * TILE + (t.w & 1) * 8), QY0 = (float)(t.ty * TILE + (t.w >> 1) * 8);
    const float TCX = (float)(t.tx * TILE) + 7.5f, TCY = (float)(t.ty * TILE) + 7.5f;
    float* const wt = &s_wt[t.w][0];
    float* const ht = &s_ht[t.w][0];
    int* const gid = &s_gid[t.w][0];

    for (int b = 0; b < rounds; b++) {
        __syncthreads();  // previous flush done (s_acc zero, staging buffers free)
        // ---- cooperative staging, back to front: slot j holds list position n_proc-1-(b*BATCH+j).
        //      4 threads per Gaussian: part 0 -> q0, part 1 -> q1 + q2, parts 2,3 -> semantic words.
        {
            const int slot = threadIdx.x >> 2, part = threadIdx.x & 3;
            const int k = b * BATCH + slot;
            if (part == 3) s_touched[slot] = 0;  // every flush reader is past the loop-top barrier
            if (k < n_proc) {
                const uint32_t id = point_list[range.x + (n_proc - 1 - k)];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id);
                if (part == 0) {
                    s_geo[slot] = r4[0];
                    s_id[slot] = id;
                } else if (part == 1) {
                    const float4 q1 = r4[1], q2 = r4[2];
                    s_geo2[slot] = q1;                // conic c, opacity, hx, hy
                    s_feat[slot * NF4] = q2;          // r, g, b, depth
                } else {
                    const float* srow = semantics + (size_t)id * S;
                    if ((S & 3) == 0) {
#pragma unroll
                        for (int i = 0; i < (S4 + 1) / 2; i++) {
                            const int w4 = (part - 2) + 2 * i;
                            if (w4 < S4) s_feat[slot * NF4 + 1 + w4] = reinterpret_cast<const float4*>(srow)[w4];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < (S4 + 1) / 2; i++) {
                            const int w4 = (part - 2) + 2 * i;
                            if (w4 < S4) {
                                float4 v;
                                v.x = (4 * w4 + 0 < S) ? srow[4 * w4 + 0] : 0.f;
                                v.y = (4 * w4 + 1 < S) ? srow[4 * w4 + 1] : 0.f;
                                v.z = (4 * w4 + 2 < S) ? srow[4 * w4 + 2] : 0.f;
                                v.w = (4 * w4 + 3 < S) ? srow[4 * w4 + 3] : 0.f;
                                s_feat[slot * NF4 + 1 + w4] = v;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- this wave's candidates: staged Gaussians whose contribution box meets the quadrant
        //      and that are not behind every pixel's last contributor
        unsigned long long cand;
        {
            const int k = b * BATCH + t.lane;
            bool hit = false;
            if (k < n_proc && (n_proc - 1 - k) < wmax) {
                const float4 g = s_geo[t.lane];
                const float4 g2 = s_geo2[t.lane];
                hit = ellipse_hits_quadrant(g.x, g.y, g.z, g.w, g2.x, g2.y, g2.z, g2.w, QX0, QY0);
            }
            cand = __ballot(hit);
        }
        int nslot = 0;  // filled slots of the current MFMA group (wave-uniform)

        // flushes `cnt` filled slots: D = [w]^T dL (NB blocks) and [h]^T basis, then adds into s_acc
        auto flush_group = [&](int cnt) {
            f32x4 acc[NB];
            f32x4 accm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int off = mm * TSTRIDE + 4 * s + kq;
                const float aw = wt[off];
                const float ah = ht[off];
                const float u = u_even + (float)(4 * (s & 1));
                const float v = v_base + (float)(s >> 1);
                const float bm = k1 + ku * u + kv * v + kuu * (u * u) + kuv * (u * v) + kvv * (v * v);
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bfrag[nb][s], acc[nb], 0, 0, 0);
                accm = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bm, accm, 0, 0, 0);
            }
            // D[row = 4*(lane>>4) + r][col = lane&15]: row = group slot, col = channel / moment
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * kq + r;
                if (row < cnt) {
                    const int gi = gid[row];
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int ch = nb * 16 + mm;
                        if (ch < NCH) atomicAdd(&s_acc[gi * NQ + ch], acc[nb][r]);
                    }
                    if (mm < 6) atomicAdd(&s_acc[gi * NQ + NCH + mm], accm[r]);
                    if (mm == 0) s_touched[gi] = 1;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        while (cand) {
            const int j = __builtin_ctzll(cand);
            cand &= cand - 1;
            const int pos0 = n_proc - 1 - (b * BATCH + j);  // 0-based list position
            const float4 g = s_geo[j];
            const float4 g2 = s_geo2[j];
            // same coefficients, bit for bit, as the ones the forward staged for this (quadrant, Gaussian)
            const PolyCoef pc = poly_coefs(g.x, g.y, g.z, g.w, g2.x, g2.y, QX0 + 3.5f, QY0 + 3.5f);
            const PairEval e = eval_poly(pc.A35, pc.A12, pc.A0, pc.A4, pc.lim,
                                         f32x2{t.pxf - (QX0 + 3.5f), t.pyf - (QY0 + 3.5f)});
            const bool c = (pos0 < last_contributor) && e.hit;
            if (__builtin_amdgcn_ballot_w64(c) == 0) continue;  // (the builtin takes the bool: no int round trip)

            const float4 f0 = s_feat[j * NF4];
            float dotv = f0.x * dLch[0] + f0.y * dLch[1] + f0.z * dLch[2] + f0.w * dLch[3] + dLa;
#pragma unroll
            for (int i = 0; i < S4; i++) {
                const float4 f = s_feat[j * NF4 + 1 + i];
                dotv += f.x * dLch[4 + 4 * i + 0] + f.y * dLch[4 + 4 * i + 1] + f.z * dLch[4 + 4 * i + 2] +
                        f.w * dLch[4 + 4 * i + 3];
            }
            const float one_m_a = 1.f - e.alpha;
            const float inv = __builtin_amdgcn_rcpf(one_m_a);
            const float Tn = T * inv;
            float wgt = 0.f, hval = 0.f;
            if (c) {
                const float dL_dopa = (dotv - R) * Tn - (T_final * inv) * bg_dot;
                R = e.alpha * dotv + one_m_a * R;
                T = Tn;
                wgt = e.alpha * Tn;
                hval = e.E * dL_dopa;  // opacity * G * dL/dalpha: the moments carry the factor `opacity`
            }
            wt[nslot * TSTRIDE + t.lane] = wgt;
            ht[nslot * TSTRIDE + t.lane] = hval;
            if (t.lane == 0) gid[nslot] = j;
            nslot++;
            if (nslot == GROUP) {
                flush_group(GROUP);
                nslot = 0;
            }
        }
        if (nslot > 0) flush_group(nslot);
        __syncthreads();

        // ---- flush this batch: one global atomic per (Gaussian, quantity) that received anything.
        //      Items 0..NCH-1 of a slot are feature sums; item NCH turns the six moments into
        //      (mean2D.x, mean2D.y, conic a, b, c, opacity).
        for (int i = threadIdx.x; i < BATCH * (NCH + 1); i += 256) {
            const int gi = i / (NCH + 1), q = i - gi * (NCH + 1);
            if (!s_touched[gi]) continue;
            const uint32_t id = s_id[gi];
            if (q < NCH) {
                const float val = s_acc[gi * NQ + q];
                s_acc[gi * NQ + q] = 0.f;
                float* dst = nullptr;
                if (q < 3)
                    dst = dL_dcolor + (size_t)id * 3 + q;
                else if (q == 3)
                    dst = dL_ddepths + id;
                else if (q - 4 < S)
                    dst = dL_dsemantic + (size_t)id * S + (q - 4);
                if (dst) atomicAdd(dst, val);
            } else {
                float* m = &s_acc[gi * NQ + NCH];
                const float4 g = s_geo[gi];
                const float4 g2 = s_geo2[gi];
                const float Dx = g.x - TCX, Dy = g.y - TCY;  // dx = Dx - u, dy = Dy - v
                const float m0 = m[0], mu = m[1], mv = m[2], muu = m[3], muv = m[4], mvv = m[5];
#pragma unroll
                for (int k = 0; k < 6; k++) m[k] = 0.f;
                const float sx = Dx * m0 - mu;                             // sum h dx
                const float sy = Dy * m0 - mv;                             // sum h dy
                const float sxx = Dx * Dx * m0 - 2.f * Dx * mu + muu;      // sum h dx^2
                const float sxy = Dx * Dy * m0 - Dx * mv - Dy * mu + muv;  // sum h dx dy
                const float syy = Dy * Dy * m0 - 2.f * Dy * mv + mvv;      // sum h dy^2
                atomicAdd(dL_dmean2D + (size_t)id * 3 + 0, -(0.5f * W) * (g.z * sx + g.w * sy));
                atomicAdd(dL_dmean2D + (size_t)id * 3 + 1, -(0.5f * H) * (g2.x * sy + g.w * sx));
                atomicAdd(dL_dconic + (size_t)id * 4 + 0, -0.5f * sxx);
                atomicAdd(dL_dconic + (size_t)id * 4 + 1, -0.5f * sxy);
                atomicAdd(dL_dconic + (size_t)id * 4 + 3, -0.5f * syy);
                atomicAdd(dL_dopacity + id, m0 / g2.y);  // a contributing Gaussian has opacity >= 1/255
            }
        }
    }
}

template <int S4>
void launch_bwd_tile_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                   const float* out_alpha, const float* dL_dpix, const float* dL_dsem, const float* dL_ddepth,
                   const float* dL_dalpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                   float* dL_dsemantic, float* dL_ddepths, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_bwd_tile_k<S4><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec, sc.semantics,
                                                       sc.bg, out_alpha, im.n_contrib, dL_dpix, dL_dsem, dL_ddepth,
                                                       dL_dalpha, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                                       dL_dsemantic, dL_ddepths, g.counters);
}

}  // namespace

void launch_render_bwd_tile(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       const float* out_alpha, const float* dL_dpix, const float* dL_dsem, const float* dL_ddepth,
                       const float* dL_dalpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolor, float* dL_dsemantic, float* dL_ddepths, hipStream_t s) {
#define GOI_CALL(N)                                                                                              \
    launch_bwd_tile_s4<N>(sc, g, im, point_list, out_alpha, dL_dpix, dL_dsem, dL_ddepth, dL_dalpha, dL_dmean2D, dL_dconic, \
                     dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepths, s)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

}  // namespace goi
