// Feature-gradient-only backward blend for gfx950.
//
// The reference's training script optimises ONLY the per-Gaussian semantic features by default
// (arguments/__init__.py:85-90: semantic_finetune = True, every other *_finetune = False;
// scene/gaussian_model.py:185-246 freezes the rest), yet its backward kernel always computes every
// gradient (cuda_rasterizer/backward.cu:415-625).  dL/dsemantics needs none of the alpha-gradient
// machinery:
//     dL/dsem[g][ch] = sum_pix w[pix][g] * dL/dpixel_sem[pix][ch],      w = alpha * T
// so this kernel keeps the structure of render_bwd_rows_k (one wave = one 8x8 quadrant, lane-parallel
// staging of box-test hits, back-to-front walk from the wave's own last contributor, T recovered as
// T_final / prod(1 - alpha), MFMA reduction over the 64 pixels, one scratch row per (quadrant,
// Gaussian), no atomics) and drops the rest: no feature rows are staged, there is no <feature, dL>
// product, no dL/dalpha recurrence, no moments; 16 members form a group and all 16 MFMA rows carry w.
// The w values, the arithmetic of the MFMA reduction (F16: the split-f16 flush of render_bwd.hip -- the same operand
// construction (blend_common.h: f16_b_operand, f16_a_operands) and the same instruction order per output, two sets of
// eight members per group; otherwise the exact-fp32 flush) and the reduce order are those of the full kernel, so the
// result is bit-identical to the dL/dsemantics of the full backward in the same mode.
#include "blend_common.h"

namespace goi {

namespace {


constexpr int SGROUP = 16;
constexpr int SBATCH = 32;
constexpr int STSTRIDE = 66;

// MASKS: the wave walks the member masks the forward blend left instead of testing every list entry against its quadrant
// (render_bwd.hip, render_bwd_rows_k: same batches, same order, same bits).
template <int S4, bool F16, bool MASKS>
__global__ __launch_bounds__(64) void render_bwd_sem_k(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int gy,
    int n_quads, int S, const GaussRec* __restrict__ rec, const int* __restrict__ radii,
    const uint4* __restrict__ aux, const float* __restrict__ out_alpha, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpixsem, float* __restrict__ rows, uint8_t* __restrict__ flags, int row_floats,
    const uint32_t* __restrict__ counters, const uint32_t* __restrict__ qorder,
    const unsigned long long* __restrict__ qmask0, const unsigned long long* __restrict__ qmask,
    const uint32_t* __restrict__ qcost) {
    const bool cull = counters[COUNTER_CULL] != 0;  // the rectangles the forward listed
    constexpr int NSEM = 4 * S4, NB = (NSEM + 15) / 16;
    __shared__ f32x4 s_geo[SBATCH];           // (A3, A5, A1, A2) of the quadrant-centred log2-alpha polynomial
    __shared__ f32x4 s_geo2[SBATCH];          // (A0, A4, lim, slot index (bits))
    // w columns, [member][pixel] floats; the split flush keeps them at f16_row() (blend_common.h)
    constexpr int TS = STSTRIDE;  // (fp32 flush)
    constexpr int T_BYTES = F16 ? F16_FLOATS * 4 : SGROUP * STSTRIDE * 4;
    static_assert(64 * (F16 ? 17 : 16) * 4 <= T_BYTES, "staging region too small");
    __shared__ __attribute__((aligned(16))) char s_traw[T_BYTES];
    float* const s_t = reinterpret_cast<float*>(s_traw);
    __shared__ uint32_t s_slot[SGROUP];

    const QuadGeom t = quad_geom_of(qorder ? (int)qorder[quad_slot()] : quad_slot(), W, H, gx, n_quads);  // (render_bwd.hip: quad_order_k)
    if (t.tile < 0) return;
    if (counters[COUNTER_OVF]) return;  // truncated frame: no rows (the row reduction writes zero gradients)
    const int lane = t.lane;
    const uint2 range = ranges[t.tile];
    const float QCX = t.QX0 + 3.5f, QCY = t.QY0 + 3.5f;  // quadrant centre
    const f32x2 uv = {t.pxf - QCX, t.pyf - QCY};         // this lane's pixel, quadrant-centred
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    // (all per-pixel inputs are requested before anything waits, lanes outside the image read pixel 0: render_bwd.hip)
    const size_t pix_ld = t.inside ? pix_id : 0;
    const uint32_t nc_ld = n_contrib[pix_ld];
    const float oa_ld = out_alpha[pix_ld];
    float dLsem[NSEM];
    if (S == NSEM) {
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLsem[ch] = dL_dpixsem[ch * HW + pix_ld];
    } else {
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLsem[ch] = ch < S ? dL_dpixsem[ch * HW + pix_ld] : 0.f;
    }
    const int last_contributor = t.inside ? (int)nc_ld : 0;
    // (the forward blend left it per quadrant for the launch order: one scalar load instead of a wave reduction)
    const int n_proc = qcost ? (int)qcost[__builtin_amdgcn_readfirstlane(4 * t.tile + t.q)] : wave_max_i32(last_contributor);
    if (n_proc == 0) return;
    const int rounds = (n_proc + SBATCH - 1) / SBATCH;
    float T = t.inside ? (1.f - oa_ld) : 0.f;

    // MFMA B operands: bfrag[nb][s] = dL[pixel 4s + (lane>>4)][channel 16 nb + (lane&15)]
    const int kq = lane >> 4, mm = lane & 15;
    float bfrag[F16 ? 1 : NB][F16 ? 1 : 16];
    f16x8 Wh[F16 ? NB : 1][2], Wl[F16 ? NB : 1][2];  // split: planes of dL[pixel 32 chunk + 8 kq + i][column mm] 2^k(column)
    float unscale[F16 ? NB : 1];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int ch = nb * 16 + c;
            s_t[lane * (F16 ? 17 : 16) + c] = (t.inside && ch < NSEM) ? dLsem[ch] : 0.f;  // (row stride: render_bwd.hip, BT)
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (!F16) {
#pragma unroll
            for (int s = 0; s < 16; s++) bfrag[nb][s] = s_t[(4 * s + kq) * 16 + mm];
        } else {
            float y[2][8];
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[c2][i] = s_t[(32 * c2 + 8 * kq + i) * 17 + mm];
            f16_b_operand(y, Wh[nb], Wl[nb], unscale[nb]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    uint32_t id_n = 0;
    unsigned long long w_n = 0;  // (MASKS) the forward's member word that holds the next batch's 32 bits
    float4 q0_n = make_float4(0, 0, 0, 0), q1_n = make_float4(1.f, 0.f, -1.f, -1.f);  // (conic c, opacity, hx, hy)
    const int tile_u = __builtin_amdgcn_readfirstlane(t.tile), q_u = __builtin_amdgcn_readfirstlane(t.q);
    const uint32_t x0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);
    auto prefetch = [&](int b) {
        if constexpr (MASKS) {  // batch b = list positions [32 b, 32 b + 32), lane l position 32 b + l; members only
            // (member word and the batch's 32 ids requested together, a batch ahead, neither waited for here; the record is
            // fetched when the member is staged, with the gather that waits there anyway: render_bwd.hip)
            w_n = *member_mask_ptr(const_cast<unsigned long long*>(qmask0), const_cast<unsigned long long*>(qmask), tile_u, q_u,
                                   x0_u, b >> 1);
            const int pos = b * SBATCH + lane;
            if (lane < SBATCH && pos < n_proc) id_n = point_list[range.x + (uint32_t)pos];
        } else {
            const int k = b * SBATCH + lane;
            q1_n.z = -1.f;
            if (lane < SBATCH && k < n_proc) {
                id_n = point_list[range.x + (n_proc - 1 - k)];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id_n);
                q0_n = r4[0];
                q1_n = r4[1];
            }
        }
    };
    prefetch(MASKS ? rounds - 1 : 0);
    int nslot = 0;

    auto flush_group = [&](int cnt) {
        f32x4 acc[F16 ? 1 : NB];
#pragma unroll
        for (int nb = 0; nb < (F16 ? 1 : NB); nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
        if constexpr (!F16) {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float a = s_t[mm * TS + 4 * s + kq];
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag[nb][s], acc[nb], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * kq + r;  // D[row = member][col = channel]
                if (row < cnt) {
                    float* dst = rows + (size_t)s_slot[row] * row_floats;
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int ch = nb * 16 + mm;
                        if (ch < NSEM) dst[ch] = acc[nb][r];
                    }
                }
            }
        } else {
            // two sets of eight members; per set the operands, the four products per block and their order are those of
            // render_bwd_rows_k (rows of D are independent: what the other rows hold does not matter)
#pragma unroll
            for (int set = 0; set < 2; set++) {
                if (8 * set >= cnt) break;  // (wave-uniform)
                const f32x4* wsrc = reinterpret_cast<const f32x4*>(s_t + f16_lane_offset(mm, kq) + set * (f16_row(8) - f16_row(0)));
                f16x8 A0, A1;
                f16_a_operands(wsrc, A0, A1);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Wl[nb][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Wl[nb][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Wh[nb][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Wh[nb][1], a, 0, 0, 0);
                    const f32x2 sem = (a.xy + a.zw) * unscale[nb];  // members 8 set + 2 kq (+ 1): hi products + lo products
                    const int ch = nb * 16 + mm;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int j = 8 * set + 2 * kq + e;
                        if (j < cnt && ch < NSEM) rows[(size_t)s_slot[j] * row_floats + ch] = sem[e];
                    }
                }
            }
        }
        if (lane < cnt) flags[s_slot[lane]] = 1;
        __builtin_amdgcn_wave_barrier();
    };

    for (int bi = 0; bi < rounds; bi++) {
        const int b = MASKS ? rounds - 1 - bi : bi;
        const uint32_t id = id_n;
        float4 q0 = q0_n, q1 = q1_n;  // (this kernel needs nothing else of a record)
        bool hit;
        unsigned long long m;
        if constexpr (MASKS) {
            const uint32_t mem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_n >> (32 * (b & 1))));
            m = mem;
            hit = lane < SBATCH && ((mem >> lane) & 1u);
            if (bi + 1 < rounds) prefetch(b - 1);
        } else {
            hit = ellipse_hits_quadrant(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, t.QX0, t.QY0);
            if (bi + 1 < rounds) prefetch(b + 1);
            m = __ballot(hit);
        }
        if (m == 0) continue;
        if (hit) {
            if constexpr (MASKS) {
                const float4* r4 = reinterpret_cast<const float4*>(rec + id);
                q0 = r4[0];
                q1 = r4[1];
            }
            int x0, y0, x1, y1;
            const uint4 ax = aux[id];  // first slot, radius, tile mask: one 16-byte gather (three scattered ones before)
            listed_rect(q0.x, q0.y, (int)ax.y, q1.z, q1.w, cull, gx, gy, x0, y0, x1, y1);
            const uint32_t inst = ax.x + tile_instance(aux_mask(ax), t.tx, t.ty, x0, y0, x1);
            const PolyCoef pc = poly_coefs(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, QCX, QCY);
            s_geo[lane] = f32x4{pc.A35.x, pc.A35.y, pc.A12.x, pc.A12.y};
            s_geo2[lane] = f32x4{pc.A0, pc.A4, pc.lim, __uint_as_float(inst * 4u + (uint32_t)t.q)};
        }
        __builtin_amdgcn_wave_barrier();
        // the next candidate's coefficients are requested one trip ahead: a member costs ~20 VALU instructions here, the
        // LDS round trip at the head of every trip was most of it (in the full backward the same prefetch lost: it is
        // register-bound)
        auto next_slot = [&](unsigned long long mm_) {  // MASKS: back to front = the batch's highest member first
            return MASKS ? 31 - __builtin_clz((uint32_t)mm_) : __builtin_ctzll(mm_);
        };
        int j_n = next_slot(m);
        f32x4 g_n = s_geo[j_n], g2_n = s_geo2[j_n];
        while (m) {
            const int j = j_n;
            m &= ~(1ull << j);
            const int pos0 = MASKS ? b * SBATCH + j : n_proc - 1 - (b * SBATCH + j);
            const f32x4 g = g_n;
            const f32x4 g2 = g2_n;
            j_n = m ? next_slot(m) : j;
            g_n = s_geo[j_n];
            g2_n = s_geo2[j_n];
            const PairEval e = eval_poly(g.xy, g.zw, g2.x, g2.y, g2.z, uv);
            const bool live = pos0 < last_contributor;
            if constexpr (!MASKS)  // (a member contributes somewhere by construction)
                if (!any_all(live, e.below, e.seen)) continue;
            const bool c = live && e.hit;
            const float one_m_a = 1.f - e.alpha;
            const float inv = __builtin_amdgcn_rcpf(one_m_a);
            const float Tn = T * inv;
            float wgt = 0.f;
            if (c) {
                T = Tn;
                wgt = e.alpha * Tn;
            }
            s_t[(F16 ? f16_row(nslot) : nslot * TS) + lane] = wgt;
            if (lane == 0) s_slot[nslot] = __float_as_uint(g2.w);
            nslot++;
            if (nslot == SGROUP) {
                flush_group(SGROUP);
                nslot = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nslot > 0) flush_group(nslot);
}

template <int S4>
void launch_bwd_sem_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       const int* radii, const float* out_alpha, const float* dL_dsem, float* rows, uint8_t* flags,
                       int row_floats, hipStream_t s, const unsigned long long* qmask) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
#define GOI_LAUNCH_SEM(SP, MK)                                                                                         \
    render_bwd_sem_k<S4, SP, MK><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(                                          \
        im.ranges, point_list, sc.W, sc.H, gx, gy, n_quads, sc.S, g.rec, radii, g.aux, out_alpha, im.n_contrib, dL_dsem, \
        rows, flags, row_floats, g.counters, quad_order_enabled(sc.W, sc.H) ? im.qorder : nullptr, im.qmask0, qmask, im.qcost)
    const bool masks = qmask != nullptr && g_options.bwd_masks != 0;
    if ((g_options.bwd_variant & 15) == 2) {  // exact-fp32 flush, as in the full backward
        if (masks) GOI_LAUNCH_SEM(false, true);
        else GOI_LAUNCH_SEM(false, false);
    } else {
        if (masks) GOI_LAUNCH_SEM(true, true);
        else GOI_LAUNCH_SEM(true, false);
    }
#undef GOI_LAUNCH_SEM
}

}  // namespace

// rows: [4N][row_floats] with row_floats = 16 * ceil(4*ceil(S/4) / 16); flags [4N] zeroed by the caller
void launch_render_bwd_sem(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                           const int* radii, const float* out_alpha, const float* dL_dsem, float* rows, uint8_t* flags,
                           int row_floats, hipStream_t s, const unsigned long long* qmask) {
#define GOI_CALL(N) launch_bwd_sem_s4<N>(sc, g, im, point_list, radii, out_alpha, dL_dsem, rows, flags, row_floats, s, qmask)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

}  // namespace goi
