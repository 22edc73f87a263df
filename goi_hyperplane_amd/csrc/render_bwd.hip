// Backward tile blend for gfx950 -- ATOMIC-FREE, deterministic.  Restates the reference's backward renderCUDA
// (cuda_rasterizer/backward.cu:415-625): per pixel, back-to-front over the tile's sorted list
// starting at the forward's last contributor, same guards, T recovered as T_final / prod(1-alpha),
// gradients w.r.t. colour, semantics, depth, 2D mean (NDC units), conic (a, b, c) and opacity with
// the 0.99 clamp ignored.  What is different is HOW it is computed on CDNA4:
//
//  1. ONE WAVE = ONE WORKGROUP = one 8x8 quadrant (as in the forward): no workgroup barriers, no
//     cross-wave imbalance.  Each wave starts at ITS OWN last contributor (max over its 64 pixels of
//     n_contrib): the saturated tail of a tile list is never touched.
//  2. Lane-parallel staging in batches of 32 list entries.  The forward blend has left, per quadrant and 64 list positions,
//     the mask of the Gaussians that contributed to some pixel of the quadrant (MASKS): the wave stages and evaluates those
//     MEMBERS only -- mask word and ids requested a batch ahead; a member's record and `aux` word fetched when it is staged,
//     its colour / depth word and semantic row sent from memory to LDS by DMA at the same moment (no registers, one round
//     trip per batch).  Without the masks (bwd_masks 0) every lane tests one candidate's contribution ellipse against the
//     quadrant.  The member / hit mask lives in SGPRs (scalar bit scan).
//  3. The per-channel "accum_rec" recurrences (backward.cu:557,571,583,589) collapse into ONE scalar
//     recurrence per pixel: with d_i = <feature_i, dL/dpixel> (+ depth and alpha terms),
//         dL/dalpha_i = (d_i - R_i) * T_i - T_final/(1-alpha_i) * <bg, dL/dcolour>,
//         R_{i-1} = alpha_i * d_i + (1-alpha_i) * R_i          (R = <accum_rec, dL/dpixel>).
//  4. The per-Gaussian sums over the 64 pixels of the wave are MATRIX PRODUCTS and run on the matrix
//     cores (see "Two flushes" below for the arithmetic):
//         dL/dfeature[j][ch] = sum_pix w[pix][j] * dL/dpixel[pix][ch],      w = alpha * T
//         moments[j][m]      = sum_pix h[pix][j] * basis[pix][m],           h = opacity * G * dL/dalpha
//     with basis = (1, u, v, u^2, uv, v^2) in quadrant-centred pixel coordinates; the 2D-mean, conic
//     and opacity gradients are exact linear combinations of the six moments, expanded around the
//     Gaussian's centre right after the MFMA.  8 contributing Gaussians form a group; their w columns
//     (rows 0..7) and h columns (rows 8..15) are STACKED along M of one A operand and transposed
//     through a single LDS buffer; the B operand of the last block carries (r, g, b, depth) in columns
//     0..3 and the six basis functions in columns 4..9, so one MFMA yields the colour/depth gradients
//     (upper half of D) and the moments (lower half) at once.  The halves of D that pair w with the
//     basis or h with dL are never read.
//  5. NO ATOMICS.  The reference adds one float atomic per (pixel, Gaussian, quantity); on MI355X L2
//     atomics retire about one dword per clock per channel and would dominate this kernel.  Instead
//     every (quadrant, Gaussian) pair owns one row of a scratch buffer, addressed by the Gaussian's
//     EMIT-ORDER instance index (Gaussian-major: goff[g] + tile offset inside g's rectangle) times 4
//     plus the quadrant.  Rows are written once with plain stores plus a validity byte;
//     reduce_rows_k then sums each Gaussian's rows, which are contiguous in slot space, in a fixed
//     order.  Gradients are bit-reproducible run to run.
//
// Two flushes.  fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the vector rate (32 clocks per instruction per SIMD)
// and its time ADDS to the VALU time of the co-resident waves (tools/mfma_mix_probe.hip): the 32 instructions of
// one group flush were ~20 % of this kernel.  The default flush therefore runs at the 16-bit matrix rate on SPLIT
// operands and is still fp32-grade:
//   * w (= alpha T, in (0, 1]) and dL are carried as TWO f16 planes, hi = rne(s x), lo = rne(s x - hi), after an exact
//     power-of-two scaling s that puts them where f16 is densest: w 2^15, and every channel of dL times the 2^k that
//     brings the channel's largest |dL| of the quadrant to [2^14, 2^15).  |s x - hi - lo| <= 2^-22 |s x| for every value
//     within 2^17 of the top of the range (2^-25 of the top below that: for w that is alpha T < 2^-17, kept to >= 18
//     bits).  A product is  hi hi + hi lo + lo hi + lo lo  -- all four terms, each EXACT in the fp32 accumulator of
//     v_mfma_f32_16x16x32_f16 -- so a sum differs from the exact-fp32 chain by the roundings of the accumulation only.
//     The four terms cost TWO instructions per 32 pixels: rows 0..7 of A are not the eight members and rows 8..15 idle
//     weight, they are (member, plane) pairs -- A[4 i + r] = plane r >> 1 of member 2 i + (r & 1) -- so one instruction
//     forms  hi B  and  lo B  of eight members, and the two land in the SAME lane of D (elements r and r + 2): one add.
//     Colour / depth (4 channels) ride in one more block whose columns 0..3 hold their hi plane and 4..7 their lo plane.
//   * h (= opacity G dL/dalpha: any sign, any magnitude) is carried as THREE bf16 planes against the moment basis,
//     which is exact in bf16: hi + lo + t carries h to 2^-25, three v_mfma_f32_16x16x32_bf16 per flush (one per plane:
//     the lanes of rows 0..7 / 8..15 hold the two 32-pixel chunks against ONE chunk-centred basis, and the member's
//     lane shifts either chunk's moments to the Gaussian's centre).  The moments are sums that can cancel to a small
//     fraction of their terms: the opacity gradient of an isolated Gaussian under a +- upstream gradient
//     (|sum| ~ sum|.| / 600) was 2.1e-3 off with two bf16 planes and is 5.9e-5 off with three (soak case 8102/201).
//   9 matrix instructions per flush of 8 members (S <= 16).  Measured against the exact-fp32 flush on 1200 random
//   configurations (tools/flush_soak.py, profiles/r04_flush_equivalence*.json): the two differ by less than two legal
//   builds of the reference do; rounds 2-3 used two bf16 planes for w and dL (3 of 4 terms, 2^-17 per product), which
//   that soak showed to be ~6x noisier than fp32 on dL/dsh and dL/dsemantics -- replaced.
// F16 = false (bwd_variant 2) keeps the exact-fp32 flush (one fp32 FMA chain per output) for comparison.
#include "blend_common.h"

namespace goi {

namespace {


// Up to 16 semantic channels the kernel fits 128 VGPRs without spilling and 8 KB of LDS per wave:
// 4 waves per SIMD.  Wider features (17..32 channels) need ~200 VGPRs: 2 waves per SIMD.
#if !defined(GOI_BWD_WAVES)
#define GOI_BWD_WAVES 4
#endif
#define GOI_BWD_LAUNCH_BOUNDS __launch_bounds__(64, (S4 <= 4 ? GOI_BWD_WAVES : 2))

#ifndef GOI_TSTRIDE
#define GOI_TSTRIDE 66
#endif
constexpr int GROUP = 8;     // contributing Gaussians per MFMA group: rows 0..7 = w, rows 8..15 = h of 16x16x4
constexpr int BATCH = 32;    // list entries examined / staged per round (lanes 0..31): LDS, not lanes, is scarce
constexpr int TSTRIDE = GOI_TSTRIDE;  // row stride (floats) of the transposition buffer

template <int S4>
struct BwdCfg {
    static constexpr int NF4 = 1 + S4;          // staged float4 words per Gaussian
    static constexpr int NSEM = 4 * S4;         // padded semantic channels
    static constexpr int NCH = NSEM + 4;        // channel order: sem0.., r, g, b, depth
    static constexpr int NB = (NSEM + 15) / 16;  // 16-column MFMA blocks of semantic channels (+ 1 mixed block)
};

// F16: split-f16 flush (default); otherwise the exact-fp32 flush.
// MASKS (default): the forward blend has left the MEMBER mask of every (quadrant, 64 list positions) -- which Gaussians
// contributed to some pixel of the quadrant (render_fwd.hip, member_mask_ptr).  The wave then walks its list in 32-entry
// batches ALIGNED with the forward's rounds, fetches and stages members only, and evaluates members only: no candidate is
// tested against the quadrant (ellipse_hits_quadrant: ~45 lane-parallel instructions per batch and a 32-byte gather per
// candidate) and no hit that contributes nowhere is evaluated.  The member set is exactly the set of pairs the
// candidate-testing form (MASKS = false, bwd_masks 0) ends up flushing, in the same order: same rows, same bits.
template <int S4, bool F16, bool MASKS>
__global__ GOI_BWD_LAUNCH_BOUNDS void render_bwd_rows_k(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int gy,
    int n_quads, int S, const GaussRec* __restrict__ rec, const float* __restrict__ semantics,
    const int* __restrict__ radii, const uint4* __restrict__ aux, const float* __restrict__ bg,
    const float* __restrict__ out_alpha, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
    const float* __restrict__ dL_dpixsem, const float* __restrict__ dL_dpixdepth, const float* __restrict__ dL_dalphas,
    float* __restrict__ rows, uint8_t* __restrict__ flags, int row_floats,
    const uint32_t* __restrict__ counters, const uint32_t* __restrict__ qorder,
    const unsigned long long* __restrict__ qmask0, const unsigned long long* __restrict__ qmask,
    const uint32_t* __restrict__ qcost) {
    using Cfg = BwdCfg<S4>;
    const bool cull = counters[COUNTER_CULL] != 0;  // the rectangles the forward listed
    constexpr int NF4 = Cfg::NF4, NSEM = Cfg::NSEM, NCH = Cfg::NCH, NB = Cfg::NB;
    // Staging slots 0..BATCH-1 belong to the current batch; slots BATCH..BATCH+GROUP-1 carry the members of an
    // unfinished MFMA group across a batch boundary (see `jpack`).
    __shared__ f32x4 s_geo[BATCH + GROUP];   // (A3, A5, A1, A2) of the quadrant-centred log2-alpha polynomial (blend_common.h)
    __shared__ f32x4 s_geo2[BATCH + GROUP];  // (A0, A4, lim, slot index (bits))
    __shared__ float2 s_cen[BATCH + GROUP];  // Gaussian centre - quadrant centre (the flush expands the moments around it)
    // staged features, word-major: s_feat[i * BATCH + slot], i = 0: (r, g, b, depth), i >= 1: the semantic row's float4 words.
    // That is the layout LDS-DMA writes (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16 l, inactive lanes write
    // nothing -- tools/probes/lds_dma_probe.hip): slot = lane, so a member's words travel from memory to LDS without passing
    // through registers.
    __shared__ __attribute__((aligned(16))) float4 s_feat[BATCH * NF4];
    // [row][pixel] floats.  fp32 flush: rows 0..7 = w, rows 8..15 = h of the group's members, row stride TSTRIDE.
    // Split-f16 flush: the same rows at f16_row() (blend_common.h: 16-byte aligned, conflict-free for its reads); the
    // flush splits them into planes after reading its A operand (two dword stores per member cost the LDS half of
    // four 16-bit ones).
    constexpr int TS = TSTRIDE;  // (fp32 flush)
    constexpr int T_BYTES = F16 ? F16_FLOATS * 4 : 2 * GROUP * TSTRIDE * 4;
    __shared__ __attribute__((aligned(16))) char s_traw[T_BYTES];
    float* const s_t = reinterpret_cast<float*>(s_traw);

    // launch slot -> quadrant: tile order, or (qorder) the band's quadrants longest-first -- see quad_order_k
    const QuadGeom t = quad_geom_of(qorder ? (int)qorder[quad_slot()] : quad_slot(), W, H, gx, n_quads);
    if (t.tile < 0) return;
    if (counters[COUNTER_OVF]) return;  // truncated frame: no rows (reduce_rows_k writes zero gradients)
    const int lane = t.lane;
    const uint2 range = ranges[t.tile];
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    // Every per-pixel input of the wave is REQUESTED here, before anything waits: lanes outside the image read pixel 0 and
    // discard it (no divergent load blocks, each with its own wait), and the wave's own trip count comes from a scalar load.
    // The head of a wave used to be five dependent memory round trips (order -> n_contrib -> reduction -> out_alpha -> dL);
    // with 26 400 waves per launch that was a tenth of the kernel.
    const size_t pix_ld = t.inside ? pix_id : 0;
    const uint32_t nc_ld = n_contrib[pix_ld];
    const float oa_ld = out_alpha[pix_ld];
    float dLch[NCH];
    // an absent upstream gradient (NULL) is zero
    if (dL_dpixsem && S == NSEM) {  // (wave-uniform; the common case without a test per channel)
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLch[ch] = dL_dpixsem[ch * HW + pix_ld];
    } else {
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) dLch[ch] = (dL_dpixsem && ch < S) ? dL_dpixsem[ch * HW + pix_ld] : 0.f;
    }
    dLch[NSEM + 0] = dL_dpix ? dL_dpix[0 * HW + pix_ld] : 0.f;
    dLch[NSEM + 1] = dL_dpix ? dL_dpix[1 * HW + pix_ld] : 0.f;
    dLch[NSEM + 2] = dL_dpix ? dL_dpix[2 * HW + pix_ld] : 0.f;
    dLch[NSEM + 3] = dL_dpixdepth ? dL_dpixdepth[pix_ld] : 0.f;
    float dLa = dL_dalphas ? dL_dalphas[pix_ld] : 0.f;

    // ---- this wave's last contributor: list positions [0, n_proc) are all it has to look at
    // (the forward blend left it per quadrant for the launch order: one scalar load instead of a wave reduction)
    const int last_contributor = t.inside ? (int)nc_ld : 0;
    const int n_proc = qcost ? (int)qcost[__builtin_amdgcn_readfirstlane(4 * t.tile + t.q)] : wave_max_i32(last_contributor);
    if (n_proc == 0) return;  // nothing was composited in this quadrant
    const int rounds = (n_proc + BATCH - 1) / BATCH;

    // ---- per-pixel upstream gradients, channel order (sem0.., r, g, b, depth)
    const float T_final = t.inside ? (1.f - oa_ld) : 0.f;
    float T = T_final;
    if (!t.inside) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) dLch[ch] = 0.f;
        dLa = 0.f;
    }
    const float bg_dot = bg[0] * dLch[NSEM] + bg[1] * dLch[NSEM + 1] + bg[2] * dLch[NSEM + 2];
    float R = 0.f;
    f32x2 dL2[NCH / 2];  // the same gradients as register pairs for the packed dot product
#pragma unroll
    for (int i = 0; i < NCH / 2; i++) dL2[i] = f32x2{dLch[2 * i], dLch[2 * i + 1]};
    const f32x4* s_feat4 = reinterpret_cast<const f32x4*>(s_feat);

    // ---- MFMA B operands (fixed for the whole kernel), built once through LDS:
    //      bfrag[nb][s] = dL[pixel 4s + (lane>>4)][channel 16 nb + (lane&15)]
    const int kq = lane >> 4, mm = lane & 15;
    float bfrag[F16 ? 1 : NB][F16 ? 1 : 16];  // semantic blocks (fp32 flush)
    float bmix[F16 ? 1 : 16];  // mixed block: columns 0..3 = dL/d(r, g, b, depth), 4..9 = moment basis
    // split-f16 flush: Wh/Wl[block][chunk] = f16 hi / lo of  dL[pixel 32 chunk + 8 kq + i][semantic channel 16 block + mm]
    // * 2^k(channel)  (the channel's largest |dL| of this quadrant brought to [2^14, 2^15): two f16 planes then carry 22
    // bits of every value within 2^17 of that largest one); unscale[block] = 2^-15 / 2^k (the 2^15 is the scale of the
    // weights, see the staging store).  The four colour / depth channels share ONE operand: columns 0..3 their hi plane,
    // columns 4..7 their lo plane.  The moment basis is one bf16 operand for BOTH 32-pixel chunks:
    // (1, u, v', u^2, u v', v'^2) with v' = v -+ 2 centred on the chunk.
    f16x8 Wh[F16 ? NB : 1][2], Wl[F16 ? NB : 1][2], Cc[2];
    float unscale[F16 ? NB : 1], unscale_c = 0.f;
    bf16x8 Bas;
    // (the operands are transposed through s_t once per kernel: lane = pixel writes [pixel][channel], lane (kq, mm) reads
    // [pixel 8 kq + i][channel mm].  A row stride of 17 floats makes the writes conflict-free and the reads 2-way; the stride
    // of 16 it replaced cost 16-way writes and 4-way reads -- ~300 LDS cycles per block and wave, a fifth of this kernel's
    // LDS time in the counters; 5 instead of 4 for the four colour / depth channels: both conflict-free)
    constexpr int BT = 17;
    static_assert(64 * BT * 4 <= F16_FLOATS * 4 && 64 * 16 * 4 <= T_BYTES, "staging region too small");
    // basis_m(pixel p) in quadrant-centred coordinates u, v in [-3.5, 3.5] (pixel p = column p & 7, row p >> 3);
    // m = mm - 4: 1, u, v, u^2, uv, v^2
    const float k1 = mm == 4 ? 1.f : 0.f, ku = mm == 5 ? 1.f : 0.f, kv = mm == 6 ? 1.f : 0.f;
    const float kuu = mm == 7 ? 1.f : 0.f, kuv = mm == 8 ? 1.f : 0.f, kvv = mm == 9 ? 1.f : 0.f;
    auto basis = [&](int p) {
        const float u = (float)(p & 7) - 3.5f, v = (float)(p >> 3) - 3.5f;
        return k1 + ku * u + kv * v + kuu * (u * u) + kuv * (u * v) + kvv * (v * v);
    };
    if constexpr (F16) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int ch = nb * 16 + c;
                s_t[lane * BT + c] = ch < NSEM ? dLch[ch] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            float y[2][8];
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[c2][i] = s_t[(32 * c2 + 8 * kq + i) * BT + mm];
            f16_b_operand(y, Wh[nb], Wl[nb], unscale[nb]);
            __builtin_amdgcn_wave_barrier();
        }
        {
#pragma unroll
            for (int c = 0; c < 4; c++) s_t[lane * 5 + c] = dLch[NSEM + c];
            __builtin_amdgcn_wave_barrier();
            float y[2][8];
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[c2][i] = s_t[(32 * c2 + 8 * kq + i) * 5 + (mm & 3)];
            f16x8 hi[2], lo[2];
            f16_b_operand(y, hi, lo, unscale_c);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const u32x4 hu = __builtin_bit_cast(u32x4, hi[c2]), lu = __builtin_bit_cast(u32x4, lo[c2]);
                u32x4 cu;
#pragma unroll
                for (int i = 0; i < 4; i++) cu[i] = mm < 4 ? hu[i] : mm < 8 ? lu[i] : 0u;
                Cc[c2] = __builtin_bit_cast(f16x8, cu);
            }
            __builtin_amdgcn_wave_barrier();
        }
        {
            float y[8];
            const float vb = (float)kq - 1.5f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float u = (float)i - 3.5f;
                y[i] = mm == 0 ? 1.f : mm == 1 ? u : mm == 2 ? vb : mm == 3 ? u * u : mm == 4 ? u * vb : mm == 5 ? vb * vb : 0.f;
            }
            bf16x8 unused;
            split_pack8(y, Bas, unused);  // (exact: multiples of 1/4 up to 12.25)
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int ch = nb * 16 + c;
                s_t[lane * 16 + c] = ch < NSEM ? dLch[ch] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; s++) bfrag[nb][s] = s_t[(4 * s + kq) * 16 + mm];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int c = 0; c < 4; c++) s_t[lane * 4 + c] = dLch[NSEM + c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; s++) bmix[s] = mm < 4 ? s_t[(4 * s + kq) * 4 + mm] : basis(4 * s + kq);
        __builtin_amdgcn_wave_barrier();
    }
    // (wave-uniform values, pinned to SGPRs: the kernel sits at its VGPR limit and spilled them to scratch otherwise)
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    const float QCX = uniform(t.QX0 + 3.5f), QCY = uniform(t.QY0 + 3.5f);  // quadrant centre (pixel coordinates)
    const f32x2 uv = {t.pxf - QCX, t.pyf - QCY};                           // this lane's pixel, quadrant-centred
    const float half_W = uniform(0.5f * W), half_H = uniform(0.5f * H);

    // software prefetch of the next batch's id / position / box (one list entry per lane)
    // MASKS: batch kb covers list positions [32 kb, 32 kb + 32), lane l position 32 kb + l, and is walked from its highest
    // position down; its members are bits [32 (kb & 1), +32) of the forward's mask word kb >> 1 (wave-uniform, SGPRs).
    // Otherwise: batch b covers positions n_proc-1-32b downwards, lane l position n_proc-1-(32 b + l), every lane a candidate.
    uint32_t id_n = 0;
    unsigned long long w_n = 0;  // (MASKS) the forward's member word that holds the next batch's 32 bits
    float4 q0_n = make_float4(0, 0, 0, 0), q1_n = make_float4(1.f, 0.f, -1.f, -1.f);  // (conic c, opacity, hx, hy)
    const int tile_u = __builtin_amdgcn_readfirstlane(t.tile), q_u = __builtin_amdgcn_readfirstlane(t.q);
    const uint32_t x0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);
    auto prefetch = [&](int b) {
        if constexpr (MASKS) {
            // the batch's member word and its 32 ids (one line, whoever is a member) are requested TOGETHER, a batch ahead, and
            // neither is waited for here: the ids used to be fetched for the members only, i.e. behind a wait for the word --
            // a scalar-memory round trip at the top of every batch before its staging could begin.  (The id only: a member's
            // record is fetched when it is staged, together with the other gathers that wait there anyway -- its first two
            // words, held across the batch, were eight VGPRs of a kernel that sits at its limit.)
            w_n = *member_mask_ptr(const_cast<unsigned long long*>(qmask0), const_cast<unsigned long long*>(qmask), tile_u, q_u,
                                   x0_u, b >> 1);
            const int pos = b * BATCH + lane;
            if (lane < BATCH && pos < n_proc) id_n = point_list[range.x + (uint32_t)pos];
        } else {
            const int k = b * BATCH + lane;  // list position n_proc-1-k, walking back to front
            q1_n.z = -1.f;
            if (lane < BATCH && k < n_proc) {
                id_n = point_list[range.x + (n_proc - 1 - k)];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id_n);
                q0_n = r4[0];
                q1_n = r4[1];
            }
        }
    };
    prefetch(MASKS ? rounds - 1 : 0);

    int nslot = 0;  // filled slots of the current MFMA group (wave-uniform)
    // staging slot of every group member, one byte per member, in SGPRs: the flush finds a member's coefficients,
    // centre and row index where the staging lane left them -- no per-member copy (a lane-0 ds_write_b128 costs
    // the LDS 13 cycles whatever the number of active lanes)
    unsigned long long jpack = 0;

    // flushes `cnt` filled slots: D = [w]^T dL (NB blocks) and [h]^T basis, then one row per member
    // exchange area of the split-f16 flush: row rho of the moment product (8 floats); rows 8..15 start 16 floats later, so
    // that the four 16-lane groups of a store (rows 4 kq + r) fall on different banks
    auto xrow = [](int rho) { return rho * 8 + 16 * (rho >> 3); };
    auto flush_group = [&](int cnt) {
        f32x4 acc[NB];
        f32x4 accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
        if constexpr (F16) {
            // A[row mm] = plane (mm >> 1) & 1 (f16 hi / lo) of the weights of member 2 (mm >> 2) + (mm & 1): ONE instruction
            // forms  hi * B  and  lo * B, and lane (kq, mm) of D holds for column mm the hi products of members 2 kq and
            // 2 kq + 1 in elements 0, 1 and their lo products in elements 2, 3 -- one add each, in registers.  A lane reads
            // and splits 8 pixels of ONE w row (chunk = its plane) and takes the plane it needs of the other chunk from its
            // partner lane mm ^ 2 (one quad-permute DPP per register).  The h row of the same member, the same chunk,
            // goes to the moment product as three bf16 planes: row mm of that D is the member's moments over one 32-pixel
            // chunk in chunk-centred coordinates, and the member's lane shifts both chunks to the Gaussian's centre.
            const f32x4* wsrc = reinterpret_cast<const f32x4*>(s_t + f16_lane_offset(mm, kq));
            // (two phases, fenced for the instruction scheduler: interleaved, the planes of both live at once and the
            // kernel, already at its register limit, spills)
            {
                const f32x4* src = wsrc + (f16_row(GROUP) - f16_row(0)) / 4;  // (same lane address as the w row + a constant)
                const f32x4 a0 = src[0], a1 = src[1];
                const float y[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                bf16x8 Hh, Hl, Ht;
                split3_pack8(y, Hh, Hl, Ht);
                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ht, Bas, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Hl, Bas, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Hh, Bas, accx, 0, 0, 0);
            }
            f32x4 accc = {0.f, 0.f, 0.f, 0.f};
            {
                f16x8 A0, A1;
                f16_a_operands(wsrc, A0, A1);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Wl[nb][0], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Wl[nb][1], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Wh[nb][0], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Wh[nb][1], acc[nb], 0, 0, 0);
                }
                accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Cc[0], accc, 0, 0, 0);
                accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Cc[1], accc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
            if (mm < 8) {  // moments -> exchange area (16 rows of 8): rows 4 kq + r, one base address + immediates
                float* xb = s_t + xrow(4 * kq) + mm;
#pragma unroll
                for (int r = 0; r < 4; r++) xb[8 * r] = accx[r];
            }
            // members 2 kq (element 0) and 2 kq + 1 (element 1): hi products + lo products, unscaled -- as register pairs
            f32x2 sem[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) sem[nb] = (acc[nb].xy + acc[nb].zw) * unscale[nb];
            // colour / depth: columns 0..3 = (hi + lo of the weights) x hi of dL, columns 4..7 = x lo of dL
            f32x2 col = accc.xy + accc.zw;
            col.x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(col.x), 0x104, 0xF, 0xF, true));  // row_shl:4 = lane mm + 4
            col.y += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(col.y), 0x104, 0xF, 0xF, true));
            col *= unscale_c;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int j = 2 * kq + e;
                if (j < cnt) {
                    const int idx = (int)((jpack >> (8 * j)) & 0xFF);
                    float* dst = rows + (size_t)__float_as_uint(s_geo2[idx].w) * row_floats;
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int ch = nb * 16 + mm;
                        if (ch < NSEM) dst[ch] = sem[nb][e];
                    }
                    if (mm < 4) dst[NSEM + mm] = col[e];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float a = s_t[mm * TS + 4 * s + kq];
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag[nb][s], acc[nb], 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bmix[s], accx, 0, 0, 0);
            }
        }
        // D[row = 4*kq + r][col = mm]: rows 0..7 = w of slot `row` (x dL), rows 8..15 = h of slot row-8 (x basis)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4 && !F16; r++) {
            const int row = 4 * kq + r;
            if (row >= GROUP && mm >= 4 && mm < 12) s_t[(row - GROUP) * 8 + (mm - 4)] = accx[r];  // moments -> exchange area
            if (row < cnt) {
                const int idx = (int)((jpack >> (8 * row)) & 0xFF);
                float* dst = rows + (size_t)__float_as_uint(s_geo2[idx].w) * row_floats;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int ch = nb * 16 + mm;
                    if (ch < NSEM) dst[ch] = acc[nb][r];
                }
                if (mm < 4) dst[NSEM + mm] = accx[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // moments -> (mean2D.x, mean2D.y, conic a, b, c, opacity): one lane per group member
        if (lane < cnt) {
            const float4 m03 = *reinterpret_cast<const float4*>(&s_t[lane * 8]);
            const float2 m45 = *reinterpret_cast<const float2*>(&s_t[lane * 8 + 4]);
            const int idx = (int)((jpack >> (8 * lane)) & 0xFF);
            const f32x4 g = s_geo[idx];
            const f32x4 g2 = s_geo2[idx];
            const float2 cen = s_cen[idx];
            const uint32_t slot = __float_as_uint(g2.w);  // (emit-order instance) * 4 + quadrant
            const float Dx = cen.x, Dy = cen.y;           // centre - quadrant centre: dx = Dx - u, dy = Dy - v
            float ca, cb, cc, inv_o;  // the conic and 1 / opacity, back from the staged words
            coef_decode(g, g2, ca, cb, cc, inv_o);
            float m0, sx, sy, sxx, sxy, syy;
            // moments (1, u, v, uu, uv, vv) of h about a point (Dx, Dy) away from the Gaussian's centre -> sums of h dx^k dy^l
            auto shift = [&](const float4& a, const float2& b, float DX, float DY) {
                m0 += a.x;
                sx += DX * a.x - a.y;                                  // sum h dx
                sy += DY * a.x - a.z;                                  // sum h dy
                sxx += DX * DX * a.x - 2.f * DX * a.y + a.w;           // sum h dx^2
                sxy += DX * DY * a.x - DX * a.z - DY * a.y + b.x;      // sum h dx dy
                syy += DY * DY * a.x - 2.f * DY * a.z + b.y;           // sum h dy^2
            };
            m0 = sx = sy = sxx = sxy = syy = 0.f;
            if constexpr (F16) {
                // rows 4 (l >> 1) + (l & 1) and + 2 of the exchange area: the member's moments over pixel rows 0..3 (centre
                // 2 above the quadrant's) and over pixel rows 4..7 (2 below)
                const int r1 = 4 * (lane >> 1) + (lane & 1);
                shift(*reinterpret_cast<const float4*>(&s_t[xrow(r1)]), *reinterpret_cast<const float2*>(&s_t[xrow(r1) + 4]), Dx, Dy + 2.f);
                shift(*reinterpret_cast<const float4*>(&s_t[xrow(r1 + 2)]), *reinterpret_cast<const float2*>(&s_t[xrow(r1 + 2) + 4]), Dx, Dy - 2.f);
            } else {
                shift(m03, m45, Dx, Dy);
            }
            float* dst = rows + (size_t)slot * row_floats + NCH;
            // (the moments already carry the factor `opacity` of dL/dG = opacity dL/dalpha)
            dst[0] = -half_W * (ca * sx + cb * sy);  // dL/dmean2D.x (NDC units)
            dst[1] = -half_H * (cc * sy + cb * sx);  // dL/dmean2D.y
            dst[2] = -0.5f * sxx;                    // dL/dconic a
            dst[3] = -0.5f * sxy;                    // dL/dconic b
            dst[4] = -0.5f * syy;                    // dL/dconic c
            dst[5] = m0 * inv_o;                     // dL/dopacity = sum G dL/dalpha
            flags[slot] = 1;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (int bi = 0; bi < rounds; bi++) {
        const int b = MASKS ? rounds - 1 - bi : bi;
        const uint32_t id = id_n;
        float4 q0 = q0_n, q1 = q1_n;
        bool hit;
        unsigned long long m;
        if constexpr (MASKS) {
            const uint32_t mem = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_n >> (32 * (b & 1))));
            m = mem;  // (wave-uniform)
            hit = lane < BATCH && ((mem >> lane) & 1u);
            if (bi + 1 < rounds) prefetch(b - 1);
        } else {
            hit = ellipse_hits_quadrant(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, t.QX0, t.QY0);  // hx < 0 for absent lanes
            if (bi + 1 < rounds) prefetch(b + 1);
            m = __ballot(hit);
        }
        if (m == 0) continue;
        // ---- stage the hits (slot = lane)
        if (hit) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + id);
            // (r, g, b, depth) and the semantic row go to LDS by DMA, requested FIRST: no registers (the row alone would be 16
            // in a kernel at its limit), no ds_write, and -- unlike loads placed where their data is stored, after the
            // arithmetic on the record -- in flight together with the record and aux: one memory round trip per batch
            // instead of two.  Loads return in order: when the record has arrived, so have these.
            const float* srow = semantics + (size_t)id * S;
            const bool rows_of_float4 = (S & 3) == 0;  // (wave-uniform)
            lds_dma16(r4 + 2, &s_feat[0]);
            if (rows_of_float4) {
#pragma unroll
                for (int i = 0; i < S4; i++)
                    lds_dma16(reinterpret_cast<const float4*>(srow) + i, &s_feat[(1 + i) * BATCH]);
            }
            if constexpr (MASKS) {
                q0 = r4[0];
                q1 = r4[1];
            }
            int x0, y0, x1, y1;
            const uint4 ax = aux[id];  // first slot, radius, tile mask: one 16-byte gather (three scattered ones before)
            listed_rect(q0.x, q0.y, (int)ax.y, q1.z, q1.w, cull, gx, gy, x0, y0, x1, y1);
            const uint32_t inst = ax.x + tile_instance(aux_mask(ax), t.tx, t.ty, x0, y0, x1);
            const PolyCoef pc = poly_coefs(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, QCX, QCY);
            // the staging addresses are formed HERE from an opaque copy of the lane id: hoisted out of the batch
            // loop they cost one live VGPR per destination (the compiler spilled four of them to scratch)
            int sl = lane;
            asm volatile("" : "+v"(sl));
            s_geo[sl] = f32x4{pc.A35.x, pc.A35.y, pc.A12.x, pc.A12.y};
            s_geo2[sl] = f32x4{pc.A0, pc.A4, pc.lim, __uint_as_float(inst * 4u + (uint32_t)t.q)};
            s_cen[sl] = make_float2(q0.x - QCX, q0.y - QCY);
            if (!rows_of_float4) {  // a row that is not a whole number of float4 words: through registers
#pragma unroll
                for (int i = 0; i < S4; i++) {
                    float4 v;
                    v.x = (4 * i + 0 < S) ? srow[4 * i + 0] : 0.f;
                    v.y = (4 * i + 1 < S) ? srow[4 * i + 1] : 0.f;
                    v.z = (4 * i + 2 < S) ? srow[4 * i + 2] : 0.f;
                    v.w = (4 * i + 3 < S) ? srow[4 * i + 3] : 0.f;
                    s_feat[(1 + i) * BATCH + sl] = v;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA words are in LDS before any lane reads them
        __builtin_amdgcn_wave_barrier();

        while (m) {
            int j, pos0;  // staging slot, 0-based list position
            if constexpr (MASKS) {
                j = 31 - __builtin_clz((uint32_t)m);  // back to front: the batch's highest member first
                m &= ~(1ull << j);
                pos0 = b * BATCH + j;
            } else {
                j = __builtin_ctzll(m);
                m &= m - 1;
                pos0 = n_proc - 1 - (b * BATCH + j);
            }
            const f32x4 g = s_geo[j];
            const f32x4 g2 = s_geo2[j];
            const PairEval e = eval_poly(g.xy, g.zw, g2.x, g2.y, g2.z, uv);
            const bool live = pos0 < last_contributor;
            if constexpr (!MASKS)  // (a member contributes somewhere by construction)
                if (!any_all(live, e.below, e.seen)) continue;
            const bool c = live && e.hit;

            // <feature, dL/dpixel> as packed fp32 FMAs (v_pk_fma_f32: two channels per instruction)
            const f32x4 f0 = s_feat4[j];  // r, g, b, depth
            f32x2 da = f0.xy * dL2[NSEM / 2];
            f32x2 db = f0.zw * dL2[NSEM / 2 + 1];
#pragma unroll
            for (int i = 0; i < S4; i++) {
                const f32x4 f = s_feat4[(1 + i) * BATCH + j];
                da = __builtin_elementwise_fma(f.xy, dL2[2 * i], da);
                db = __builtin_elementwise_fma(f.zw, dL2[2 * i + 1], db);
            }
            const f32x2 dot2 = da + db;
            const float dotv = (dot2.x + dot2.y) + dLa;
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
            if constexpr (F16) {
                // (one wave-uniform row offset: the h row is the w row plus a constant, which the LDS instruction's immediate
                // offset carries)
                const int wrow = nslot * 72 + ((nslot & 1) << 2);  // = f16_row(nslot), nslot < 8
                s_t[wrow + lane] = wgt;
                s_t[wrow + lane + (f16_row(GROUP) - f16_row(0))] = hval;
            } else {
                s_t[nslot * TS + lane] = wgt;
                s_t[(GROUP + nslot) * TS + lane] = hval;
            }
            jpack |= (unsigned long long)j << (8 * nslot);
            nslot++;
            if (nslot == GROUP) {
                flush_group(GROUP);
                nslot = 0;
                jpack = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (nslot > 0 && bi + 1 < rounds) {
            // the next batch overwrites the staging slots: move the unfinished group's members to the carry slots
            // (member l -> slot BATCH + l; a member carried before is already there)
            // (lane l reads slot idx_l -- below BATCH, or BATCH + l itself -- and writes BATCH + l: no lane writes
            // what another lane reads)
            const int idx = (int)((jpack >> (8 * (lane & 7))) & 0xFF);
            if (lane < nslot) {
                s_geo[BATCH + lane] = s_geo[idx];
                s_geo2[BATCH + lane] = s_geo2[idx];
                s_cen[BATCH + lane] = s_cen[idx];
            }
            constexpr unsigned long long CARRY = 0x0706050403020100ull + 0x0101010101010101ull * BATCH;
            jpack = CARRY & ((1ull << (8 * nslot)) - 1ull);  // nslot < GROUP = 8 here
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (nslot > 0) flush_group(nslot);
}

// Launch order of the backward's quadrant waves.  The hardware deals workgroup b to XCD b % 8 and, inside an XCD, starts
// workgroups in index order as wave slots free up.  A quadrant's cost is known before the kernel starts -- its wave walks
// the tile list back from the largest n_contrib of its 64 pixels (qcost[], left by the forward blend) -- and the costs
// differ by an order of magnitude, so in tile order the kernel ends with a long tail of half-empty CUs (~15 % of its
// wave slots idle).  One workgroup per XCD band sorts the band's quadrants by cost, longest first (LPT scheduling): a
// STABLE counting sort on 64 cost buckets, so quadrants of similar cost keep their tile order and the four quadrants of
// a tile, and neighbouring tiles, still run close together on the XCD whose L2 holds their Gaussians.  (A global
// heaviest-first map had lost more in L2 locality than it gained in balance.)  Which wave computes which row does not
// change any result.
constexpr int QO_THREADS = 1024, QO_WAVES = QO_THREADS / 64, QO_BUCKETS = 64, QO_ROUNDS = 16;
constexpr int QO_CLEAR_BLOCKS = 248;  // workgroups 8 .. 255 of the launch clear the backward's validity bytes
__global__ __launch_bounds__(QO_THREADS) void quad_order_k(const uint32_t* __restrict__ qcost, int n_quads, int per,
                                                           uint32_t* __restrict__ qorder, int shift,
                                                           uint8_t* __restrict__ clear_flags,
                                                           const uint32_t* __restrict__ n_dev, uint32_t cap,
                                                           uint32_t* __restrict__ clear_ctl) {
    if (blockIdx.x >= 8) {
        if (clear_ctl && blockIdx.x == 8 && threadIdx.x < 8) clear_ctl[threadIdx.x] = 0u;  // (BwdScratchView::big_ctl)
        // validity bytes of the row slots this frame can use: 4 per instance, zeroed 16 bytes per lane (the scratch layout
        // rounds the array up to 256 bytes, so the last partial quad is inside it)
        if (!clear_flags) return;
        const size_t nq = ((size_t)min(*n_dev, cap) * 4 + 15) / 16;
        uint4* dst = reinterpret_cast<uint4*>(clear_flags);
        for (size_t i = (size_t)(blockIdx.x - 8) * QO_THREADS + threadIdx.x; i < nq; i += (size_t)QO_CLEAR_BLOCKS * QO_THREADS)
            dst[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    __shared__ uint32_t cnt[QO_WAVES][QO_BUCKETS];
    const int band = blockIdx.x, base = band * per;
    const int n = max(0, min(per, n_quads - base));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < QO_WAVES * QO_BUCKETS; i += QO_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    // wave w owns the contiguous elements [w * chunk, (w + 1) * chunk): (wave, round, lane) order == index order
    const int chunk = (per + QO_WAVES - 1) / QO_WAVES;  // <= 64 * QO_ROUNDS (checked by the launcher)
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t d_r[QO_ROUNDS], rank_r[QO_ROUNDS];
#pragma unroll
    for (int r = 0; r < QO_ROUNDS; r++) {
        if (r * 64 >= chunk) break;  // (uniform: 4 of the 16 rounds at 1600 x 1056)
        const int e = w * chunk + r * 64 + lane;
        const bool ok = r * 64 + lane < chunk && e < n;
        // bucket 0 = heaviest: 16 list positions per bucket, everything above 1008 together
        const uint32_t c = ok ? qcost[base + e] : 0u;
        const uint32_t d = (uint32_t)(QO_BUCKETS - 1) - min((uint32_t)(QO_BUCKETS - 1), c >> shift);
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t prev = ok ? cnt[w][d] : 0u;  // (same-wave LDS accesses execute in program order)
        const uint32_t below = (uint32_t)__popcll(peers & lt);
        d_r[r] = d;
        rank_r[r] = prev + below;
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) cnt[w][d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix over (bucket, wave): one wave, lane = bucket
        uint32_t tot = 0;
        for (int ww = 0; ww < QO_WAVES; ww++) tot += cnt[ww][lane];
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(incl, dd, 64);
            if (lane >= dd) incl += o;
        }
        uint32_t run = incl - tot;
        for (int ww = 0; ww < QO_WAVES; ww++) {
            const uint32_t t = cnt[ww][lane];
            cnt[ww][lane] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < QO_ROUNDS; r++) {
        const int e = w * chunk + r * 64 + lane;
        if (r * 64 + lane < chunk && e < n) qorder[base + cnt[w][d_r[r]] + rank_r[r]] = (uint32_t)(base + e);
    }
    for (int i = n + threadIdx.x; i < per; i += QO_THREADS) qorder[base + i] = 0xFFFFFFFFu;  // (slots past the last quadrant)
}

template <int S4>
void launch_bwd_rows_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                        const int* radii, const float* out_alpha, const float* dL_dpix, const float* dL_dsem,
                        const float* dL_ddepth, const float* dL_dalpha, const BwdScratchView& scr, hipStream_t s,
                        const unsigned long long* qmask) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
#define GOI_LAUNCH_ROWS(F16, MASKS)                                                                                    \
    render_bwd_rows_k<S4, F16, MASKS><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(                                     \
        im.ranges, point_list, sc.W, sc.H, gx, gy, n_quads, sc.S, g.rec, sc.semantics, radii, g.aux, sc.bg, out_alpha, \
        im.n_contrib, dL_dpix, dL_dsem, dL_ddepth, dL_dalpha, scr.rows, scr.flags, bwd_row_floats(sc.S),                 \
        g.counters, quad_order_enabled(sc.W, sc.H) ? im.qorder : nullptr, im.qmask0, qmask, im.qcost)
    const bool masks = qmask != nullptr && g_options.bwd_masks != 0;
    if ((g_options.bwd_variant & 15) == 2) {  // exact-fp32 flush
        if (masks) GOI_LAUNCH_ROWS(false, true);
        else GOI_LAUNCH_ROWS(false, false);
    } else {  // split-f16 flush
        if (masks) GOI_LAUNCH_ROWS(true, true);
        else GOI_LAUNCH_ROWS(true, false);
    }
#undef GOI_LAUNCH_ROWS
}

}  // namespace

bool quad_order_enabled(int W, int H) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int per = quad_grid(gx * gy * 4) / 8;
    // (a band of more than 16 x 1024 quadrants -- a frame beyond ~4 K x 2.2 K pixels -- keeps tile order)
    return g_options.bwd_order != 0 && (per + QO_WAVES - 1) / QO_WAVES <= 64 * QO_ROUNDS;
}

bool launch_quad_order(const GoiRasterScene& sc, const ImageView& im, hipStream_t s, uint8_t* clear_flags,
                       const uint32_t* n_dev, uint32_t cap, uint32_t* clear_ctl) {
    if (!quad_order_enabled(sc.W, sc.H)) return false;
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4, per = quad_grid(n_quads) / 8;
    quad_order_k<<<dim3(clear_flags ? 8 + QO_CLEAR_BLOCKS : 8), dim3(QO_THREADS), 0, s>>>(
        im.qcost, n_quads, per, im.qorder, 3 + g_options.bwd_order, clear_flags, n_dev, cap, clear_ctl);
    return true;
}

void launch_render_bwd_rows(const GoiRasterScene& sc, const GeomView& g, const ImageView& im,
                            const uint32_t* point_list, const int* radii, const float* out_alpha, const float* dL_dpix,
                            const float* dL_dsem, const float* dL_ddepth, const float* dL_dalpha,
                            const BwdScratchView& scr, hipStream_t s, const unsigned long long* qmask) {
#define GOI_CALL(N) \
    launch_bwd_rows_s4<N>(sc, g, im, point_list, radii, out_alpha, dL_dpix, dL_dsem, dL_ddepth, dL_dalpha, scr, s, qmask)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

}  // namespace goi
