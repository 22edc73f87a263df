// Forward tile blend, "pixels x Gaussians" mapping (fwd_variant 2, EXPERIMENT; the default is render_fwd.hip).
//
// north_star names a "wavefront-level alpha scan"; VERDICT r04 asked for a denser-lane mapping of the blend to be BUILT and
// TIMED instead of argued about.  This is that kernel: the 16 pixels x 4 Gaussians hybrid.
//
//   * one wave = one 8x8 quadrant, as in render_fwd_k (same grid, same staging of a round's candidates in LDS, same outputs:
//     images, n_contrib, qcost, the member masks the backward walks);
//   * the quadrant is FOUR 4x4 sub-blocks; every candidate is tested against each sub-block's own rectangle of pixel centres
//     (exact ellipse test), so each sub-block has its own -- shorter -- hit list;
//   * the sub-blocks are taken one after the other; inside one, lane = 4 * pixel + g: the 64 lanes hold 16 pixels x FOUR
//     CONSECUTIVE ENTRIES of the sub-block's hit list.  alpha and the 20 channel products of the four entries are evaluated in
//     parallel; the transmittance recurrence T <- T (1 - alpha) runs through the four lanes of a pixel as three dependent
//     DPP multiplications in LIST ORDER (CR/forward.cu:352-357: the stopping entry is defined by the sequential product), so
//     n_contrib, the member masks and every weight alpha T are bit-identical to render_fwd_k's; the channel sums of the four
//     lanes meet once, at the end of the kernel (one quad reduction per channel: a different association of one fp32 sum).
//
// What it costs and what it gains is in DESIGN.md ("the pixels x Gaussians hybrid, measured"): fewer loop trips (a trip covers
// a quarter of the quadrant but four list entries), more vector instructions per trip (per-lane LDS addresses, the DPP chain),
// four hit tests per candidate, and four sub-blocks' accumulators in registers.
#include "blend_common.h"

namespace goi {

namespace {

template <int CTRL, int BANK>
__device__ __forceinline__ float quad_dpp(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK, false));
}
template <int CTRL>
__device__ __forceinline__ int quad_dpp_i(int src) {
    return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xf, 0xf, false);
}
constexpr int QP_SHIFT1 = 0x90;   // quad_perm:[0,0,1,2]
constexpr int QP_SHIFT1_K2 = 0x94;  // quad_perm:[0,1,1,2]: lanes 2, 3 read the lane below, lanes 0, 1 themselves
constexpr int QP_SHIFT1_K3 = 0xA4;  // quad_perm:[0,1,2,2]
constexpr int QP_BCAST3 = 0xFF;   // quad_perm:[3,3,3,3]
constexpr int QP_SWAP1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int QP_SWAP2 = 0x4E;    // quad_perm:[2,3,0,1]
// The transmittance after each of a pixel's four entries, in LIST ORDER and with the reference's association:
// c_g = ((T (1 - a_0)) (1 - a_1)) ... (1 - a_g), lane g of the quad.  DPP cannot write single lanes of a quad (bank_mask masks
// whole quads), so step k lets the lanes below k read THEMSELVES and multiply by one: after step k lane k is final, the lanes
// above hold values that the later steps replace.  f_k = (g >= k ? om : 1) do not depend on the chain.
__device__ __forceinline__ float quad_chain(float T_in, float om, int g) {
    const float f1 = g >= 1 ? om : 1.f, f2 = g >= 2 ? om : 1.f, f3 = g >= 3 ? om : 1.f;
    float c = T_in * om;
    c = quad_dpp<QP_SHIFT1, 0xf>(0.f, c) * f1;
    c = quad_dpp<QP_SHIFT1_K2, 0xf>(0.f, c) * f2;
    c = quad_dpp<QP_SHIFT1_K3, 0xf>(0.f, c) * f3;
    return c;
}
__device__ __forceinline__ float quad_sum(float x) {
    x += quad_dpp<QP_SWAP1, 0xf>(0.f, x);
    x += quad_dpp<QP_SWAP2, 0xf>(0.f, x);
    return x;
}

template <int S4, bool MASKS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void render_fwd_g4_k(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                      int W, int H, int gx, int n_quads, int S, const GaussRec* __restrict__ rec,
                                                      const float* __restrict__ semantics, const float* __restrict__ bg,
                                                      float* __restrict__ out_color, float* __restrict__ out_sem,
                                                      float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                      uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ qcost,
                                                      unsigned long long* __restrict__ qmask0,
                                                      unsigned long long* __restrict__ qmask, uint32_t* __restrict__ frame_flags,
                                                      uint32_t* __restrict__ host_words, uint32_t stamp) {
    constexpr int NF4 = 1 + S4;
    constexpr int NSEM = 4 * S4;
    __shared__ f32x4 s_geo[64];
    __shared__ f32x4 s_geo2[64];
    __shared__ float4 s_feat[64 * NF4];
    __shared__ uint8_t s_list[4][64];  // per sub-block: the round's staging slots that hit it, in list order
    __shared__ uint8_t s_member[64];   // slot contributed to some pixel of the quadrant this round
    const f32x4* s_feat4 = reinterpret_cast<const f32x4*>(s_feat);

    if (host_words && blockIdx.x == 0 && threadIdx.x < 32) {
        host_words[threadIdx.x] = (frame_flags - COUNTER_OVF)[threadIdx.x];
        __threadfence_system();  // every lane's word is on its way to the host before ...
        if (threadIdx.x == 0 && stamp)  // ... this use's sequence number says so (api.hip: STAMP_WORD)
            __hip_atomic_store(&host_words[HOST_STAMP_WORD], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const QuadGeom t = quad_geom(W, H, gx, n_quads);
    if (t.tile < 0) return;
    const int tq = quad_slot();
    const uint2 range = ranges[t.tile];
    const float QCX = t.QX0 + 3.5f, QCY = t.QY0 + 3.5f;
    const int len = (int)(range.y - range.x);
    const int rounds = (len + 63) / 64;
    const size_t HW = (size_t)W * H;
    const int lane = t.lane;
    const int g = lane & 3, pix = lane >> 2, lx = pix & 3, ly = pix >> 2;
    const int qx0 = (int)t.QX0, qy0 = (int)t.QY0;
    // quadrant-centred coordinates of this lane's pixel in the left / right and upper / lower sub-blocks
    const float ucol[2] = {(float)lx - 3.5f, (float)lx + 0.5f}, vrow[2] = {(float)ly - 3.5f, (float)ly + 0.5f};

    // per sub-block state (s = sx + 2 sy)
    float Tl[4], Tacc[4];
    uint32_t lastc[4];
    f32x2 C2[4][2];
    f32x2 Cs2[4][NSEM / 2];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const bool in = (qx0 + (s & 1) * 4 + lx < W) && (qy0 + (s >> 1) * 4 + ly < H);
        Tl[s] = in ? 1.0f : 0.0f;
        Tacc[s] = 1.0f;
        lastc[s] = 0;
        C2[s][0] = C2[s][1] = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NSEM / 2; i++) Cs2[s][i] = f32x2{0.f, 0.f};
    }

    uint32_t id_n = 0;
    float4 q0_n = make_float4(0, 0, 0, 0), q1_n = make_float4(1.f, 0.f, -1.f, -1.f), q2_n = make_float4(0, 0, 0, 0);
    auto prefetch = [&](int b) {
        const int k = b * 64 + lane;
        q1_n.z = -1.f;
        if (k < len) {
            id_n = point_list[range.x + k];
            const float4* r4 = reinterpret_cast<const float4*>(rec + id_n);
            q0_n = r4[0];
            q1_n = r4[1];
            q2_n = r4[2];
        }
    };
    if (rounds > 0) prefetch(0);

    const int tile_u = __builtin_amdgcn_readfirstlane(t.tile), q_u = __builtin_amdgcn_readfirstlane(t.q);
    const uint32_t x0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);
    auto live_mask = [&](int s) { return __builtin_amdgcn_ballot_w64(Tl[s] != 0.0f); };

    for (int b = 0; b < rounds; b++) {
        if ((live_mask(0) | live_mask(1) | live_mask(2) | live_mask(3)) == 0) break;
        const uint32_t id = id_n;
        const float4 q0 = q0_n, q1 = q1_n, q2 = q2_n;
        // ---- the candidate against the four sub-blocks' rectangles of pixel centres (blend_common.h: ellipse_hits_quadrant,
        // here for 4x4 rectangles; what is common to the four tests is formed once)
        bool h[4] = {false, false, false, false};
        {
            const float x = q0.x, y = q0.y, ca = q0.z, cb = q0.w, cc = q1.x, o = q1.y, hx = q1.z, hy = q1.w;
            const bool box = (hx >= 0.f) && (x - hx <= t.QX0 + 7.f) && (x + hx >= t.QX0) && (y - hy <= t.QY0 + 7.f) && (y + hy >= t.QY0);
            if (__builtin_amdgcn_ballot_w64(box) != 0) {
                const bool unbounded = !(hx < 3.0e38f);
                const float tau = 1.01f * 0.6931471805599453f * __builtin_amdgcn_logf(255.f * o) + 0.0101f;
                const float lim = tau * 1.0001f + 1e-4f;
                const float rb_c = -cb * __builtin_amdgcn_rcpf(cc), rb_a = -cb * __builtin_amdgcn_rcpf(ca);
                auto edge = [&](float dx, float dy) {
                    const float t1 = 0.5f * ca * dx * dx, t2 = 0.5f * cc * dy * dy, t3 = cb * dx * dy;
                    return (t1 + t2 + t3) - 1e-6f * (fabsf(t1) + fabsf(t2) + fabsf(t3));
                };
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float X0 = t.QX0 + (float)((s & 1) * 4), Y0 = t.QY0 + (float)((s >> 1) * 4);
                    const bool bx = box && (x - hx <= X0 + 3.f) && (x + hx >= X0) && (y - hy <= Y0 + 3.f) && (y + hy >= Y0);
                    const float ux0 = X0 - x, ux1 = ux0 + 3.f, uy0 = Y0 - y, uy1 = uy0 + 3.f;
                    const bool centre = ux0 <= 0.f && ux1 >= 0.f && uy0 <= 0.f && uy1 >= 0.f;
                    const float qmin =
                        fminf(fminf(edge(ux0, fminf(fmaxf(rb_c * ux0, uy0), uy1)), edge(ux1, fminf(fmaxf(rb_c * ux1, uy0), uy1))),
                              fminf(edge(fminf(fmaxf(rb_a * uy0, ux0), ux1), uy0), edge(fminf(fmaxf(rb_a * uy1, ux0), ux1), uy1)));
#ifdef GOI_G4_BOXTEST
                    (void)centre; (void)qmin; (void)unbounded; (void)lim;
                    h[s] = bx && ellipse_hits_quadrant(x, y, ca, cb, cc, o, hx, hy, t.QX0, t.QY0);
#else
                    h[s] = bx && (unbounded || centre || !(qmin > lim));
#endif
                }
            }
        }
        if (b + 1 < rounds) prefetch(b + 1);
        const bool hit = h[0] || h[1] || h[2] || h[3];
        unsigned long long m[4];
#pragma unroll
        for (int s = 0; s < 4; s++) m[s] = __builtin_amdgcn_ballot_w64(h[s]);
        s_member[lane] = 0;
        if ((m[0] | m[1] | m[2] | m[3]) != 0 && hit) {
            const PolyCoef pc = poly_coefs(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, QCX, QCY);
            s_geo[lane] = f32x4{pc.A35.x, pc.A35.y, pc.A12.x, pc.A12.y};
            s_geo2[lane] = f32x4{pc.A0, pc.A4, pc.lim, 0.f};
            s_feat[lane * NF4] = q2;
            const float* srow = semantics + (size_t)id * S;
            if ((S & 3) == 0) {
#pragma unroll
                for (int i = 0; i < S4; i++) s_feat[lane * NF4 + 1 + i] = reinterpret_cast<const float4*>(srow)[i];
            } else {
#pragma unroll
                for (int i = 0; i < S4; i++) {
                    float4 v;
                    v.x = (4 * i + 0 < S) ? srow[4 * i + 0] : 0.f;
                    v.y = (4 * i + 1 < S) ? srow[4 * i + 1] : 0.f;
                    v.z = (4 * i + 2 < S) ? srow[4 * i + 2] : 0.f;
                    v.w = (4 * i + 3 < S) ? srow[4 * i + 3] : 0.f;
                    s_feat[lane * NF4 + 1 + i] = v;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
                if (h[s]) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[s] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[s], 0u));
                    s_list[s][rank] = (uint8_t)lane;
                }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- the sub-blocks one after the other: 16 pixels x 4 list entries per trip
        const uint32_t pos0 = (uint32_t)(b * 64 + 1);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int n = __builtin_popcountll(m[s]);  // (wave-uniform)
            if (n == 0 || live_mask(s) == 0) continue;
            const f32x2 uv = {ucol[s & 1], vrow[s >> 1]};
#ifdef GOI_G4_PREFETCH
            // the next trip's slot and coefficients are requested a trip ahead (two dependent LDS round trips otherwise open every trip)
            bool valid_n = g < n;
            int slot_n = (int)s_list[s][valid_n ? g : n - 1];
            f32x4 ga_n = s_geo[slot_n], ha_n = s_geo2[slot_n];
#endif
            for (int t0 = 0; t0 < n; t0 += 4) {
#ifdef GOI_G4_PREFETCH
                const bool valid = valid_n;
                const int slot = slot_n;
                const f32x4 ga = ga_n, ha = ha_n;
                {
                    const int idx_n = t0 + 4 + g;
                    valid_n = idx_n < n;
                    slot_n = (int)s_list[s][valid_n ? idx_n : n - 1];
                    ga_n = s_geo[slot_n];
                    ha_n = s_geo2[slot_n];
                }
#else
                const int idx = t0 + g;
                const bool valid = idx < n;
                const int slot = (int)s_list[s][valid ? idx : n - 1];
                const f32x4 ga = s_geo[slot], ha = s_geo2[slot];
#endif
                const PairEval e = eval_poly(ga.xy, ga.zw, ha.x, ha.y, ha.z, uv);
                const bool hitp = valid && e.hit;
                const float om = hitp ? 1.f - e.alpha : 1.f;
                const float c = quad_chain(Tl[s], om, g);                   // T after this entry, in list order
                const float cb = quad_dpp<QP_SHIFT1, 0xf>(0.f, c);
                const float Tp = g == 0 ? Tl[s] : cb;                       // T before this entry
                const bool contrib = hitp && c >= kTMin;                    // (c < kTMin: this or an earlier entry ended the pixel)
                if (__builtin_amdgcn_ballot_w64(contrib) != 0) {
                    if constexpr (MASKS)
                        if (contrib) s_member[slot] = 1;
                    const float wgt = contrib ? e.alpha * Tp : 0.f;
                    const f32x2 w2 = {wgt, wgt};
                    const f32x4 f0 = s_feat4[slot * NF4];
                    C2[s][0] = __builtin_elementwise_fma(f0.xy, w2, C2[s][0]);
                    C2[s][1] = __builtin_elementwise_fma(f0.zw, w2, C2[s][1]);
#pragma unroll
                    for (int i = 0; i < S4; i++) {
                        const f32x4 f = s_feat4[slot * NF4 + 1 + i];
                        Cs2[s][2 * i] = __builtin_elementwise_fma(f.xy, w2, Cs2[s][2 * i]);
                        Cs2[s][2 * i + 1] = __builtin_elementwise_fma(f.zw, w2, Cs2[s][2 * i + 1]);
                    }
                    Tacc[s] = contrib ? c : Tacc[s];
                    lastc[s] = contrib ? pos0 + (uint32_t)slot : lastc[s];
                }
                const float c3 = quad_dpp<QP_BCAST3, 0xf>(0.f, c);
                Tl[s] = c3 >= kTMin ? c3 : 0.0f;
                if (live_mask(s) == 0) break;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (MASKS) {
            const unsigned long long members = __builtin_amdgcn_ballot_w64(s_member[lane] != 0);
            if (lane == 0) *member_mask_ptr(qmask0, qmask, tile_u, q_u, x0_u, b) = members;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- the four lanes of a pixel meet: channel sums, the last accepted transmittance, the last contributor
    int qc = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        float T = Tacc[s];
        T = fminf(T, quad_dpp<QP_SWAP1, 0xf>(0.f, T));
        T = fminf(T, quad_dpp<QP_SWAP2, 0xf>(0.f, T));
        int lc = (int)lastc[s];
        lc = max(lc, quad_dpp_i<QP_SWAP1>(lc));
        lc = max(lc, quad_dpp_i<QP_SWAP2>(lc));
        qc = max(qc, lc);
        const float r = quad_sum(C2[s][0].x), gr = quad_sum(C2[s][0].y), bl = quad_sum(C2[s][1].x), dp = quad_sum(C2[s][1].y);
        float sem[NSEM];
#pragma unroll
        for (int ch = 0; ch < NSEM; ch++) sem[ch] = quad_sum(Cs2[s][ch >> 1][ch & 1]);
        const int px = qx0 + (s & 1) * 4 + lx, py = qy0 + (s >> 1) * 4 + ly;
        if (g == 0 && px < W && py < H) {
            const size_t pix_id = (size_t)W * py + px;
            n_contrib[pix_id] = (uint32_t)lc;
            out_color[0 * HW + pix_id] = r + T * bg[0];
            out_color[1 * HW + pix_id] = gr + T * bg[1];
            out_color[2 * HW + pix_id] = bl + T * bg[2];
#pragma unroll
            for (int ch = 0; ch < NSEM; ch++)
                if (ch < S) out_sem[ch * HW + pix_id] = sem[ch];
            out_alpha[pix_id] = 1.f - T;
            out_depth[pix_id] = dp;
        }
    }
    qc = wave_max_i32(qc);
    if (qcost && lane == 0) qcost[tq] = (uint32_t)qc;
}

template <int S4>
void launch_g4_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list, float* out_color,
                  float* out_sem, float* out_depth, float* out_alpha, hipStream_t s, unsigned long long* qmask,
                  uint32_t* host_words, uint32_t stamp) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
    if (qmask)
        render_fwd_g4_k<S4, true><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(
            im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem, out_depth,
            out_alpha, im.n_contrib, im.qcost, im.qmask0, qmask, g.counters + COUNTER_OVF, host_words, stamp);
    else
        render_fwd_g4_k<S4, false><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(
            im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem, out_depth,
            out_alpha, im.n_contrib, im.qcost, im.qmask0, qmask, g.counters + COUNTER_OVF, host_words, stamp);
}

}  // namespace

void launch_render_fwd_g4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                          float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s,
                          unsigned long long* qmask, uint32_t* host_words, uint32_t stamp) {
#define GOI_CALL(N) launch_g4_s4<N>(sc, g, im, point_list, out_color, out_sem, out_depth, out_alpha, s, qmask, host_words, stamp)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

}  // namespace goi
