// Forward tile blend and trace for gfx950: front-to-back compositing of RGB + S-dim semantic feature
// + depth + alpha.
//
// Behaviour follows the reference's renderCUDA (cuda_rasterizer/forward.cu:261-386) and traceCUDA
// (forward.cu:422-551): per pixel the tile's depth-sorted list is walked front to back with the same
// guards (power > 0, alpha < 1/255, T(1-alpha) < 1e-4, 0.99 clamp) and the same outputs; tile
// membership stays 16x16 (it is part of the numerical contract).  The execution design is this
// library's own, for CDNA4:
//   * ONE WAVE = ONE WORKGROUP = one 8x8 pixel quadrant of a tile, one pixel per lane.  There are no
//     workgroup barriers and no cross-wave load imbalance: a quadrant that saturates early retires
//     its wave at once (the reference's 256-thread block waits for its slowest pixel);
//   * the wave walks the tile list in batches of 64: every lane fetches one Gaussian's position and
//     exact contribution box (GaussRec hx/hy), tests it against the quadrant, and only the lanes
//     that hit fetch the rest of the record and the S-float semantic row and park them in LDS; the
//     64-bit hit mask stays in SGPRs and is walked with a scalar bit scan, so a Gaussian that
//     provably cannot reach alpha >= 1/255 anywhere in the quadrant costs nothing in the pair loop;
//   * the pair loop issues only wave-uniform (broadcast, conflict-free) LDS reads;
//   * workgroup ids are remapped so that the four quadrants of a tile, and neighbouring tiles, run
//     on the same XCD and share its L2 (blocks are dealt round-robin to the 8 XCDs).
#include "blend_common.h"

namespace goi {

namespace {

// LEARN: the speculative depth cut-off's per-tile learning / checking (opt-in; a template parameter because the pixel's stop
// position costs the default kernel its fifth wave per SIMD: 82 -> 99 VGPRs)
// Two EXPERIMENTS on the channel sums C += w f of the pair loop (10 v_pk_fma_f32 + 5 broadcast ds_read_b128 of the 36 vector
// instructions per hit pair), both built, parity-tested (tests/test_gpu_parity.py) and MEASURED SLOWER than the packed FMAs
// (profiles/r06_blend_bounds.txt): a timing build with the sums removed altogether runs in 184 us instead of
// 284, so that is all there is to win, and the kernel's time follows its vector-pipe cycles to within 2 %:
//   SFEAT (fwd_variant 3): the row of a contributing Gaussian (80 bytes, the same for all 64 lanes) never enters LDS: the wave
//     fetches it with SCALAR loads (s_load_dwordx4 / x16; the Gaussian's id by v_readlane from its staging lane) one pair ahead
//     into one of two SGPR sets, and the packed FMAs take the feature pair as their scalar operand.  Bit-identical images.
//     308 us against 286: a scalar load's result can only be waited for with lgkmcnt(0), i.e. together with everything else in
//     flight, so one pair of look-ahead is all two register sets give, and an L2 round trip is longer than a pair.
//   OUTER (fwd_variant 4): C[pixel][channel] += w[pixel] f[channel] is a rank-one update, and v_mfma_f32_32x32x1_2b_f32 is two
//     32 x 32 outer products of fp32 vectors (exact fp32 FMAs): block b = the quadrant's pixels 32 b .. 32 b + 31, a lane supplies
//     its own pixel's weight as A and ONE feature channel (lane % 32; a plain ds_read_b32) as B; the 32 accumulators live in AGPRs
//     in the matrix layout (register 16 b + r of lane l: pixel 32 b + 8 (r / 4) + 4 (l / 32) + r % 4, channel l % 32 --
//     tools/probes/mfma_outer_probe.hip) and are transposed through LDS once, at the end of the list.  Decisions and gradients
//     bit-identical, maps equal to the last bit or two.  328 us against 284: the instruction is 16 passes = 64 clocks of a pipe
//     that fp32 matrix work SHARES with the vector instructions (tools/mfma_mix_probe.hip), against 40 for the ten packed FMAs it
//     replaces: (26 x 4 + 64) / (36 x 4) = 1.17, measured 1.16.
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int S4, bool TRACE, bool UNROLL2, bool MASKS, bool LEARN = false, bool SFEAT = false, bool OUTER = false>
__global__ __launch_bounds__(64) void render_fwd_k(const uint2* __restrict__ ranges,
                                                   const uint32_t* __restrict__ point_list, int W, int H, int gx,
                                                   int n_quads, int S, const GaussRec* __restrict__ rec,
                                                   const float* __restrict__ semantics, const float* __restrict__ bg,
                                                   float* __restrict__ out_color, float* __restrict__ out_sem,
                                                   float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                   uint32_t* __restrict__ n_contrib, const float* __restrict__ img_sem,
                                                   float* __restrict__ gau_sem, int* __restrict__ num_gsem,
                                                   uint32_t* __restrict__ qcost, unsigned long long* __restrict__ qmask0,
                                                   unsigned long long* __restrict__ qmask, const float* __restrict__ zcut,
                                                   uint32_t* __restrict__ zlearn, uint32_t* __restrict__ frame_flags,
                                                   uint32_t* __restrict__ host_words, uint32_t stamp) {
    static_assert(!(SFEAT && (TRACE || UNROLL2)), "scalar features: the plain pair loop only");
    constexpr int NF4 = TRACE ? 1 : 1 + S4;  // float4 words staged per Gaussian: (r,g,b,depth) + semantics
    constexpr int NSEM = TRACE ? 0 : 4 * S4;
    __shared__ f32x4 s_geo[64];   // (A3, A5, A1, A2) of the quadrant-centred log2-alpha polynomial (blend_common.h)
    __shared__ f32x4 s_geo2[64];  // (A0, A4, lim, -)
    static_assert(!(OUTER && (TRACE || SFEAT || 4 * NF4 > 32)), "outer-product accumulation: at most 32 channels, staged rows");
    constexpr int NCH = 4 * NF4;                 // channels of a staged row: r, g, b, depth, semantics
    constexpr int OSTR = NCH + 1;                // (OUTER) row stride of the final transposition [pixel][channel]
    constexpr int FEAT4 = SFEAT ? 1 : (OUTER ? (64 * OSTR + 3) / 4 + 8 : 64 * NF4);  // (OUTER: lanes read up to 32 floats of a 20-float row)
    __shared__ float4 s_feat[FEAT4];
    const f32x4* s_feat4 = reinterpret_cast<const f32x4*>(s_feat);
    __shared__ uint32_t s_id[TRACE ? 64 : 1];

    // the frame's counters for the host (speculative forward): final before this kernel starts (frame_flags - COUNTER_OVF is
    // the counters array); 32 words into pinned, device-mapped memory -- visible to the host once the kernel has completed
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
    const float QCX = t.QX0 + 3.5f, QCY = t.QY0 + 3.5f;  // quadrant centre
    const f32x2 uv = {t.pxf - QCX, t.pyf - QCY};         // this lane's pixel, quadrant-centred
    const int len = (int)(range.y - range.x);
    const int rounds = (len + 63) / 64;
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    const int lane = t.lane;

    float T = 1.0f;                         // transmittance that is written out (frozen once the pixel is done)
    float T_live = t.inside ? 1.0f : 0.0f;  // == T while the pixel is live, 0 once it is done: a finished pixel fails
                                            // the T(1-alpha) >= 1e-4 test by itself, no separate flag to test
    auto all_done = [&]() { return __builtin_amdgcn_ballot_w64(T_live != 0.0f) == 0; };
    uint32_t last_contributor = 0;
    uint32_t stop_pos = 0;  // 1-based list position of the entry that ENDED this pixel (T (1 - alpha) < 1e-4); 0: still live
    // accumulators as register pairs: one v_pk_fma_f32 adds two channels (this TU is compiled with the SLP
    // vectoriser off -- it packs the alpha evaluations of two candidates at the price of six moves -- so the
    // packing is spelled out)
    f32x2 C2[2] = {{0.f, 0.f}, {0.f, 0.f}};  // (r, g), (b, depth)
    f32x2 Cs2[NSEM > 0 ? NSEM / 2 : 1];
#pragma unroll
    for (int i = 0; i < NSEM / 2; i++) Cs2[i] = f32x2{0.f, 0.f};
    f32x32 acc;  // (OUTER) register 16 b + r of lane l: pixel 32 b + 8 (r / 4) + 4 (l / 32) + r % 4, channel l % 32
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = 0.f;
    const float* s_featf = reinterpret_cast<const float*>(s_feat) + (lane & 31);  // (OUTER) this lane's channel of a staged row

    // software prefetch of the next batch's id / position / box (one Gaussian per lane)
    uint32_t id_n = 0;
    // q0, q1: what the hit test reads (position, conic, opacity, box); q2 (r, g, b, depth): only a hit needs it, but fetched at
    // staging time -- a dependent 16-byte gather every round begins by waiting for -- it cost 11 of the kernel's 280 us
    float4 q0_n = make_float4(0, 0, 0, 0), q1_n = make_float4(1.f, 0.f, -1.f, -1.f), q2_n = make_float4(0, 0, 0, 0);
    auto prefetch = [&](int b) {
        const int k = b * 64 + lane;
        q1_n.z = -1.f;
        if (k < len) {
            id_n = point_list[range.x + k];
            const float4* r4 = reinterpret_cast<const float4*>(rec + id_n);
            q0_n = r4[0];
            q1_n = r4[1];
            if constexpr (!SFEAT) q2_n = r4[2];
        }
    };
    if (rounds > 0) prefetch(0);

    // MEMBER mask of the round (bit j: the Gaussian at list position 64 b + j contributed to some pixel of this quadrant),
    // in SGPRs; left for the backward blend (member_mask_ptr), which then neither tests a candidate against the quadrant
    // nor evaluates a pair that cannot contribute: a (position, quadrant) pair is a member here iff some pixel has it
    // below its last contributor and passes the two alpha guards there -- exactly the backward's own condition
    // (everything the word's address is made of is wave-uniform: kept in SGPRs, or the compiler carries tile / quadrant /
    // list start and the mask itself in vector registers -- 82 -> 114 VGPRs, four waves per SIMD instead of five)
    const int tile_u = __builtin_amdgcn_readfirstlane(t.tile), q_u = __builtin_amdgcn_readfirstlane(t.q);
    const uint32_t x0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);
    auto store_members = [&](int b, unsigned long long mm) {
        if constexpr (MASKS)
            if (lane == 0) *member_mask_ptr(qmask0, qmask, tile_u, q_u, x0_u, b) = mm;
    };
    int b_end = rounds;  // rounds of 64 list positions this wave looked at
    for (int b = 0; b < rounds; b++) {
        if (all_done()) {
            b_end = b;
            break;
        }
        unsigned long long members = 0;
        const uint32_t id = id_n;
        const float4 q0 = q0_n, q1 = q1_n, q2 = q2_n;
        const bool hit = ellipse_hits_quadrant(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, t.QX0, t.QY0);  // hx < 0 for absent lanes
        if (b + 1 < rounds) prefetch(b + 1);
        unsigned long long m = __ballot(hit);
        // (a round without a hit falls through to the store of its -- empty -- member mask: a second store site in an
        // early `continue` cost 32 VGPRs and the fifth wave per SIMD)
        // ---- stage the hits (slot = lane)
        if (m != 0 && hit) {
            const PolyCoef pc = poly_coefs(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, QCX, QCY);
            s_geo[lane] = f32x4{pc.A35.x, pc.A35.y, pc.A12.x, pc.A12.y};
            s_geo2[lane] = f32x4{pc.A0, pc.A4, pc.lim, 0.f};
            if constexpr (!SFEAT) s_feat[lane * NF4] = q2;  // r, g, b, depth
            if constexpr (SFEAT) {
            } else if constexpr (TRACE) {
                s_id[lane] = id;
            } else {
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
            }
        }
        __builtin_amdgcn_wave_barrier();  // single-wave workgroup: LDS is in order, only the compiler must not reorder

        // ---- pair loop over the hits, front to back
        // applies one candidate to the per-pixel state (sequential part) and accumulates its features
        auto apply = [&](int j, const PairEval& e, bool valid) {
            const float test_T = T_live * (1.f - e.alpha);
            const bool c0 = valid && e.hit;
            const bool ok = test_T >= kTMin;  // CR/forward.cu:352-356: a failing contributor ends the pixel
            const bool c = c0 && ok;
            const bool some = any_all(e.below, e.seen, ok);  // (taken where the comparisons are: their masks are used as they are)
            if (valid && some) {                             // (valid is wave-uniform)
                if constexpr (MASKS) asm("s_bitset1_b64 %0, %1" : "+s"(members) : "s"(j));  // members |= 1 << j, pinned to SGPRs
                const float wgt = c ? e.alpha * T_live : 0.f;
                const f32x2 w2 = {wgt, wgt};
                if constexpr (OUTER) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x1f32(wgt, s_featf[j * NCH], acc, 0, 0, 0);
                } else {
                const f32x4 f0 = s_feat4[j * NF4];
                C2[0] = __builtin_elementwise_fma(f0.xy, w2, C2[0]);
                C2[1] = __builtin_elementwise_fma(f0.zw, w2, C2[1]);
                if constexpr (!TRACE) {
#pragma unroll
                    for (int i = 0; i < S4; i++) {
                        const f32x4 f = s_feat4[j * NF4 + 1 + i];
                        Cs2[2 * i] = __builtin_elementwise_fma(f.xy, w2, Cs2[2 * i]);
                        Cs2[2 * i + 1] = __builtin_elementwise_fma(f.zw, w2, Cs2[2 * i + 1]);
                    }
                } else {
                    if (c && (double)e.alpha > 0.005) {
                        const uint32_t gid = s_id[j];
                        for (int ch = 0; ch < S; ch++)
                            atomicAdd(&gau_sem[(size_t)gid * S + ch], img_sem[ch * HW + pix_id]);
                        atomicAdd(&num_gsem[gid], S);
                    }
                }
                }  // (!OUTER)
                if (c) {
                    T = test_T;
                    last_contributor = (uint32_t)(b * 64 + j + 1);
                }
            }
            if constexpr (LEARN)
                if (c0 && !ok) stop_pos = (uint32_t)(b * 64 + j + 1);  // (the depth cut-off must keep this entry)
            T_live = c0 ? (ok ? test_T : 0.0f) : T_live;
        };
        if constexpr (SFEAT) {
            // software pipeline, one pair ahead: the coefficients (LDS) and the feature row (scalar loads) of the NEXT hit are
            // requested before the current one is evaluated; one wait per trip (scalar loads return out of order: any wait for
            // them is a wait for all, so nothing else may be requested between a wait and its uses)
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            static_assert(S4 == 4, "scalar features: S = 16 (one s_load_dwordx16 per row)");
            struct Feat {
                f32x4 c;     // r, g, b, depth
                f32x16 s;    // semantic row
            };
            // (inline assembly: the compiler sinks scalar loads it knows about to just in front of their wait -- and then nothing
            // overlaps; requested here, they are in flight while the current pair is evaluated.  The wait below is the builtin, which
            // the compiler's own bookkeeping of the LDS reads understands.)
            auto request = [&](int jj, Feat& F) {
                const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id, jj);  // the staging lane's Gaussian
                const GaussRec* pr = rec + gid;
                const float* ps = semantics + (size_t)gid * 16;
                asm volatile("s_load_dwordx4 %0, %1, 0x20" : "=&s"(F.c) : "s"(pr));
                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(F.s) : "s"(ps));
            };
            auto arrived = [&](Feat& F) {
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                asm volatile("" : "+s"(F.c), "+s"(F.s));  // (the uses below stay below)
            };
            auto accumulate = [&](int j, const PairEval& e, const Feat& F) {
                const float test_T = T_live * (1.f - e.alpha);
                const bool c0 = e.hit;
                const bool ok = test_T >= kTMin;
                const bool c = c0 && ok;
                const bool some = any_all(e.below, e.seen, ok);
                if (some) {
                    if constexpr (MASKS) asm("s_bitset1_b64 %0, %1" : "+s"(members) : "s"(j));
                    const float wgt = c ? e.alpha * T_live : 0.f;
                    const f32x2 w2 = {wgt, wgt};
                    C2[0] = __builtin_elementwise_fma(F.c.xy, w2, C2[0]);
                    C2[1] = __builtin_elementwise_fma(F.c.zw, w2, C2[1]);
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        Cs2[i] = __builtin_elementwise_fma(f32x2{F.s[2 * i], F.s[2 * i + 1]}, w2, Cs2[i]);
                    if (c) {
                        T = test_T;
                        last_contributor = (uint32_t)(b * 64 + j + 1);
                    }
                }
                if constexpr (LEARN)
                    if (c0 && !ok) stop_pos = (uint32_t)(b * 64 + j + 1);
                T_live = c0 ? (ok ? test_T : 0.0f) : T_live;
            };
            if (m) {
                int j = __builtin_ctzll(m);
                Feat Fa, Fb;  // two register sets, used in turn (no copies at the end of a trip)
                f32x4 ga = s_geo[j], ga2 = s_geo2[j], gb, gb2;
                request(j, Fa);
                // one pair: wait for its row, request the next hit's row and coefficients into the OTHER set, evaluate; false = done
                auto trip = [&](Feat& F, const f32x4& g, const f32x4& g2, Feat& Fn, f32x4& gn, f32x4& gn2) {
                    arrived(F);
                    m &= m - 1;
                    const bool more = m != 0;
                    const int jn = more ? __builtin_ctzll(m) : j;
                    request(jn, Fn);
                    gn = s_geo[jn];
                    gn2 = s_geo2[jn];
                    const PairEval e = eval_poly(g.xy, g.zw, g2.x, g2.y, g2.z, uv);
                    accumulate(j, e, F);
                    j = jn;
                    return more && !all_done();
                };
                while (trip(Fa, ga, ga2, Fb, gb, gb2) && trip(Fb, gb, gb2, Fa, ga, ga2)) {
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  // (the last request is never used: it must not land in registers that live on)
            }
        } else if constexpr (UNROLL2) {
            // two candidates per trip: their alpha evaluations are independent (ILP, half the
            // branches); the state update stays strictly in list order
            while (m) {
                const int j0 = __builtin_ctzll(m);
                m &= m - 1;
                const bool has1 = m != 0;
                const int j1 = has1 ? __builtin_ctzll(m) : j0;
                if (has1) m &= m - 1;
                const f32x4 ga = s_geo[j0], gb = s_geo[j1];
                const f32x4 ha = s_geo2[j0], hb = s_geo2[j1];
                const PairEval e0 = eval_poly(ga.xy, ga.zw, ha.x, ha.y, ha.z, uv);
                const PairEval e1 = eval_poly(gb.xy, gb.zw, hb.x, hb.y, hb.z, uv);
                apply(j0, e0, true);
                apply(j1, e1, has1);
                if (all_done()) m = 0;
            }
        } else {
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const f32x4 g = s_geo[j];
                const f32x4 g2 = s_geo2[j];
                const PairEval e = eval_poly(g.xy, g.zw, g2.x, g2.y, g2.z, uv);
                apply(j, e, true);
                if (all_done()) m = 0;
            }
        }
        store_members(b, members);
        __builtin_amdgcn_wave_barrier();
    }

    // Speculative depth cut-off (api.hip, goi_raster_forward_async_cut).  LEARN: the view depth up to which this tile's list
    // is worth listing the next time this camera is rendered -- the depth of the entry an eighth (and 32 positions) beyond
    // the last entry that ended a pixel of the tile (max over its four quadrant waves); +inf when a pixel reached the end of its list
    // unsaturated (such a tile is never cut).  CHECK: a pixel that reaches the end of a list that WAS cut (zcut finite) without
    // saturating may have wanted what was dropped: the frame's flag is raised -- its backward writes zero gradients, the host
    // redoes or skips the view and forgets the camera's cut.
    if constexpr (LEARN) {
        {
            const bool unsat = !all_done();
            const float zc_in = zcut ? zcut[tile_u] : __builtin_inff();
            float zc = __builtin_inff();
            if (unsat) {
                if (zc_in < 3.0e38f && lane == 0) atomicOr(frame_flags, OVF_CUT_TOO_TIGHT);
            } else {
                // every pixel has stopped: what the tile needs of its list ends at the LAST stopping entry (a pixel that has
                // not met its stopping entry keeps walking: the cut must contain it); lanes outside the image never started
                int seen = min(len, wave_max_i32((int)stop_pos));
                (void)b_end;
                // A cut list is NOT a prefix of the uncut one: the cut lives in the ellipse tile masks, and a rectangle of
                // more than 64 tiles has none -- such a Gaussian stays listed at every depth.  A pixel that passes zcut
                // unsaturated and then stops on a deeper (big) entry has skipped the small Gaussians that were dropped in
                // between.  Up to zcut the list is exact, so the frame is exact iff no pixel looked beyond it: the deepest
                // entry any pixel of this wave looked at is its last stopping entry (the list is in depth order).
                if (zc_in < 3.0e38f && seen > 0) {
                    const float z_last = rec[point_list[range.x + (uint32_t)(seen - 1)]].q2.w;
                    if (z_last > zc_in && lane == 0) atomicOr(frame_flags, OVF_CUT_TOO_TIGHT);
                }
                const int pos = seen + (seen >> 3) + 32;  // margin: an eighth + 32 list positions
                if (pos < len)
                    zc = rec[point_list[range.x + (uint32_t)pos]].q2.w;  // (depth of that entry: the list is in depth order)
                else
                    zc = zc_in;  // the margin runs past the end of the list: keep the cut the list was built with (or none)
            }
            if (zlearn && lane == 0) atomicMax(&zlearn[tile_u], __float_as_uint(zc));  // (depths are positive: their bits order like uints)
        }
    }

    {   // how far the backward's wave of this quadrant has to walk: the largest last contributor of its 64 pixels
        const int qc = wave_max_i32((int)last_contributor);
        if (qcost && lane == 0) qcost[tq] = (uint32_t)qc;
    }
    if constexpr (OUTER) {
        // matrix layout -> one pixel per lane: [pixel][channel] through LDS (the staging area is free now), once per wave
        float* s_out = reinterpret_cast<float*>(s_feat);
        const int ch = lane & 31;
        __builtin_amdgcn_wave_barrier();
        if (ch < NCH) {
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int pix = 32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * (lane >> 5) + (r & 3);
                s_out[pix * OSTR + ch] = acc[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float* mine = s_out + lane * OSTR;
        C2[0] = f32x2{mine[0], mine[1]};
        C2[1] = f32x2{mine[2], mine[3]};
#pragma unroll
        for (int i = 0; i < NSEM / 2; i++) Cs2[i] = f32x2{mine[4 + 2 * i], mine[5 + 2 * i]};
    }
    if (t.inside) {
        n_contrib[pix_id] = last_contributor;
        out_color[0 * HW + pix_id] = C2[0].x + T * bg[0];
        out_color[1 * HW + pix_id] = C2[0].y + T * bg[1];
        out_color[2 * HW + pix_id] = C2[1].x + T * bg[2];
        if constexpr (!TRACE) {
#pragma unroll
            for (int ch = 0; ch < NSEM; ch++)
                if (ch < S) out_sem[ch * HW + pix_id] = Cs2[ch >> 1][ch & 1];
            out_alpha[pix_id] = 1.f - T;
            out_depth[pix_id] = C2[1].y;
        }
    }
}


template <int S4>
void launch_fwd_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                   float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s,
                   unsigned long long* qmask, const float* zcut, uint32_t* zlearn, uint32_t* host_words, uint32_t stamp) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
#define GOI_LAUNCH_FWD(U2, MK)                                                                                         \
    do {                                                                                                               \
        if (zlearn || zcut) /* a cut that is applied is always CHECKED, whether or not the frame learns a new one */    \
            render_fwd_k<S4, false, U2, MK, true><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(                        \
                im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem,   \
                out_depth, out_alpha, im.n_contrib, nullptr, nullptr, nullptr, im.qcost, im.qmask0, qmask, zcut,        \
                zlearn, g.counters + COUNTER_OVF, host_words, stamp);                                                   \
        else                                                                                                           \
            render_fwd_k<S4, false, U2, MK><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(                              \
                im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem,   \
                out_depth, out_alpha, im.n_contrib, nullptr, nullptr, nullptr, im.qcost, im.qmask0, qmask, zcut,        \
                zlearn, g.counters + COUNTER_OVF, host_words, stamp);                                                   \
    } while (0)
    // The member masks are recorded by EVERY forward (a backward may follow with either setting of bwd_masks, and a frame
    // without masks back-propagated through them would be garbage).  BUILD SWITCH for measuring what recording costs the
    // forward: GOI_EXTRA_FLAGS=-DGOI_FWD_NO_MASKS (such a build must run with GOI_OPTIONS=bwd_masks=0).
#ifdef GOI_FWD_NO_MASKS
    const bool masks = false;
#else
    const bool masks = qmask != nullptr;
#endif
    if constexpr (S4 <= 7) {
        // fwd_variant 4 (EXPERIMENT, measured negative: profiles/r06_blend_bounds.txt): outer-product accumulation on the fp32 matrix
        // instruction; frames with a depth cut keep the default kernel
        if (g_options.fwd_variant == 4 && masks && !zlearn && !zcut) {
            render_fwd_k<S4, false, true, true, false, false, true><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(
                im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem, out_depth,
                out_alpha, im.n_contrib, nullptr, nullptr, nullptr, im.qcost, im.qmask0, qmask, zcut, zlearn,
                g.counters + COUNTER_OVF, host_words, stamp);
            return;
        }
    }
    bool scalar_features = false;
    if constexpr (S4 == 4) scalar_features = g_options.fwd_variant == 3 && sc.S == 16 && masks && !zlearn && !zcut;
    if (scalar_features) {
        if constexpr (S4 == 4)
            render_fwd_k<S4, false, false, true, false, true><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(
                im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, sc.semantics, sc.bg, out_color, out_sem, out_depth,
                out_alpha, im.n_contrib, nullptr, nullptr, nullptr, im.qcost, im.qmask0, qmask, zcut, zlearn,
                g.counters + COUNTER_OVF, host_words, stamp);
    } else if (g_options.fwd_variant == 1 || g_options.fwd_variant >= 3) {
        if (masks) GOI_LAUNCH_FWD(true, true);
        else GOI_LAUNCH_FWD(true, false);
    } else {
        if (masks) GOI_LAUNCH_FWD(false, true);
        else GOI_LAUNCH_FWD(false, false);
    }
#undef GOI_LAUNCH_FWD
}

}  // namespace

void launch_render_fwd(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s,
                       unsigned long long* qmask, const float* zcut, uint32_t* zlearn, uint32_t* host_words, uint32_t stamp) {
    // fwd_variant 2 (EXPERIMENT, render_fwd_g4.hip): the 16 pixels x 4 Gaussians mapping; frames with a depth cut and S > 16 keep
    // the default kernel
    if (g_options.fwd_variant == 2 && !zcut && !zlearn && sc.S <= 16) {
        launch_render_fwd_g4(sc, g, im, point_list, out_color, out_sem, out_depth, out_alpha, s, qmask, host_words, stamp);
        return;
    }
#define GOI_CALL(N) launch_fwd_s4<N>(sc, g, im, point_list, out_color, out_sem, out_depth, out_alpha, s, qmask, zcut, zlearn, host_words, stamp)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

void launch_trace_fwd(const GoiRasterScene& sc, const float* img_sem, const GeomView& g, const ImageView& im,
                      const uint32_t* point_list, float* out_color, float* gau_sem, int* num_gsem, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
    render_fwd_k<1, true, false, false><<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(
        im.ranges, point_list, sc.W, sc.H, gx, n_quads, sc.S, g.rec, nullptr, sc.bg, out_color, nullptr, nullptr, nullptr,
        im.n_contrib, img_sem, gau_sem, num_gsem, nullptr, nullptr, nullptr, nullptr, nullptr, g.counters + COUNTER_OVF,
        nullptr, 0u);
}

}  // namespace goi
