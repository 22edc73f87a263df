// Row reduction of the atomic-free backward: sums the partial-gradient rows render_bwd_rows_k (render_bwd.hip) leaves per
// (emit-order instance, quadrant) into one record (or six per-id arrays) per Gaussian, in a fixed order.  Pure additions:
// compiled with the default flags (it used to sit in preprocess.hip under -ffp-contract=off for no reason).
// Reference: the float atomicAdd accumulation of cuda_rasterizer/backward.cu:565-621, which this replaces.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "row_sum.h"

namespace goi {

namespace {

// Sums the partial-gradient rows of every Gaussian (written by render_bwd_rows_k, one row per
// (emit-order instance, quadrant) slot, a Gaussian's slots contiguous) in a fixed order and writes
// the six blend-gradient arrays for ALL Gaussians (zeros where nothing contributed): no memsets, no
// atomics, bit-reproducible.  A quarter wave (16 lanes) owns one Gaussian.  Memory-level parallelism
// is what matters here: the lanes fetch 16 instances x 4 validity bytes in one load, the flagged
// slots of the chunk are packed into a 64-bit mask (quadrant-major), and up to 16 rows are requested
// back to back before the first is consumed (lane e reads row elements e, e+16, ...: coalesced).
// The kernel is LATENCY bound, not bandwidth bound: a Gaussian costs a chain of three dependent memory round trips
// (slot range -> validity bytes -> rows) for ~8 rows of payload, and with one Gaussian per quarter wave the chip works
// through 250 K short-lived waves in ~30 rounds of that chain (223 us for 0.5 GB).  So every quarter wave walks GPQ
// Gaussians in a software pipeline: while the rows of Gaussian k are summed, the validity word of k+1 and the slot
// range of k+2 are already on their way -- one exposed round trip per Gaussian instead of three, 1/GPQ of the waves.
// The order in which a Gaussian's rows are added is unchanged (bit-identical gradients).
// RECORD (the full backward): the sums do not leave as six per-Gaussian arrays at all.  A Gaussian's record -- its summed
// row, 128 bytes at S <= 16 -- goes back into the row scratch, over the first slot the Gaussian owns (every listed
// Gaussian owns at least four; its rows have all been read by then): ONE full-line store per Gaussian instead of six
// scattered partial ones, no zeros for the unlisted Gaussians (the old form wrote 104 bytes of them for each), and
// preprocess_bwd_k, which runs over the ids anyway, fetches the line through goff[] and writes every per-id output itself,
// coalesced.  Measured on the headline view before it was built (timing builds): the zero phase 21 us, the scattered
// stores 40 us of the kernel's 213; a dense 128-byte store instead 8 us.

// BIG Gaussians.  A quarter wave sums ITS Gaussian's rows one trip after the other; a frame-filling blob or a long needle
// of a reconstructed scene owns thousands of slots and ten thousand rows (clustered workload: 20 blobs x 6600 tiles, needles
// with whole-frame rectangles), and ONE quarter wave working through them kept this kernel running for 1.2 ms while the chip
// idled.  A Gaussian with more than BIG_INST instances is therefore not summed here: its quarter wave registers it (a
// descriptor in the scratch, slots handed out by an atomic counter), and reduce_big_k gives it a whole WORKGROUP: the quarter
// waves of the workgroup sum contiguous parts of its slot range (same walk, same order inside a part), the partial rows meet
// in LDS and are added in part order.  Two sizes (round 5): 16 parts (a quarter of the workgroup) for a Gaussian of up to
// HUGE_INST instances, 64 parts (the whole workgroup) beyond.  The threshold was 1024 whatever the frame: on a close-up of a
// 3 M scene (BASELINE config 5's shape: 18 instances per listed Gaussian on average, a long tail of several hundred)
// reduce_rows_k then took 315 us for 450 k rows -- the duration of its longest serial walks; with 384 it takes 67 (+ 28 for
// the big ones).  On the headline scene (8 instances per listed Gaussian) the few hundred Gaussians between 384 and 1024
// instances are NOT the tail of reduce_rows_k, and a workgroup of their own costs 10-20 us more than it saves: so the frame
// chooses -- BIG_INST when it holds more than REDUCE_DENSE_RATIO instances per listed Gaussian, 1024 otherwise, decided by every
// block from the frame's device counters (no host involvement; same-box A/B of both, tools/gpu_ab_k.sh, profiles/README.md).
// Which path a Gaussian takes depends on its instance count and those two frame counts alone and every order is fixed:
// bit-reproducible as before.  (The counters are zeroed with the validity bytes: quad_order_k's extra workgroups, or a
// memset.  A frame without big Gaussians pays one launch that finds nothing to do.)
constexpr uint32_t BIG_INST = REDUCE_BIG_INST;  // (common.h: the scratch layout sizes the descriptor list from it)
constexpr uint32_t HUGE_INST = REDUCE_HUGE_INST;

struct ReduceOut {
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dsemantic, *dL_ddepth;
};

// What a Gaussian's summed row turns into: its RECORD (back into the row scratch, over the first slot it owns), or the six
// per-id arrays.  Lane e of the quarter wave holds elements 2e, 2e+1 (K == 2) or e, e + 16, .. of the row.
template <int K, bool RECORD>
__device__ __forceinline__ void store_sums(const float (&sum)[K], float* rows, size_t inst0, uint32_t g, int e, int S, int nch,
                                           const ReduceOut& o) {
    constexpr int RF = 16 * K;
    if constexpr (RECORD) {
        float* dst = rows + inst0 * 4 * RF;  // (the row elements this lane summed, back where it read them)
        if (K == 2) {
            reinterpret_cast<float2*>(dst)[e] = make_float2(sum[0], sum[K - 1]);
        } else {
#pragma unroll
            for (int kk = 0; kk < K; kk++) dst[e + 16 * kk] = sum[kk];
        }
    } else {
        const int nsem = nch - 4;
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
            const float v = sum[kk];
            const int el = K == 2 ? 2 * e + kk : e + 16 * kk;  // element of the row this lane summed
            if (el < nsem) {
                if (el < S) o.dL_dsemantic[(size_t)g * S + el] = v;
            } else if (el < nsem + 3) {
                o.dL_dcolor[(size_t)g * 3 + (el - nsem)] = v;
            } else if (el == nsem + 3) {
                o.dL_ddepth[g] = v;
            } else if (el < nch + 2) {
                o.dL_dmean2D[(size_t)g * 3 + (el - nch)] = v;
                if (el == nch + 1) o.dL_dmean2D[(size_t)g * 3 + 2] = 0.f;
            } else if (el < nch + 5) {
                const int c = el - nch - 2;  // a, b, c -> x, y, w of the [P,2,2] conic gradient
                o.dL_dconic[(size_t)g * 4 + (c == 2 ? 3 : c)] = v;
                if (c == 2) o.dL_dconic[(size_t)g * 4 + 2] = 0.f;
            } else if (el == nch + 5) {
                o.dL_dopacity[g] = v;
            }
        }
    }
}

// INFLIGHT: rows a quarter wave requests back to back (its registers: 2 K per row).  32 is best up to ~2 M Gaussians; on larger
// scenes (a 12.9 GB slot space at 3 M: every round trip longer) 16 -- fewer registers, more resident waves -- is, and the launcher
// picks by P (same-box A/B, profiles/r06_tail.txt: 3 M 282 -> 250 us and -> 227 with one Gaussian per quarter wave, 1 M 153 -> 183 us
// the other way round).  Same sums.
template <int K, int GPQ, bool RECORD, int INFLIGHT = GOI_REDUCE_INFLIGHT>  // K = row_floats / 16; GPQ = Gaussians per quarter wave
__global__ __launch_bounds__(256) void reduce_rows_k(int P, int S, int nch, uint32_t N_cap, const uint32_t* __restrict__ n_dev,
                                                     const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ tiles_touched,
                                                     float* rows, const uint8_t* __restrict__ flags, ReduceOut out,
                                                     uint32_t* __restrict__ big_ctl, uint4* __restrict__ big_desc,
                                                     uint32_t cap_big) {
    // N_cap: the slot capacity the scratch was laid out for; n_dev: the forward's instance count on the device (the
    // exact forward passes N_cap = num_rendered; the speculative one a capacity, and an overflowed frame stored only
    // the first N_cap instances)
    // a TRUNCATED frame (COUNTER_OVF, set by emit) has no valid rows: every Gaussian gets zeros
    const bool truncated = n_dev[COUNTER_OVF - COUNTER_N] != 0;
    const uint32_t N = truncated ? 0u : min(N_cap, *n_dev);
    const int V = truncated ? 0 : (int)n_dev[COUNTER_V - COUNTER_N];  // listed Gaussians: the only ones that own rows
    const int lane = threadIdx.x & 63, quarter = lane >> 4, e = lane & 15;
    const uint32_t* flags32 = reinterpret_cast<const uint32_t*>(flags);
    // ---- phase 0 (the first ceil(P/256) workgroups): zeros for the Gaussians that are NOT listed (culled, or a culled
    // rectangle without tiles; all of them for a truncated frame) -- one Gaussian per lane.  The listed ones are written
    // by phase 1 below, so every element of the six arrays is written exactly once.
    if constexpr (!RECORD) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < P && (truncated || tiles_touched[i] == 0)) {
            if ((S & 3) == 0) {
                float4* d4 = reinterpret_cast<float4*>(out.dL_dsemantic + (size_t)i * S);
                for (int ch = 0; ch < S / 4; ch++) d4[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int ch = 0; ch < S; ch++) out.dL_dsemantic[(size_t)i * S + ch] = 0.f;
            }
            if (out.dL_dopacity) {  // (NULL in the feature-gradient-only reduction)
                out.dL_dopacity[i] = 0.f;
                out.dL_ddepth[i] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) out.dL_dcolor[(size_t)i * 3 + k] = out.dL_dmean2D[(size_t)i * 3 + k] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) out.dL_dconic[(size_t)i * 4 + k] = 0.f;
            }
        }
    }
    // ---- phase 1: the V LISTED Gaussians in DEPTH order.  That is the order of the slot space (emit order), so
    // consecutive quarter waves stream through rows[] and flags[] front to back -- and it is the order in which the valid
    // rows are DENSE: near Gaussians contribute in most of their tiles, far ones are behind the saturation front and own
    // hardly any row, so the rows that exist sit close together at the front of the slot space (DRAM pages, TLB).  A
    // slot space in id order (tried: the outputs then leave as neighbouring lines instead of six scattered partial
    // stores per Gaussian) spreads the same rows evenly over 2.6 GB and costs more than the scatter saves (0.235 ->
    // 0.32 ms).  Step k of the block covers 16 consecutive listed Gaussians.
    const int i0 = blockIdx.x * (16 * GPQ) + (threadIdx.x >> 4);
    if (blockIdx.x * (16 * GPQ) >= V) return;  // (block-uniform)
    struct Meta {
        uint32_t g, off0, off1;
    };
    // slots of the i-th listed Gaussian in depth order: [offsets[i], offsets[i+1]) -- straight from the prefix sum; the
    // Gaussian's id is only needed for the final store
    auto load_meta = [&](int k) {
        const int i = i0 + 16 * k;
        Meta m{0u, 0u, 0u};
        if (k < GPQ && i < V) {
            m.g = order[i];
            m.off0 = min(offsets[i], N);
            m.off1 = i + 1 < V ? min(offsets[i + 1], N) : N;
        }
        return m;
    };
    // the 4 quadrant bytes of instance off0 + e (the first chunk of a Gaussian; a BIG one is not walked here)
    // this frame's threshold (see "BIG Gaussians" above)
    const uint32_t big_inst = (N > REDUCE_DENSE_RATIO * (uint32_t)V) ? BIG_INST : 1024u;
    auto load_first = [&](const Meta& m) {
        const uint32_t c = m.off1 - m.off0;
        return (c <= big_inst && (uint32_t)e < c) ? flags32[m.off0 + e] : 0u;
    };

    Meta cur = load_meta(0), nxt = load_meta(1);
    uint32_t w_cur = load_first(cur);
#pragma unroll 1
    for (int k = 0; k < GPQ; k++) {
        if (i0 + 16 * k - (int)(threadIdx.x >> 4) >= V) break;  // (block-uniform: nothing left for any quarter wave)
        const Meta nn = load_meta(k + 2);        // two Gaussians ahead: slot range
        const uint32_t w_nxt = load_first(nxt);  // one ahead: validity bytes of its first 16 instances
        const bool live = i0 + 16 * k < V;
        const uint32_t cnt_all = cur.off1 - cur.off0;
        const bool big = cnt_all > big_inst;
        const size_t inst0 = cur.off0;
        if (big && e == 0) {  // hand the Gaussian over to reduce_big_k: huge ones from the front of the list, the others from its end
            const uint4 d = make_uint4((uint32_t)inst0, cnt_all, 0u, cur.g);
            if (cnt_all > HUGE_INST) big_desc[atomicAdd(&big_ctl[1], 1u)] = d;
            else big_desc[cap_big - 1u - atomicAdd(&big_ctl[2], 1u)] = d;
        }
        const uint32_t cnt = big ? 0u : cnt_all;
        float sum[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) sum[kk] = 0.f;
        sum_instances<K, false, INFLIGHT>(rows, flags32, inst0, cnt, w_cur, quarter, e, sum, sum);  // (comp unused)
        if (live && !big && (!RECORD || cnt > 0)) store_sums<K, RECORD>(sum, rows, inst0, cur.g, e, S, nch, out);
        cur = nxt;
        nxt = nn;
        w_cur = w_nxt;
    }
}

// bwd_records 2 (the per-Gaussian backward sums its Gaussians' rows itself: preprocess.hip): nobody walks the listed Gaussians
// in here any more, so the BIG ones are registered by this kernel -- one thread per listed Gaussian in depth order, the same
// threshold from the same frame counters, the same descriptors -- and reduce_big_k leaves their records in the row scratch as
// before.  (The order of the descriptors differs from run to run; every Gaussian's sum is its own: bit-reproducible.)
__global__ __launch_bounds__(256) void find_big_k(uint32_t N_cap, const uint32_t* __restrict__ n_dev,
                                                  const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                  uint32_t* __restrict__ big_ctl, uint4* __restrict__ big_desc, uint32_t cap_big) {
    const bool truncated = n_dev[COUNTER_OVF - COUNTER_N] != 0;
    const uint32_t N = truncated ? 0u : min(N_cap, *n_dev);
    const int V = truncated ? 0 : (int)n_dev[COUNTER_V - COUNTER_N];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const uint32_t off0 = min(offsets[i], N), off1 = i + 1 < V ? min(offsets[i + 1], N) : N;
    const uint32_t cnt = off1 - off0;
    const uint32_t big_inst = (N > REDUCE_DENSE_RATIO * (uint32_t)V) ? BIG_INST : 1024u;
    if (cnt > big_inst) {
        const uint4 d = make_uint4(off0, cnt, 0u, order[i]);
        if (cnt > HUGE_INST) big_desc[atomicAdd(&big_ctl[1], 1u)] = d;
        else big_desc[cap_big - 1u - atomicAdd(&big_ctl[2], 1u)] = d;
    }
}

// The big Gaussians (persistent workgroups of 1024 threads: the grid is fixed, the counts live on the device).  First the ones of
// up to HUGE_INST instances, FOUR at a time: each quarter of the workgroup (16 quarter waves = 16 parts) takes one; then the
// huge ones, the whole workgroup (64 parts) each.  Quarter wave p sums part p of its Gaussian's instances -- contiguous, a
// multiple of 64 instances long -- and the first quarter wave of the group adds the partial rows in part order and writes what
// reduce_rows_k writes for a Gaussian.  (With 16 parts for a frame-filling blob the kernel's time was that of its largest
// Gaussian: 6 600 instances were 400 per quarter wave, one row trip after the other.)
constexpr int BIG_PARTS = 64, MID_PARTS = 16;
template <int K, bool RECORD>
__global__ __launch_bounds__(16 * BIG_PARTS) void reduce_big_k(const uint32_t* __restrict__ big_ctl,
                                                              const uint4* __restrict__ big_desc, uint32_t cap_big, float* rows,
                                                              const uint8_t* __restrict__ flags, int S, int nch, ReduceOut out) {
    constexpr int RF = 16 * K;
    __shared__ float s_part[BIG_PARTS][RF];
    const int lane = threadIdx.x & 63, quarter = lane >> 4, e = lane & 15, part = threadIdx.x >> 4;
    const uint32_t* flags32 = reinterpret_cast<const uint32_t*>(flags);
    // PARTS quarter waves starting at quarter wave `first` sum the Gaussian of descriptor d (active: the group has one)
    auto one = [&](const uint4 d, bool active, int parts, int first) {
        const int p = part - first;  // this quarter wave's part
        const uint32_t n = active ? d.y : 0u;
        const uint32_t per = ((n + (uint32_t)parts - 1u) / (uint32_t)parts + 63u) & ~63u;  // instances per part
        const uint32_t p0 = min(n, (uint32_t)p * per), p1 = min(n, (uint32_t)(p + 1) * per);
        const uint32_t cnt = p1 - p0;
        const size_t inst0 = (size_t)d.x + p0;
        const uint32_t w0 = ((uint32_t)e < cnt) ? flags32[inst0 + e] : 0u;
        float sum[K], comp[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) sum[kk] = comp[kk] = 0.f;
        sum_instances<K, true>(rows, flags32, inst0, cnt, w0, quarter, e, sum, comp);
#pragma unroll
        for (int kk = 0; kk < K; kk++) s_part[part][K == 2 ? 2 * e + kk : e + 16 * kk] = sum[kk];
        __syncthreads();
        if (p == 0 && active) {
            float tot[K], c[K];
#pragma unroll
            for (int kk = 0; kk < K; kk++) tot[kk] = c[kk] = 0.f;
            for (int pp = 0; pp < parts; pp++)
#pragma unroll
                for (int kk = 0; kk < K; kk++) {  // (compensated as well: the parts' sums cancel like their rows do)
                    const float y = s_part[first + pp][K == 2 ? 2 * e + kk : e + 16 * kk] - c[kk];
                    const float t = tot[kk] + y;
                    c[kk] = (t - tot[kk]) - y;
                    tot[kk] = t;
                }
            store_sums<K, RECORD>(tot, rows, (size_t)d.x, d.w, e, S, nch, out);
        }
        __syncthreads();
    };
    const uint32_t nmid = big_ctl[2], nhuge = big_ctl[1];
    const int sub = part / MID_PARTS;  // which quarter of the workgroup
    for (uint32_t b0 = blockIdx.x * 4u; b0 < nmid; b0 += gridDim.x * 4u) {  // (block-uniform trip count)
        const uint32_t b = b0 + (uint32_t)sub;
        const bool active = b < nmid;
        const uint4 d = active ? big_desc[cap_big - 1u - b] : make_uint4(0u, 0u, 0u, 0u);  // (first instance, instances, -, id)
        one(d, active, MID_PARTS, sub * MID_PARTS);
    }
    for (uint32_t b = blockIdx.x; b < nhuge; b += gridDim.x) one(big_desc[b], true, BIG_PARTS, 0);
}

}  // namespace

#ifndef GOI_REDUCE_GPQ
#define GOI_REDUCE_GPQ 2
#endif
constexpr int REDUCE_GPQ = GOI_REDUCE_GPQ;  // Gaussians per quarter wave of reduce_rows_k
#ifndef GOI_REDUCE_BIG_GRID
#define GOI_REDUCE_BIG_GRID 512
#endif
// persistent workgroups of reduce_big_k: two 1024-thread workgroups per CU (the kernel waits on memory: with one 256-thread
// workgroup per CU it ran the clustered workload's big Gaussians at an eighth of the memory-level parallelism the chip has)
constexpr size_t REDUCE_BIG_GRID = GOI_REDUCE_BIG_GRID;

#ifndef GOI_REDUCE_LARGE_SCENE
#define GOI_REDUCE_LARGE_SCENE 2000000
#endif
constexpr int REDUCE_LARGE_SCENE = GOI_REDUCE_LARGE_SCENE;  // Gaussians from which reduce_rows_k keeps 16 instead of 32 rows in flight
template <int K, bool RECORD>
static void launch_reduce_k(const GoiRasterScene& sc, const GeomView& g, int N, int nch, float* rows, const uint8_t* flags,
                            const BwdScratchView& scr, const ReduceOut& out, hipStream_t s) {
    const dim3 grid((sc.P + 16 * REDUCE_GPQ - 1) / (16 * REDUCE_GPQ));
    const uint32_t* order = g.sort_vals[depth_sort_result_index()];
    if (sc.P >= REDUCE_LARGE_SCENE && GOI_REDUCE_INFLIGHT > 16)  // (one Gaussian per quarter wave there as well: 3 M 250 -> 227 us, 6 M 420 -> 377)
        reduce_rows_k<K, 1, RECORD, 16><<<dim3((sc.P + 15) / 16), dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N,
                                                                                    order, g.offsets, g.tiles_touched, rows, flags, out,
                                                                                    scr.big_ctl, scr.big_desc, (uint32_t)scr.cap_big);
    else
        reduce_rows_k<K, REDUCE_GPQ, RECORD><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order,
                                                                        g.offsets, g.tiles_touched, rows, flags, out, scr.big_ctl,
                                                                        scr.big_desc, (uint32_t)scr.cap_big);
    // the big Gaussians: fixed, small grids of persistent workgroups (their numbers are on the device; N == 0: no blend ran,
    // nothing cleared the counters and nothing can be registered)
    if (N > 0)
        reduce_big_k<K, RECORD><<<dim3((unsigned)std::min<size_t>(REDUCE_BIG_GRID, scr.cap_big)), dim3(16 * BIG_PARTS), 0, s>>>(
            scr.big_ctl, scr.big_desc, (uint32_t)scr.cap_big, rows, flags, sc.S, nch, out);
}

// records: the sums stay in the row scratch as per-Gaussian records (see reduce_rows_k); the six arrays are not written
void launch_reduce_rows(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, float* dL_dmean2D,
                        float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic, float* dL_ddepth,
                        hipStream_t s, bool records) {
    const int rf = bwd_row_floats(sc.S), nch = 4 * ((sc.S + 3) / 4) + 4;
    const ReduceOut out{dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepth};
#define GOI_REDUCE(K)                                                                   \
    do {                                                                                \
        if (records) launch_reduce_k<K, true>(sc, g, N, nch, scr.rows, scr.flags, scr, out, s);  \
        else launch_reduce_k<K, false>(sc, g, N, nch, scr.rows, scr.flags, scr, out, s);         \
    } while (0)
    if (rf == 32) GOI_REDUCE(2);
    else if (rf == 16) GOI_REDUCE(1);
    else GOI_REDUCE(3);
#undef GOI_REDUCE
}

// bwd_records 2: only the BIG Gaussians are summed here (records into the row scratch); everything else is summed by
// preprocess_bwd_k.  Only laid out for 128-byte rows (K = 2: S = 5 .. 20); the caller checks.
void launch_reduce_big_only(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, hipStream_t s) {
    if (N <= 0) return;
    const int nch = 4 * ((sc.S + 3) / 4) + 4;
    const ReduceOut out{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const uint32_t* order = g.sort_vals[depth_sort_result_index()];
    find_big_k<<<dim3((sc.P + 255) / 256), dim3(256), 0, s>>>((uint32_t)N, g.counters + COUNTER_N, order, g.offsets, scr.big_ctl,
                                                             scr.big_desc, (uint32_t)scr.cap_big);
    reduce_big_k<2, true><<<dim3((unsigned)std::min<size_t>(REDUCE_BIG_GRID, scr.cap_big)), dim3(16 * BIG_PARTS), 0, s>>>(
        scr.big_ctl, scr.big_desc, (uint32_t)scr.cap_big, scr.rows, scr.flags, sc.S, nch, out);
}

void launch_reduce_sem_rows(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, int row_floats,
                            float* dL_dsemantic, hipStream_t s) {
    // rows hold semantic channels only: with nch = row_floats + 4 every element index is a semantic one
    const int nch = row_floats + 4;
    const ReduceOut out{nullptr, nullptr, nullptr, nullptr, dL_dsemantic, nullptr};
    if (row_floats == 16) launch_reduce_k<1, false>(sc, g, N, nch, scr.rows, scr.flags, scr, out, s);
    else launch_reduce_k<2, false>(sc, g, N, nch, scr.rows, scr.flags, scr, out, s);
}

}  // namespace goi
