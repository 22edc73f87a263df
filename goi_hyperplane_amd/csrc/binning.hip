// Binning between the per-Gaussian preprocess and the tile blend: compaction of the listed Gaussians for the depth sort,
// the load-balanced emit of (tile, Gaussian) instances in depth order, tile ranges.  Integer work (the one float divide in
// emit is corrected to the exact quotient), so this unit is compiled with the default contraction -- only the per-Gaussian
// arithmetic of preprocess.hip needs -ffp-contract=off to keep the reference's operation order.
// Reference: duplicateWithKeys / identifyTileRanges, cuda_rasterizer/rasterizer_impl.cu:70-138.
#include <algorithm>

#include "common.h"

namespace goi {

namespace {

// Compaction of the LISTED Gaussians (tiles_touched > 0) for the depth sort.  A workgroup covers COMPACT_ROUNDS
// consecutive blocks of preprocess_fwd_k (2048 Gaussians).  Its base rank is the sum of the per-block aggregates
// preprocess left behind for the blocks in front of it -- every workgroup adds them up itself (at most 3907 pairs at
// 1 M Gaussians, out of L2: cheaper than a scan kernel of its own plus the launch) -- and workgroup 0 also leaves the
// totals in counters[COUNTER_V] (listed Gaussians) and counters[COUNTER_N] (tiles touched = num_rendered).  A listed
// Gaussian puts (depth key, id) at its rank among the listed ones, i.e. in id order: the sort is stable, so ties keep
// ascending id as in the reference.  Reading the keys anyway, the workgroup counts their four digits for the onesweep
// sort (which then skips its own histogram pass); 2048 keys per workgroup keep the global atomics of that flush at the
// level of sweep_hist_k (one flush per 256 keys cost 2.3 M same-line atomics: +45 us).
// pad (a sort that cannot take its count from the device): the unlisted Gaussians follow with key 0xFFFFFFFF.
// COMPACT_ROUNDS: 8 blocks of preprocess per workgroup up to ~1 M Gaussians, 24 above (the flush of the digit histograms is
// 4 x 256 global atomics per WORKGROUP whatever its size: 3 M Gaussians in workgroups of 2048 were 1.5 M same-line atomics).
template <int COMPACT_ROUNDS>
__global__ __launch_bounds__(PRE_BLOCK) void compact_listed_k(int P, const uint32_t* __restrict__ tiles_touched,
                                                              const uint32_t* __restrict__ raw_key,
                                                              const uint2* __restrict__ blk_agg,
                                                              const unsigned long long* __restrict__ blk_coarse,
                                                              uint32_t* __restrict__ counters,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                              uint32_t* __restrict__ ghist, int pad) {
    __shared__ uint32_t s_h[4][256];
    __shared__ uint32_t s_wv[COMPACT_ROUNDS][4];
    __shared__ uint32_t s_red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nblk = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    const int blk0 = blockIdx.x * COMPACT_ROUNDS;
    // ---- all of this workgroup's Gaussians are requested before anything is waited for
    uint32_t t[COMPACT_ROUNDS], key[COMPACT_ROUNDS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROUNDS; r++) {
        const int i = (blk0 + r) * PRE_BLOCK + tid;
        const bool live = i < P;
        t[r] = live ? tiles_touched[i] : 0u;
        key[r] = live ? raw_key[i] : 0xFFFFFFFFu;
    }
    // ---- base rank (blocks in front) and totals: whole groups of COARSE_BLOCKS preprocess blocks from their packed sums, the
    // blocks of this workgroup's own group one by one
    uint32_t before = 0, all_v = 0, all_t = 0;
    const int c0 = blk0 / COARSE_BLOCKS, ncoarse = (nblk + COARSE_BLOCKS - 1) / COARSE_BLOCKS;
    for (int i = tid; i < ncoarse; i += PRE_BLOCK) {
        const unsigned long long a = blk_coarse[(size_t)i * COARSE_STRIDE];
        const uint32_t av = (uint32_t)(a >> 40);
        before += i < c0 ? av : 0u;
        all_v += av;
        all_t += (uint32_t)(a & ((1ull << 40) - 1ull));
    }
    for (int i = c0 * COARSE_BLOCKS + tid; i < blk0; i += PRE_BLOCK) before += blk_agg[i].x;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += (uint32_t)__shfl_xor((int)before, d, 64);
        all_v += (uint32_t)__shfl_xor((int)all_v, d, 64);
        all_t += (uint32_t)__shfl_xor((int)all_t, d, 64);
    }
    if (lane == 0) {
        s_red[0][w] = before;
        s_red[1][w] = all_v;
        s_red[2][w] = all_t;
    }
    if (ghist) {
#pragma unroll
        for (int p = 0; p < 4; p++) s_h[p][tid] = 0;
    }
    unsigned long long bal[COMPACT_ROUNDS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROUNDS; r++) {
        bal[r] = __ballot(t[r] > 0);
        if (lane == 0) s_wv[r][w] = (uint32_t)__popcll(bal[r]);
    }
    __syncthreads();
    uint32_t base = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const uint32_t V = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    if (blockIdx.x == 0 && tid == 0) {
        counters[COUNTER_V] = V;
        counters[COUNTER_N] = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
    }
#pragma unroll
    for (int r = 0; r < COMPACT_ROUNDS; r++) {
        const int i = (blk0 + r) * PRE_BLOCK + tid;
        uint32_t v_ex = base + (uint32_t)__popcll(bal[r] & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; k++) v_ex += s_wv[r][k];
        base += s_wv[r][0] + s_wv[r][1] + s_wv[r][2] + s_wv[r][3];
        if (i < P) {
            if (t[r] > 0) {
                keys[v_ex] = key[r];
                vals[v_ex] = (uint32_t)i;
                if (ghist) {
#pragma unroll
                    for (int p = 0; p < 4; p++) atomicAdd(&s_h[p][(key[r] >> (8 * p)) & 255u], 1u);
                }
            } else if (pad) {
                const uint32_t pos = V + ((uint32_t)i - v_ex);  // unlisted Gaussians before i
                keys[pos] = 0xFFFFFFFFu;
                vals[pos] = (uint32_t)i;
            }
        }
    }
    if (ghist) {
        __syncthreads();
        // (one of SORT_GH_COPIES copies of the histograms, by workgroup: 488 workgroups adding into the SAME thousand words
        // serialise at the L2's atomic units; the sort's passes add the copies up)
        uint32_t* gh = ghist + (size_t)(blockIdx.x % SORT_GH_COPIES) * SORT_MAX_PASSES * 256;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t c = s_h[p][tid];
            if (c) atomicAdd(&gh[p * 256 + tid], c);
        }
    }
}

// Emits the (tile id, Gaussian id) instances of every visible Gaussian, walking the Gaussians in
// depth order so that a stable sort by tile alone reproduces the reference's (tile, depth, id)
// order (CR/rasterizer_impl.cu:70-111 emits 64-bit tile|depth keys in id order instead).
// Wave-cooperative: each lane prepares one Gaussian (rectangle, output offset), then the wave walks
// its 64 Gaussians one at a time and all lanes write that Gaussian's instances side by side, so every
// store instruction covers one contiguous run instead of 64 scattered words.
// COUNT: the block also histograms its keys per tile in LDS and adds the non-empty bins to tile_count[]
// (stride 2: the .y words of the ranges array).  The per-tile counts are the tile ranges before their
// prefix sum AND, folded by digit, the global histograms the onesweep tile sort needs: counting here
// removes the sort's histogram pass over the 8 M keys and the ranges pass over the sorted keys.
// Rounds of 256 Gaussians per workgroup of the counting variant: 4 amortise zeroing and flushing the per-tile histogram (T
// words of LDS, up to T global atomics per workgroup) on a frame of thousands of tiles; a small tile grid (a 512 x 512 close-up:
// 1024 tiles) flushes next to nothing, and what it needs is workgroups: with 240 k listed Gaussians four rounds left 235
// workgroups for 256 CUs (emit 261 us for 4.4 M instances; one round: see profiles/README.md)
#ifndef GOI_EMIT_SMALL_T
#define GOI_EMIT_SMALL_T 2048
#endif
constexpr int EMIT_ROUNDS = 4;
#ifndef GOI_EMIT_MAX_GRID
#define GOI_EMIT_MAX_GRID 512  // persistent workgroups of the wide counting emit: two per CU
#endif
constexpr int emit_rounds_for(int T) { return T <= GOI_EMIT_SMALL_T ? 1 : EMIT_ROUNDS; }

// Rectangles above this are written by the big-rectangle loop of emit_k.  (1024 until round 5; the depth order puts the near --
// large -- Gaussians at the front, so the first waves of a close-up hold nothing but rectangles of several hundred tiles and ran
// two hundred trips of the common expansion while the rest of the chip had finished: emit 171 -> 90 us on the 512 x 512
// close-up of a 3 M scene with 128, 99 -> 93 on the clustered scene, unchanged on the headline; same-box A/B.)
#ifndef GOI_EMIT_BIG_TILES
#define GOI_EMIT_BIG_TILES 128
#endif
constexpr int EMIT_BIG_TILES = GOI_EMIT_BIG_TILES;
// WIDE (round 5, the counting variant with four rounds): the four rounds of a workgroup run SIDE BY SIDE -- 1024 threads, each
// group of 256 takes one round -- instead of one after the other.  Same per-tile histogram, same number of flushes, four times
// the waves: with 488 workgroups of four waves the headline frame kept 7 waves per CU busy, and emit is nothing but dependent
// gathers (rank -> id -> record) in front of its stores.
template <bool COUNT, int ROUNDS_COUNTING = EMIT_ROUNDS, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? 1024 : 256) void emit_k(int P, int gx, int gy, const GaussRec* __restrict__ rec,
                                              const int* __restrict__ radii, const uint32_t* __restrict__ order,
                                              const uint32_t* __restrict__ offsets, uint4* __restrict__ aux,
                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                              uint32_t* __restrict__ tile_count, uint32_t* __restrict__ counters,
                                              uint32_t* __restrict__ clear, uint32_t clear_words, uint32_t cap,
                                              uint32_t* __restrict__ bigq) {
    // cap: number of instances keys[] / vals[] can hold.  The exact forward sizes them for num_rendered, so the
    // guard below never fires; the speculative forward sizes them from a guess, and a frame that overflows must
    // stay memory-safe and self-consistent (the tile counts only count what was stored) until the host notices.
    constexpr uint32_t NT = WIDE ? 1024u : 256u;  // threads of the workgroup
    constexpr int NWV = WIDE ? 16 : 4;           // its waves
    const bool cull = counters[COUNTER_CULL] != 0;
    P = min(P, (int)counters[COUNTER_V]);  // order[] / offsets[] hold the LISTED Gaussians only (front of the depth order)
    // the frame's "truncated" flag (COUNTER_OVF): emit is the first kernel that knows both the count and the capacity
    // (... or was depth-sorted wrongly because a look-back of the sort timed out: COUNTER_SORTERR)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counters[COUNTER_OVF] = (counters[COUNTER_N] > cap ? 1u : 0u) | (counters[COUNTER_SORTERR] ? 2u : 0u);
    // COUNT: the blocks also zero the control words of the tile sort that follows (its own memset launch otherwise)
    if (COUNT)
        for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < clear_words; i += gridDim.x * NT) clear[i] = 0u;
    constexpr int ROUNDS_PER_BLOCK = COUNT ? ROUNDS_COUNTING : 1;
    if ((int)blockIdx.x * ROUNDS_PER_BLOCK * 256 >= P) return;  // (block-uniform) nothing listed left for this block
    extern __shared__ uint32_t s_cnt[];  // [gx * gy] when COUNT
    __shared__ unsigned long long s_mask[NWV][64];  // the rectangles' tile masks (cull_variant 2)
    __shared__ unsigned long long s_mark[NWV];  // per wave and trip: bit p = some rectangle's last instance is at position p
    __shared__ uint4 s_info[NWV][64];  // (x0 | y0 << 16, exclusive count, offsets[] - exclusive count, Gaussian id)
    __shared__ int s_w[NWV][64];       // rectangle width in tiles
    const int T = gx * gy;
    if (COUNT) {
        for (int t = threadIdx.x; t < T; t += (int)NT) s_cnt[t] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the counting variant amortises zeroing and flushing its tile histogram over EMIT_ROUNDS x 256 Gaussians
    constexpr int ROUNDS = COUNT ? ROUNDS_COUNTING : 1;
    // WIDE: this thread's group of 256 takes ONE round (the loop below runs once); otherwise the block walks its rounds
    const int rnd_first = WIDE ? (int)(threadIdx.x >> 8) : 0, rnd_end = WIDE ? rnd_first + 1 : ROUNDS;
    const int tid256 = (int)(threadIdx.x & 255u);
    // a round's Gaussian: id -> radius, position and box are dependent gathers (two DRAM round trips); the next round's
    // are requested before this round's instances are written, or every round would start with both exposed (emit is a
    // small kernel: two waves per SIMD have nothing to hide them behind)
    struct Fetched {
        uint32_t g, off;
        int r;
        float4 q0, q2;
        unsigned long long mask;
    };
    // PERSISTENT (round 5): a workgroup walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... of ROUNDS x 256 Gaussians and
    // flushes its per-tile histogram ONCE at the end -- the flush is up to T global atomics per workgroup, and with a workgroup
    // per chunk a 3 M scene paid 1 465 x 6 600 of them (launch_emit_counting caps the grid)
    int chunk = (int)blockIdx.x;
    auto fetch = [&](int rnd) {
        Fetched f{0u, 0u, 0, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, -1.f, -1.f), TMASK_FULL};
        const int i = (chunk * ROUNDS + rnd) * 256 + tid256;
        if (rnd < rnd_end && i < P) {
            f.g = order[i];
            f.off = offsets[i];
            const uint4 a = aux[f.g];  // radius and tile mask: one gather
            f.r = (int)a.y;
            f.q0 = rec[f.g].q0;
            f.q2 = rec[f.g].q1;  // (conic c, opacity, hx, hy)
            f.mask = aux_mask(a);
        }
        return f;
    };
    // chunk order of workgroup b: b, 2G-1-b, 2G+b, 4G-1-b, ... (G workgroups): the depth order puts the near Gaussians -- big
    // rectangles, expensive chunks -- in front, so a workgroup that draws an expensive chunk on the way up draws a cheap one
    // on the way down (with a plain stride the clustered scene's emit went 74 -> 83 us)
    const int nchunks = (P + ROUNDS * 256 - 1) / (ROUNDS * 256);
#pragma unroll 1
    for (int trip = 0;; trip++) {
    const int G2 = 2 * (int)gridDim.x;
    chunk = (trip >> 1) * G2 + ((trip & 1) ? G2 - 1 - (int)blockIdx.x : (int)blockIdx.x);
    if ((trip >> 1) * G2 >= nchunks) break;  // (block-uniform)
    if (chunk >= nchunks) continue;          // (the way down starts beyond the last chunk)
    Fetched nxt = fetch(rnd_first);
#pragma unroll 1
    for (int rnd = rnd_first; rnd < rnd_end; rnd++) {
        const int i = (chunk * ROUNDS + rnd) * 256 + tid256;
        const Fetched cur = nxt;
        nxt = fetch(rnd + 1);
        const uint32_t g = cur.g, off = cur.off;
        int x0 = 0, y0 = 0, w = 1, cnt = 0;
        if (i < P) {
            reinterpret_cast<uint32_t*>(aux + g)[0] = off;  // the Gaussian's first row slot in the backward (slot space = emit order = depth order)
            if (cur.r > 0) {
                int x1, y1;
                listed_rect(cur.q0.x, cur.q0.y, cur.r, cur.q2.z, cur.q2.w, cull, gx, gy, x0, y0, x1, y1);
                w = x1 - x0;
                cnt = cur.mask == TMASK_FULL ? w * (y1 - y0) : __popcll(cur.mask);  // (cull_variant 2: the ellipse's tiles)
            }
        }
        // Load-balanced expansion: the wave's 64 rectangles hold `total` (tile, Gaussian) instances; lane l of trip
        // t0 produces instance t0 + l, whichever rectangle it falls into.  A rectangle has ~10 tiles on average: one
        // rectangle per trip would leave 5/6 of the lanes idle.  Consecutive instances are consecutive addresses
        // (offsets[] is the exclusive scan of the same counts in the same order): full-line stores.
        // Which rectangle: the non-empty rectangles are numbered in lane order (their records sit at that number in
        // LDS); each marks the position of its LAST instance in a 64-bit word for the trip it falls into, and an
        // instance belongs to rectangle (rectangles that ended before the trip) + (marks below its own position).  Two
        // dependent LDS round trips per trip; the binary search in the scanned counts this replaces had seven, and with
        // two waves per SIMD (emit is a small kernel) their latency was the kernel: 59 -> 3x us.
        // A BIG rectangle (a frame-filling blob, a long needle: thousands of tiles) is taken out of the wave's common
        // expansion -- one such rectangle kept its wave in the loop below for a hundred trips of ~80 instructions and two LDS
        // round trips each while the rest of the chip had finished (clustered workload: emit 0.16 ms against 0.085 for its
        // instance count) -- and written by a loop that needs none of that: all 64 lanes walk ITS instances, a division per
        // instance.  Same keys, same values, same positions (a big rectangle is always a full one: the ellipse masks cover
        // rectangles of at most 64 tiles).
        // Round 5: the counting variant for SMALL tile grids does not write its big rectangles at all -- the depth order puts the near Gaussians in
        // front, so on a close-up the first few waves of the grid held nothing but rectangles of hundreds of tiles and wrote
        // a third of the frame's instances while the rest of the chip had finished (emit 87 us for 2.6 M instances).  They go to
        // a queue (one atomic per wave that has any), and emit_big_k, launched behind this kernel, spreads them over the chip:
        // a workgroup per rectangle.  Same keys, same values, same positions.
        // Only on SMALL tile grids (ROUNDS_COUNTING == 1: a close-up), where rectangles are large against the frame: on a
        // 1600 x 1056 frame the second launch costs what it saves (headline +5 us for an empty queue, clustered 96 -> 70 + 33 us:
        // a frame-filling blob makes its workgroup flush all 6600 tile counters).
        constexpr bool QUEUE = COUNT && ROUNDS_COUNTING == 1;
        const int cnt_own = cnt;
        const unsigned long long big_lanes = __ballot(cnt > EMIT_BIG_TILES);
        if (cnt > EMIT_BIG_TILES) cnt = 0;
        if (QUEUE) {
            if (big_lanes) {
                uint32_t qbase = 0;
                if (lane == 0) qbase = atomicAdd(&counters[COUNTER_BIGQ], (uint32_t)__popcll(big_lanes));
                qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
                if (cnt_own > EMIT_BIG_TILES)  // (its rank in the depth order: emit_big_k looks the rest up again)
                    bigq[qbase + (uint32_t)__popcll(big_lanes & ((1ull << lane) - 1ull))] = (uint32_t)i;
            }
        } else {
            for (unsigned long long bl = big_lanes; bl; bl &= bl - 1) {
                const int l = __builtin_ctzll(bl);
                const int bx0 = __builtin_amdgcn_readlane(x0, l), by0 = __builtin_amdgcn_readlane(y0, l);
                const int bw = __builtin_amdgcn_readlane(w, l), bcnt = __builtin_amdgcn_readlane(cnt_own, l);
                const uint32_t boff = (uint32_t)__builtin_amdgcn_readlane((int)off, l), bg = (uint32_t)__builtin_amdgcn_readlane((int)g, l);
                const float inv_w = __builtin_amdgcn_rcpf((float)bw);
                for (int k = lane; k < bcnt; k += 64) {
                    int row = (int)((float)k * inv_w);  // k / bw, off by at most one
                    row -= (row * bw > k);
                    row += ((row + 1) * bw <= k);
                    const uint32_t key = (uint32_t)((by0 + row) * gx + bx0 + (k - row * bw));
                    const uint32_t pos = boff + (uint32_t)k;
                    if (pos < cap) {
                        keys[pos] = key;
                        vals[pos] = bg;
                        if (COUNT) atomicAdd(&s_cnt[key], 1u);
                    }
                }
            }
        }
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int excl = incl - cnt;
        const int my_rank = __popcll(__ballot(cnt > 0) & ((1ull << lane) - 1ull));
        if (cnt > 0) {
            s_info[wv][my_rank] = make_uint4((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)excl, off - (uint32_t)excl, g);
            s_w[wv][my_rank] = w;
            s_mask[wv][my_rank] = cur.mask;
        }
        int ended = 0;  // non-empty rectangles that end before the current trip (wave-uniform)
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            if (lane == 0) s_mark[wv] = 0ull;
            const int last = incl - 1 - t0;  // position of this rectangle's last instance relative to the trip
            if (cnt > 0 && last >= 0 && last < 64)
                atomicOr(reinterpret_cast<unsigned int*>(&s_mark[wv]) + (last >> 5), 1u << (last & 31));
            __builtin_amdgcn_wave_barrier();
            const unsigned long long marks = s_mark[wv];
            __builtin_amdgcn_wave_barrier();
            const int lo = ended + __popcll(marks & ((1ull << lane) - 1ull));
            ended += __popcll(marks);
            if (t >= total) continue;
            const uint4 info = s_info[wv][lo];
            const int wl = s_w[wv][lo];
            const unsigned long long mk = s_mask[wv][lo];
            int k = t - (int)info.y;
            if (mk != TMASK_FULL) k = select_bit(mk, k);  // the k-th tile the ellipse reaches -> its index in the rectangle
            int row = (int)((float)k * __builtin_amdgcn_rcpf((float)wl));  // k / wl, off by at most one
            row -= (row * wl > k);
            row += ((row + 1) * wl <= k);
            const int col = k - row * wl;
            const uint32_t key = (uint32_t)(((int)(info.x >> 16) + row) * gx + (int)(info.x & 0xFFFFu) + col);
            const uint32_t pos = info.z + (uint32_t)t;
            if (pos < cap) {
                keys[pos] = key;
                vals[pos] = info.w;
                if (COUNT) atomicAdd(&s_cnt[key], 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    }  // chunks
    if (COUNT) {
        __syncthreads();
        // (the counts of tiles 2 k and 2 k + 1 leave as ONE 64-bit atomic on a dense count array -- the first T words of the
        // ranges array, which tile_ranges_hist_k reads before it writes the ranges: half the atomics; a count is < 2^31)
        unsigned long long* pair = reinterpret_cast<unsigned long long*>(tile_count);
        for (int k = threadIdx.x; 2 * k < T; k += (int)NT) {
            const unsigned long long lo = s_cnt[2 * k], hi = 2 * k + 1 < T ? s_cnt[2 * k + 1] : 0u;
            if (lo | hi) atomicAdd(&pair[k], lo | (hi << 32));
        }
    }
}

// The big rectangles emit_k<true> queued (more than EMIT_BIG_TILES tiles: always full rectangles): a workgroup per rectangle,
// 256 lanes walk its instances, a division per instance; the per-tile counts go through the same LDS histogram.
__global__ __launch_bounds__(256) void emit_big_k(int gx, int gy, const GaussRec* __restrict__ rec,
                                                  const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                  const uint4* __restrict__ aux, const uint32_t* __restrict__ bigq,
                                                  const uint32_t* __restrict__ counters, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals, uint32_t* __restrict__ tile_count, uint32_t cap) {
    const uint32_t nbig = counters[COUNTER_BIGQ];
    if (blockIdx.x >= nbig) return;  // (block-uniform)
    const bool cull = counters[COUNTER_CULL] != 0;
    const int T = gx * gy;
    extern __shared__ uint32_t s_cnt[];  // [T]
    for (int t = threadIdx.x; t < T; t += 256) s_cnt[t] = 0;
    __syncthreads();
    // The queue holds ranks in the depth order; the rectangle is formed again exactly as emit_k formed it (a big rectangle is
    // always a full one: the ellipse masks cover rectangles of at most 64 tiles).  Rank -> id, first instance -> record: three
    // dependent gathers, so the block looks up 256 of its rectangles at a time, one per thread, and then walks them together
    // (looked up one by one in front of each rectangle the gathers were the kernel: 19 -> 38 us on the close-up).
    __shared__ uint4 s_rect[256];   // (x0 | y0 << 16, w, instances, first instance)
    __shared__ uint32_t s_gid[256];
    for (uint32_t j0 = blockIdx.x; j0 < nbig; j0 += 256u * gridDim.x) {
        const uint32_t j = j0 + threadIdx.x * gridDim.x;
        if (j < nbig) {
            const uint32_t i = bigq[j];
            const uint32_t gid = order[i], boff = offsets[i];
            const uint4 a = aux[gid];
            const float4 q0 = rec[gid].q0, q1 = rec[gid].q1;
            int bx0, by0, bx1, by1;
            listed_rect(q0.x, q0.y, (int)a.y, q1.z, q1.w, cull, gx, gy, bx0, by0, bx1, by1);
            s_rect[threadIdx.x] = make_uint4((uint32_t)bx0 | ((uint32_t)by0 << 16), (uint32_t)(bx1 - bx0),
                                             (uint32_t)((bx1 - bx0) * (by1 - by0)), boff);
            s_gid[threadIdx.x] = gid;
        }
        __syncthreads();
        const uint32_t left = (nbig - j0 + gridDim.x - 1) / gridDim.x;  // rectangles of this block from j0 on
        const int nb = (int)(left < 256u ? left : 256u);
        for (int b = 0; b < nb; b++) {
            const uint4 q = s_rect[b];
            const uint32_t gid = s_gid[b];
            const int bx0 = (int)(q.x & 0xFFFFu), by0 = (int)(q.x >> 16), bw = (int)q.y, bcnt = (int)q.z;
            const float inv_w = __builtin_amdgcn_rcpf((float)bw);
            for (int k = threadIdx.x; k < bcnt; k += 256) {
                int row = (int)((float)k * inv_w);  // k / bw, off by at most one
                row -= (row * bw > k);
                row += ((row + 1) * bw <= k);
                const uint32_t key = (uint32_t)((by0 + row) * gx + bx0 + (k - row * bw));
                const uint32_t pos = q.w + (uint32_t)k;
                if (pos < cap) {
                    keys[pos] = key;
                    vals[pos] = gid;
                    atomicAdd(&s_cnt[key], 1u);
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    unsigned long long* pair = reinterpret_cast<unsigned long long*>(tile_count);  // (dense counts, two tiles per atomic: emit_k)
    for (int k = threadIdx.x; 2 * k < T; k += 256) {
        const unsigned long long lo = s_cnt[2 * k], hi = 2 * k + 1 < T ? s_cnt[2 * k + 1] : 0u;
        if (lo | hi) atomicAdd(&pair[k], lo | (hi << 32));
    }
}

// One workgroup: per-tile counts (in ranges[t].y) -> ranges[t] = [start, end) ((0,0) for an empty tile, as
// the reference leaves it) and the global digit histograms of the tile sort's passes.  A thread owns IT = ceil(T / 1024)
// consecutive tiles (IT <= 12: emit only counts grids of at most 12288 tiles), so the prefix sum is ONE block scan
// (chunks of 1024 tiles with three barriers each took 14 us at 6600 tiles: pure latency).
constexpr int TRH_MAX_IT = 12;
__global__ __launch_bounds__(1024) void tile_ranges_hist_k(int T, uint2* __restrict__ ranges, int passes, int shift0,
                                                           int nbits0, int shift1, int nbits1,
                                                           uint32_t* __restrict__ ghist, uint32_t* __restrict__ counters) {
    if (threadIdx.x == 0) counters[COUNTER_BIGQ] = 0u;  // (emit_big_k has run: the next emit from this workspace -- a redo -- starts an empty queue)
    __shared__ uint32_t s_h[2][256];
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 512) (&s_h[0][0])[tid] = 0;
    const int IT = (T + 1023) / 1024;
    const int t0 = tid * IT;
    uint32_t c[TRH_MAX_IT];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < TRH_MAX_IT; k++) {
        c[k] = (k < IT && t0 + k < T) ? reinterpret_cast<const uint32_t*>(ranges)[t0 + k] : 0u;  // (dense counts: emit_k's flush)
        sum += c[k];
    }
    uint32_t v = sum;  // inclusive scan of the 1024 per-thread sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    if (lane == 63) s_wave[w] = v;
    __syncthreads();  // (also orders the zeroing of s_h before the atomics below)
    uint32_t run = v - sum;
    for (int k = 0; k < w; k++) run += s_wave[k];
#pragma unroll
    for (int k = 0; k < TRH_MAX_IT; k++) {
        const int t = t0 + k;
        if (k < IT && t < T) {
            const uint32_t ck = c[k];
            ranges[t] = ck ? make_uint2(run, run + ck) : make_uint2(0u, 0u);
            if (ck) {
                atomicAdd(&s_h[0][((uint32_t)t >> shift0) & ((1u << nbits0) - 1u)], ck);
                if (passes > 1) atomicAdd(&s_h[1][((uint32_t)t >> shift1) & ((1u << nbits1) - 1u)], ck);
            }
            run += ck;
        }
    }
    __syncthreads();
    if (tid < 256) {
        ghist[tid] = s_h[0][tid];
        ghist[256 + tid] = s_h[1][tid];
    }
}

// Per-tile [start,end) from the tile-sorted key list (CR/rasterizer_impl.cu:116-138).
__global__ __launch_bounds__(256) void ranges_k(int N_cap, const uint32_t* __restrict__ n_dev,
                                                const uint32_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const int N = n_dev ? (int)min((uint32_t)N_cap, *n_dev) : N_cap;  // (speculative forward: the count is on the device)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t cur = keys[i];
    if (i == 0)
        ranges[cur].x = 0;
    else {
        const uint32_t prev = keys[i - 1];
        if (cur != prev) {
            ranges[prev].y = i;
            ranges[cur].x = i;
        }
    }
    if (i == N - 1) ranges[cur].y = N;
}

}  // namespace

void launch_compact_listed(int P, const GeomView& g, uint32_t* ghist, bool pad, hipStream_t s) {
    const int nblk = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    if (nblk <= 6144)
        compact_listed_k<8><<<dim3((nblk + 7) / 8), dim3(PRE_BLOCK), 0, s>>>(
            P, g.tiles_touched, g.sort_keys[1], g.blk_agg, g.blk_coarse, g.counters, g.sort_keys[0], g.sort_vals[0], ghist, pad ? 1 : 0);
    else
        compact_listed_k<24><<<dim3((nblk + 23) / 24), dim3(PRE_BLOCK), 0, s>>>(
            P, g.tiles_touched, g.sort_keys[1], g.blk_agg, g.blk_coarse, g.counters, g.sort_keys[0], g.sort_vals[0], ghist, pad ? 1 : 0);
}

void launch_emit(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                 uint32_t* vals, uint32_t cap, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    emit_k<false><<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, gx, gy, g.rec, radii, order, g.offsets, g.aux, keys, vals,
                                                              nullptr, g.counters, nullptr, 0u, cap, nullptr);
}

bool emit_can_count_tiles(int W, int H) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    // (tile_ranges_hist_k: at most TRH_MAX_IT x 1024 tiles; the per-tile LDS counters of emit are DYNAMIC LDS on top of the
    // kernel's static arrays -- 28.2 KB in the 1024-thread form: masks 8, info 16, widths 4 KB -- and the sum has to stay inside
    // the 64 KB a workgroup may be given without asking for more: 8 960 tiles, e.g. 2048 x 1120.  Larger grids take the
    // non-counting emit + ranges_k.)
    constexpr size_t EMIT_STATIC_LDS = 16 * 64 * (sizeof(unsigned long long) + sizeof(uint4) + sizeof(int)) + 16 * sizeof(unsigned long long);
    return (size_t)gx * gy <= (size_t)TRH_MAX_IT * 1024 && (size_t)gx * gy * sizeof(uint32_t) + EMIT_STATIC_LDS <= 64 * 1024 &&
           tile_key_bits((uint32_t)(gx * gy)) <= 16;
}

// emit + per-tile counts; ranges must be zeroed by the caller's stream order (done here)
void launch_emit_counting(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                          uint32_t* vals, uint2* ranges, uint32_t* clear, size_t clear_words, uint32_t cap, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    // `ranges` was zeroed by preprocess_fwd_k
    if (emit_rounds_for(gx * gy) == 1) {
        emit_k<true, 1><<<dim3((P + 255) / 256), dim3(256), (size_t)gx * gy * sizeof(uint32_t), s>>>(
            P, gx, gy, g.rec, radii, order, g.offsets, g.aux, keys, vals, reinterpret_cast<uint32_t*>(ranges), g.counters,
            clear, (uint32_t)clear_words, cap, g.bigq);
        // the big rectangles the kernel queued: a workgroup per CU walks the queue (workgroups beyond its length leave at once)
        emit_big_k<<<dim3(256), dim3(256), (size_t)gx * gy * sizeof(uint32_t), s>>>(
            gx, gy, g.rec, order, g.offsets, g.aux, g.bigq, g.counters, keys, vals, reinterpret_cast<uint32_t*>(ranges), cap);
    } else
        emit_k<true, EMIT_ROUNDS, true><<<dim3(std::min((P + 256 * EMIT_ROUNDS - 1) / (256 * EMIT_ROUNDS), GOI_EMIT_MAX_GRID)), dim3(1024), (size_t)gx * gy * sizeof(uint32_t), s>>>(
            P, gx, gy, g.rec, radii, order, g.offsets, g.aux, keys, vals, reinterpret_cast<uint32_t*>(ranges), g.counters,
            clear, (uint32_t)clear_words, cap, g.bigq);
}

// per-tile counts -> ranges and the two digit histograms (written to ghist[0..511]) of a sort on
// key bits [0, bits) split as radix_sort_pairs splits them
void launch_tile_ranges_hist(int W, int H, uint2* ranges, uint32_t* ghist, uint32_t* counters, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int bits = tile_key_bits((uint32_t)(gx * gy));
    const int passes = (bits + 7) / 8;
    const int n0 = (bits + passes - 1) / passes, n1 = bits - n0;
    tile_ranges_hist_k<<<dim3(1), dim3(1024), 0, s>>>(gx * gy, ranges, passes, 0, n0, n0, n1 > 0 ? n1 : 1, ghist, counters);
}

void launch_ranges(int N, const uint32_t* n_dev, const uint32_t* sorted_keys, uint2* ranges, int T, hipStream_t s) {
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, s);
    if (N > 0) ranges_k<<<dim3((N + 255) / 256), dim3(256), 0, s>>>(N, n_dev, sorted_keys, ranges);
}

}  // namespace goi
