// Device-wide exclusive scan and stable LSD radix sort for gfx950 (wave64), hand-written.
//
// They replace the two cub calls of the reference's binning stage
// (cub::DeviceScan::InclusiveSum, CR/rasterizer_impl.cu:281; cub::DeviceRadixSort::SortPairs,
// CR/rasterizer_impl.cu:307-312).  Integer work: results are bit-exact by construction
// (prefix sum; stable ascending sort on a bit range).
//
// Scan: reduce-then-scan over 2048-element chunks (3 launches).
// Sort: per pass, (a) per-block digit histograms, (b) one exclusive scan over the
// digit-major [RADIX][blocks] table, (c) stable scatter.  Stability inside a block comes from
// wave-level match-any ranking (ballots over the digit bits) with elements laid out so that
// (wave, round, lane) order equals index order.
#include <atomic>

#include "common.h"

namespace goi {

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;  // 2048

#ifndef GOI_SORT_THREADS
#define GOI_SORT_THREADS 512  // 8192-key onesweep tiles: 256 -> 512 threads took the tile sort from 107 to 89 us (look-back chain)
#endif
constexpr int SORT_THREADS = GOI_SORT_THREADS;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096
constexpr int SORT_WAVES = SORT_THREADS / WAVE;        // 4
constexpr int SORT_WAVE_ITEMS = SORT_TILE / SORT_WAVES;  // 1024
constexpr int RADIX_MAX = 256;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread across a block of NT threads; returns the exclusive
// prefix, *block_total gets the sum.  `sm` needs NT/64 + 1 words.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* sm, uint32_t* block_total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = NT / WAVE;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            uint32_t t = sm[i];
            sm[i] = run;
            run += t;
        }
        sm[NW] = run;
    }
    __syncthreads();
    uint32_t res = inc - v + sm[w];
    *block_total = sm[NW];
    __syncthreads();  // sm may be reused by the caller
    return res;
}

// n_dev (may be NULL): the element count lives on the device and `n` is a capacity (the grid covers n, the kernels
// work on min(*n_dev, n) elements; chunks past the count contribute 0 / write nothing).
__device__ __forceinline__ size_t scan_n(size_t n, const uint32_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    const size_t m = (size_t)*n_dev;
    return m < n ? m : n;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_k(const uint32_t* __restrict__ in,
                                                              const uint32_t* __restrict__ gather, size_t n_cap,
                                                              const uint32_t* __restrict__ n_dev,
                                                              uint32_t* __restrict__ partials,
                                                              uint32_t* __restrict__ stash) {
    // stash (with gather): the gathered values are left there in scan order -- the second kernel then reads them as a stream
    // instead of gathering them again (3 M Gaussians: scan_apply_k 25 -> 10 us)
    __shared__ uint32_t sm[SCAN_THREADS / WAVE + 1];
    const size_t n = scan_n(n_cap, n_dev);
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    if (base >= n) {  // (block-uniform)
        if (threadIdx.x == 0) partials[blockIdx.x] = 0u;
        return;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + (size_t)k * SCAN_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t v = gather ? in[gather[i]] : in[i];
            if (stash) stash[i] = v;
            sum += v;
        }
    }
    uint32_t total;
    (void)block_exclusive_scan<SCAN_THREADS>(sum, sm, &total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void scan_partials_k(uint32_t* __restrict__ partials, size_t nb,
                                                        uint32_t* __restrict__ total_out) {
    __shared__ uint32_t sm[1024 / WAVE + 1];
    uint32_t carry = 0;
    for (size_t c = 0; c < nb; c += 1024) {
        size_t i = c + threadIdx.x;
        uint32_t v = i < nb ? partials[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(v, sm, &tot);
        if (i < nb) partials[i] = ex + carry;
        carry += tot;
    }
    if (total_out && threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_k(const uint32_t* in,  // may alias out
                                                             const uint32_t* __restrict__ gather,
                                                             uint32_t* out, size_t n_cap,
                                                             const uint32_t* __restrict__ n_dev,
                                                             const uint32_t* __restrict__ partials, int raw_partials) {
    __shared__ uint32_t sm[SCAN_THREADS / WAVE + 1];
    const size_t n = scan_n(n_cap, n_dev);
    if ((size_t)blockIdx.x * SCAN_CHUNK >= n) return;  // (block-uniform)
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + k;
        v[k] = 0;
        if (i < n) v[k] = gather ? in[gather[i]] : in[i];
        sum += v[k];
    }
    // RAW partials (the three-kernel scan scanned them with a one-block kernel in between: 4.7 us of launch and latency for a few
    // hundred words): every block adds up the block sums in front of it itself -- at most n / SCAN_CHUNK words out of L2
    uint32_t before = 0;
    if (raw_partials) {
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += SCAN_THREADS) before += partials[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
        __shared__ uint32_t s_b[SCAN_THREADS / WAVE];
        if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = before;
        __syncthreads();
        before = 0;
#pragma unroll
        for (int w2 = 0; w2 < SCAN_THREADS / WAVE; w2++) before += s_b[w2];
    } else {
        before = partials[blockIdx.x];
    }
    uint32_t tot;
    uint32_t run = block_exclusive_scan<SCAN_THREADS>(sum, sm, &tot) + before;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}

// ---- radix sort -------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_k(const uint32_t* __restrict__ keys, size_t n, int shift,
                                                             uint32_t mask, uint32_t* __restrict__ hist,
                                                             uint32_t nblk) {
    __shared__ uint32_t h[SORT_WAVES][RADIX_MAX];
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX_MAX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        size_t i = base + (size_t)k * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[w][(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= mask; d += SORT_THREADS) {
        uint32_t t = 0;
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ww++) t += h[ww][d];
        hist[(size_t)d * nblk + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_k(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, size_t n, int shift,
                                                                int nbits, const uint32_t* __restrict__ base_tab,
                                                                uint32_t nblk) {
    __shared__ uint32_t cnt[SORT_WAVES][RADIX_MAX];
    const uint32_t mask = (1u << nbits) - 1u;
    for (int i = threadIdx.x; i < SORT_WAVES * RADIX_MAX; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    const size_t wbase = (size_t)blockIdx.x * SORT_TILE + (size_t)w * SORT_WAVE_ITEMS;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        size_t i = wbase + (size_t)r * WAVE + lane;
        const bool ok = i < n;
        key[r] = ok ? keys_in[i] : 0xFFFFFFFFu;
        val[r] = ok ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        size_t i = wbase + (size_t)r * WAVE + lane;
        const bool ok = i < n;
        const uint32_t d = (key[r] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            if (b < nbits) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
        }
        // Same-wave LDS accesses execute in program order: every lane reads the running count
        // of its digit before the group leader bumps it.
        const uint32_t prev = ok ? cnt[w][d] : 0u;
        const uint32_t below = (uint32_t)__popcll(peers & lt);
        rank[r] = prev + below;
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) cnt[w][d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= mask; d += SORT_THREADS) {
        uint32_t run = base_tab[(size_t)d * nblk + blockIdx.x];
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ww++) {
            uint32_t t = cnt[ww][d];
            cnt[ww][d] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        size_t i = wbase + (size_t)r * WAVE + lane;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = cnt[w][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

// ---- single-pass-per-digit radix sort ("onesweep") -----------------------------------------------
// One kernel per digit: every block ranks its 8192 elements, publishes its per-digit counts, obtains
// the counts of all earlier blocks by decoupled look-back over per-(block, digit) status words, and
// scatters through LDS so that each digit's elements leave as one contiguous run.  A prologue kernel
// computes the global digit histograms of every pass in one read of the keys.
//
// Inter-workgroup protocol (MI355X: per-XCD L2s are not coherent): a status word is ONE aligned
// 32-bit granule {2-bit state, 30-bit count} written and read with relaxed AGENT-scope atomics
// (write-through / L1-bypassing), so the data is its own flag and no fence is needed; block ids
// come from a ticket counter so a block only ever waits for blocks that have already started;
// every spin is bounded and reports through an error word instead of hanging the device.
constexpr uint32_t ST_EMPTY = 0u, ST_AGGREGATE = 1u << 30, ST_PREFIX = 2u << 30, ST_MASK = 3u << 30;
constexpr uint32_t ST_VALUE = (1u << 30) - 1u;
constexpr int MAX_PASSES = SORT_MAX_PASSES;
constexpr size_t GH_WORDS = (size_t)SORT_GH_COPIES * MAX_PASSES * RADIX_MAX;  // all copies of the global digit histograms

struct SweepPlan {
    int passes;
    int shift[MAX_PASSES];
    int nbits[MAX_PASSES];
};

// n_dev (may be NULL): the element count lives on the device (speculative forward: the host sized grids and buffers for
// a CAPACITY `n` and never learned the count); the kernels then work on min(*n_dev, n) elements and surplus blocks exit.
__device__ __forceinline__ size_t effective_n(size_t n, const uint32_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    const size_t m = (size_t)__builtin_nontemporal_load(n_dev);
    return m < n ? m : n;
}

__global__ __launch_bounds__(SORT_THREADS) void sweep_hist_k(const uint32_t* __restrict__ keys, size_t n_cap,
                                                             const uint32_t* __restrict__ n_dev, SweepPlan plan,
                                                             uint32_t* __restrict__ ghist) {
    const size_t n = effective_n(n_cap, n_dev);
    if ((size_t)blockIdx.x * SORT_TILE >= n) return;
    __shared__ uint32_t h[MAX_PASSES][RADIX_MAX];
    for (int i = threadIdx.x; i < MAX_PASSES * RADIX_MAX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    // (one LDS atomic per key and pass.  The keys are far from uniform -- float bits with a handful of exponents, half of
    // them the 0xFFFFFFFF of culled Gaussians -- but counting lanes that share a digit with ballots and adding once per
    // group was SLOWER than letting the LDS serialise the same-address atomics: depth sort 0.098 -> 0.118 ms.)
#pragma unroll 4
    for (int k = 0; k < SORT_ITEMS; k++) {
        size_t i = base + (size_t)k * SORT_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t key = keys[i];
            for (int p = 0; p < plan.passes; p++)
                atomicAdd(&h[p][(key >> plan.shift[p]) & ((1u << plan.nbits[p]) - 1u)], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = ghist + (size_t)(blockIdx.x % SORT_GH_COPIES) * MAX_PASSES * RADIX_MAX;  // (one of the copies: common.h)
    for (int i = threadIdx.x; i < plan.passes * RADIX_MAX; i += SORT_THREADS) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// Tile shape: THREADS x ITEMS keys per workgroup.  A pass lasts about as long as ONE tile takes (every tile of a
// 1 M-key sort is resident at once), and that is set by the ITEMS ranking rounds each thread runs back to back:
// 1024 x 4 sorts 1 M keys in 89 us (4 passes) where 512 x 16 needs 102; with 5 M keys the larger tile wins
// (88 vs 96 us for 2 passes: fewer tiles to look back over, fewer partial runs per digit).
// ADAPTIVE tile (min_items < ITEMS; round 5): the count lives on the device and can be a small fraction of the capacity the grid
// and the status table were laid out for (a 512 x 512 close-up of a 3 M scene lists 240 k Gaussians: 30 tiles of 8192 keys kept
// 30 of 256 CUs busy for 24 us per pass).  Every block derives the SAME keys-per-thread from the count -- the smallest power
// of two between min_items and ITEMS that leaves at most SWEEP_TARGET_TILES tiles -- and blocks beyond the tiles of that shape
// leave before they take a ticket.  (The status table has a row per tile of the SMALLEST shape: sweep_status_words.)
constexpr int SWEEP_TARGET_TILES = 512;
constexpr uint32_t SWEEP_GROUPED_MAX_TILES = 640;  // (grouped look-back up to this many tiles, the chained one above)
template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void sweep_pass_k(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, size_t n_cap,
                                                             const uint32_t* __restrict__ n_dev, int shift,
                                                             int nbits, const uint32_t* __restrict__ ghist,
                                                             uint32_t* status, uint32_t* ticket, uint32_t* error,
                                                             uint32_t* frame_error, int min_items, uint32_t* gstat) {
    constexpr int WAVES = THREADS / WAVE;
    __shared__ uint32_t cnt[WAVES][RADIX_MAX];  // per-wave digit counts -> per-wave local offsets
    __shared__ uint32_t gbase[RADIX_MAX];            // global position of this block's first element of digit d
    __shared__ uint32_t lbase[RADIX_MAX];            // local (in-block) exclusive offset of digit d
    __shared__ uint32_t s_gh[RADIX_MAX];             // the pass's global digit histogram (final before the pass starts)
    extern __shared__ uint32_t s_dyn[];  // [2][THREADS * ITEMS]: the tile's keys and values in digit order
    uint32_t* s_keys = s_dyn;
    uint32_t* s_vals = s_dyn + THREADS * ITEMS;
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_trivial;
    const uint32_t radix = 1u << nbits, mask = radix - 1u;
    // THE HEAD OF A BLOCK IS A CHAIN OF MEMORY ROUND TRIPS, and a pass is barely longer than one block: the count, the first
    // key, the histogram bin of its digit (the trivial-pass test), the ticket and only then the keys used to be FIVE dependent
    // trips (~4 us of a 14 us depth pass).  Now everything that does not depend on anything is requested at once: the count,
    // the whole digit histogram (kept in LDS: the digit bases further down need it anyway, and "one bin holds all n keys" is
    // the trivial-pass test without looking at a key) and -- where every block takes a ticket -- the ticket.
    const bool adaptive = min_items < ITEMS;
    uint32_t early_ticket = 0;
    if (!adaptive && threadIdx.x == 0 && ticket) early_ticket = atomicAdd(ticket, 1u);
    uint32_t gh = 0;
    if (threadIdx.x < radix) {
#pragma unroll
        for (int c = 0; c < SORT_GH_COPIES; c++) gh += ghist[(size_t)c * MAX_PASSES * RADIX_MAX + threadIdx.x];
    }
    const size_t n = effective_n(n_cap, n_dev);
    int items = ITEMS;
    while (items > min_items && n <= (size_t)THREADS * (size_t)(items / 2) * SWEEP_TARGET_TILES) items >>= 1;
    const int TILE_KEYS = THREADS * items, WAVE_ITEMS = TILE_KEYS / WAVES;
    // (adaptive: blocks beyond the tiles of the chosen shape leave WITHOUT a ticket -- exactly the tiles take one; otherwise
    // every block has taken one above and the ones past the count leave below, as they always did)
    if (adaptive && (size_t)blockIdx.x * TILE_KEYS >= n) return;
    if (threadIdx.x == 0) s_trivial = 0u;
    for (int i = threadIdx.x; i < WAVES * RADIX_MAX; i += THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    if (threadIdx.x < RADIX_MAX) {
        s_gh[threadIdx.x] = gh;
        if (n > 0 && gh == (uint32_t)n) s_trivial = 1u;  // (at most one bin can hold all n keys)
    }
    // (ticket == NULL, sort_tickets 0 -- an EXPERIMENT: the tile is the workgroup's own index, which is only safe while the
    // hardware starts workgroups in index order)
    if (threadIdx.x == 0) s_bid = !ticket ? blockIdx.x : (adaptive ? atomicAdd(ticket, 1u) : early_ticket);
    __syncthreads();
    // A TRIVIAL pass: every key has the same digit (the global histogram says so: one bin holds all n) -- the top byte of
    // the depth keys of a scene whose depths span less than a factor of four, typically.  The pass is then the identity
    // permutation: the tile is copied across, no ranking, no look-back (nobody will look back at this pass; its tickets are unused).
    if (s_trivial) {
        const size_t base = (size_t)blockIdx.x * TILE_KEYS;
        for (int r = 0; r < items; r++) {
            const size_t i = base + (size_t)r * THREADS + threadIdx.x;
            if (i < n) {
                keys_out[i] = keys_in[i];
                vals_out[i] = vals_in[i];
            }
        }
        return;
    }
    const uint32_t bid = s_bid;
    // the grid covers the capacity: tickets past the last tile of the real count have nothing to rank and nobody
    // looks back at them (tickets are dealt in order, so every tile below is owned by a running block)
    if ((size_t)bid * TILE_KEYS >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    const size_t tile_base = (size_t)bid * TILE_KEYS;
    const size_t wbase = tile_base + (size_t)w * WAVE_ITEMS;
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        size_t i = wbase + (size_t)r * WAVE + lane;
        const bool ok = r < items && i < n;
        key[r] = ok ? keys_in[i] : 0xFFFFFFFFu;
        val[r] = ok ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (r >= items) break;  // (wave-uniform)
        size_t i = wbase + (size_t)r * WAVE + lane;
        const bool ok = i < n;
        const uint32_t d = (key[r] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            if (b < nbits) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
        }
        const uint32_t prev = ok ? cnt[w][d] : 0u;
        const uint32_t below = (uint32_t)__popcll(peers & lt);
        rank[r] = prev + below;
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) cnt[w][d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- per digit: block count, wave offsets, publish, look back, global base
    uint32_t my_total = 0;
    if (threadIdx.x < radix) {
        const uint32_t d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) {
            const uint32_t t = cnt[ww][d];
            cnt[ww][d] = run;  // exclusive offset of wave ww inside the block's run of digit d
            run += t;
        }
        my_total = run;
        uint32_t* st = status + (size_t)d;  // status[block][digit], digit-minor
        __hip_atomic_store(st + (size_t)bid * RADIX_MAX, ST_AGGREGATE | my_total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        // Look back LB tiles per round trip.  Every tile of a pass is resident at once and they finish ranking at about the
        // same time, so a tile meets aggregates, not prefixes, 10-20 tiles deep (measured: 15-20 steps of ~600 clocks, 4-5 us of
        // a 16 us pass); the statuses of the next LB predecessors are independent loads.  LB = 1 / 2 / 3 / 4 / 8 / 16 / 32: depth
        // sort 72.7 / 70.9 / 70.2 / 70.4 / 71.4 / 74.3 / 79.2 us, tile sort 73.1 / 68.7 / 67.7 / 67.6 / 68.7 / 83.3 / 92.3 us
        // (deeper batches read statuses nobody has written yet and re-read them).
#ifndef GOI_SORT_LB
#define GOI_SORT_LB 4
#endif
        constexpr int LB = GOI_SORT_LB;
        uint32_t excl = 0, spins = 0;
        const uint32_t ntiles = (uint32_t)((n + (size_t)TILE_KEYS - 1) / (size_t)TILE_KEYS);
        if (gstat && ntiles <= SWEEP_GROUPED_MAX_TILES) {
            // GROUPED look-back (round 5).  Measured with timestamps in the kernel: of the ~12 us a 4 096-key tile of a depth
            // pass lives, the chained look-back above is 3.3-4.6 us -- every tile of a pass is resident and they all publish
            // at about the same time, so nobody meets a prefix for a dozen round trips.  With all aggregates there at once the
            // prefix needs no chain: tiles form groups of GS (16 / 32 / 64, ~sqrt of the tile count); a tile adds up the
            // aggregates of the tiles in front of it IN ITS GROUP (independent loads, one round trip), the last tile of a
            // group publishes the group's total, and every tile adds up the totals of the groups in front of its own (one more
            // round trip).  A tile only ever waits for tiles with lower tickets, which are running.  Same-box A/B (per pass):
            // depth sort 14.5 -> 12.6 us (125 tiles), close-up 12.6 -> 10.9 (236 tiles), 3 M 24.5 -> 22.4 (366 tiles); the tile
            // sort 33 -> 31 us at 500 tiles, but 83 -> 85 us at 1 465 tiles (63 + 22 words per thread): the chain stays above
            // SWEEP_GROUPED_MAX_TILES tiles.
            uint32_t gs_log = 4;
            while ((1u << (2 * gs_log)) < ntiles && gs_log < 6) gs_log++;
            const uint32_t G = bid >> gs_log, first = G << gs_log;
            auto gather = [&](const uint32_t* base, uint32_t lo, uint32_t hi) {
                uint32_t sum = 0;
                for (uint32_t r0 = lo; r0 < hi; r0 += 8) {
                    uint32_t v[8];
                    bool ok;
                    do {
                        ok = true;
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            v[j] = r0 + j < hi ? __hip_atomic_load(base + (size_t)(r0 + j) * RADIX_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : ST_PREFIX;
#pragma unroll
                        for (int j = 0; j < 8; j++) ok = ok && (v[j] & ST_MASK) != ST_EMPTY;
                        if (!ok) {
                            if (++spins > (1u << 22)) {  // (never hang the device: see the chained look-back below)
                                atomicOr(error, 2u);
                                if (frame_error) atomicOr(frame_error, 2u);
                                ok = true;
                            } else {
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                    } while (!ok);
#pragma unroll
                    for (int j = 0; j < 8; j++) sum += v[j] & ST_VALUE;
                }
                return sum;
            };
            const uint32_t local = gather(st, first, bid);
            uint32_t* gst = gstat + (size_t)d;
            if ((bid & ((1u << gs_log) - 1u)) == (1u << gs_log) - 1u)
                __hip_atomic_store(gst + (size_t)G * RADIX_MAX, ST_PREFIX | ((local + my_total) & ST_VALUE), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            excl = gather(gst, 0u, G) + local;
        } else {
            int64_t pb = (int64_t)bid - 1;
            while (pb >= 0) {
                uint32_t v[LB];
    #pragma unroll
                for (int j = 0; j < LB; j++)
                    v[j] = pb - j >= 0 ? __hip_atomic_load(st + (size_t)(pb - j) * RADIX_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : ST_PREFIX;  // (before tile 0: an inclusive prefix of zero)
                int used = 0;
                bool found = false;
    #pragma unroll
                for (int j = 0; j < LB; j++) {
                    if (found || used < j) continue;  // (past the prefix, or past a tile that has not published yet)
                    if ((v[j] & ST_MASK) == ST_EMPTY) continue;
                    excl += v[j] & ST_VALUE;
                    used = j + 1;
                    found = (v[j] & ST_MASK) == ST_PREFIX;
                }
                if (found) break;
                pb -= used;
                if (used > 0) spins = 0;  // (the budget is per predecessor that keeps the tile waiting, not per look-back)
                if (used < LB) {  // tile pb has not published yet: wait for it
                    if (++spins > (1u << 22)) {
                        // never hang the device: carry on with garbage, but SAY so -- in the sort's own error word and in the
                        // caller's (the frame's COUNTER_SORTERR / COUNTER_OVF: a frame sorted wrongly is treated as a truncated
                        // one, its backward writes zero gradients and the host's read-back fails the call)
                        atomicOr(error, 2u);
                        if (frame_error) atomicOr(frame_error, 2u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(st + (size_t)bid * RADIX_MAX, ST_PREFIX | ((excl + my_total) & ST_VALUE), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[d] = excl;  // + global digit base, added after the scan below
        lbase[d] = my_total;
    }
    __syncthreads();
    // exclusive scans over the digits: global digit base (from ghist) and local base (from lbase)
    if (threadIdx.x < 64) {
        // one wave scans up to 256 digits: 4 per lane
        uint32_t g[4], l[4], gs = 0, ls = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t d = threadIdx.x * 4 + k;
            g[k] = d < radix ? s_gh[d] : 0u;
            l[k] = d < radix ? lbase[d] : 0u;
            gs += g[k];
            ls += l[k];
        }
        const uint32_t gi = wave_inclusive_scan(gs, lane) - gs, li = wave_inclusive_scan(ls, lane) - ls;
        uint32_t gr = gi, lr = li;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t d = threadIdx.x * 4 + k;
            if (d < radix) {
                gbase[d] += gr;
                lbase[d] = lr;
            }
            gr += g[k];
            lr += l[k];
        }
    }
    __syncthreads();
    // ---- reorder through LDS: element -> local position lbase[d] + wave offset + rank
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        size_t i = wbase + (size_t)r * WAVE + lane;
        if (r < items && i < n) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t lp = lbase[d] + cnt[w][d] + rank[r];
            s_keys[lp] = key[r];
            s_vals[lp] = val[r];
        }
    }
    __syncthreads();
    const uint32_t count = (uint32_t)(tile_base + (size_t)TILE_KEYS <= n ? (size_t)TILE_KEYS : (n > tile_base ? n - tile_base : 0));
    for (uint32_t i = threadIdx.x; i < count; i += THREADS) {
        const uint32_t k = s_keys[i];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t pos = gbase[d] + (i - lbase[d]);
        keys_out[pos] = k;
        vals_out[pos] = s_vals[i];
    }
}

inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace

size_t scan_scratch_words(size_t n) { return div_up(n, SCAN_CHUNK) + 16; }

// the smallest tile the adaptive 512 x 16 kernel may choose for a capacity of n keys: 512 x 2 = 1024 keys, more when that
// would need more than 2048 rows of status words per pass (the table is cleared every frame: 1 KB per row and pass)
// rows of group totals behind a pass's status rows (grouped look-back: groups of at least 16 tiles)
static size_t sweep_group_rows(size_t tile_rows) { return tile_rows / 16 + 2; }
static int sweep_min_items_for(size_t n, bool small) {
    if (n <= ((size_t)2 << 20) && !small) return 4;  // (1024 x 4: not adaptive)
    int items = 2;
    while (items < 16 && (n + (size_t)512 * items - 1) / ((size_t)512 * items) > 2048) items <<= 1;
    return items;
}
static bool sweep_adaptive(size_t n) { return n > ((size_t)2 << 20) || g_options.sort_small != 0; }
static int sweep_min_items(size_t n) { return sweep_min_items_for(n, g_options.sort_small != 0); }
static size_t sweep_min_tile_keys(size_t n) { return sweep_adaptive(n) ? (size_t)512 * sweep_min_items(n) : 4096; }

size_t sort_scratch_words(size_t n) {
    size_t nblk = div_up(n, 4096);  // the smallest tile of the three-kernel variant
    size_t table = (size_t)RADIX_MAX * nblk;
    size_t three_kernel = table + scan_scratch_words(table) + 16;
    // (laid out for the smallest tile ANY setting of sort_small may choose: a workspace outlives the option)
    const size_t min_tile = (size_t)512 * sweep_min_items_for(n, true);
    const size_t rows = div_up(n, min_tile < 4096 ? min_tile : 4096);
    size_t onesweep = (size_t)MAX_PASSES * RADIX_MAX * (rows + sweep_group_rows(rows)) + GH_WORDS + 64;
    return three_kernel > onesweep ? three_kernel : onesweep;
}

void exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, size_t n, uint32_t* total,
                        uint32_t* scratch, hipStream_t s, const uint32_t* n_dev) {
    if (n == 0) {
        if (total) (void)hipMemsetAsync(total, 0, sizeof(uint32_t), s);
        return;
    }
    const size_t nb = div_up(n, SCAN_CHUNK);
    const bool stash = gather != nullptr && out != in;  // the gathered values pass through `out`
    scan_reduce_k<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, gather, n, n_dev, scratch, stash ? out : nullptr);
    // nobody wants the total and the block sums are few: the second kernel sums the ones in front of it itself (one launch less)
    const bool raw = total == nullptr && nb <= 4096;
    if (!raw) scan_partials_k<<<dim3(1), dim3(1024), 0, s>>>(scratch, nb, total);
    scan_apply_k<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(stash ? out : in, stash ? nullptr : gather, out, n, n_dev, scratch,
                                                                  raw ? 1 : 0);
}

// Control words of the onesweep sort, contiguous at the start of the scratch so that ONE memset clears them:
// [passes][nblk][256] status | [MAX_PASSES][256] global digit histograms | [MAX_PASSES] tickets | error
static size_t sweep_tile_keys(size_t n) { return sweep_adaptive(n) ? 8192 : 4096; }  // 512 x 16 (adaptive) or 1024 x 4
static size_t sweep_status_words(size_t n, int lo, int hi) {
    const size_t rows = div_up(n, sweep_min_tile_keys(n));
    return (size_t)((hi - lo + 7) / 8) * (rows + sweep_group_rows(rows)) * RADIX_MAX;
}
size_t radix_sort_control_words(size_t n, int lo, int hi) {
    return n ? sweep_status_words(n, lo, hi) + GH_WORDS + MAX_PASSES + 1 : 0;
}
uint32_t* radix_sort_ghist(uint32_t* scratch, size_t n, int lo, int hi) {
    return scratch + sweep_status_words(n, lo, hi);
}

int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], size_t n, int lo, int hi, uint32_t* scratch,
                     hipStream_t s, bool cleared, bool ghist_ready, const uint32_t* n_dev, uint32_t* frame_error) {
    int cur = 0;
    if (n == 0) return cur;
    const uint32_t nblk = (uint32_t)div_up(n, SORT_TILE);
    uint32_t* table = scratch;
    uint32_t* scan_scratch = scratch + (size_t)RADIX_MAX * nblk;
    // split [lo,hi) into the fewest passes of at most 8 bits, as evenly as possible
    const int bits = hi - lo;
    const int passes = (bits + 7) / 8;
    int shift = lo;
    if (g_options.sort_variant == 1 && passes <= MAX_PASSES) {
        // scratch: [passes][nblk][256] status words | [passes][256] global histograms | tickets | error
        uint32_t* status = scratch;
        uint32_t* ghist = radix_sort_ghist(scratch, n, lo, hi);
        uint32_t* ticket = ghist + GH_WORDS;
        uint32_t* error = ticket + MAX_PASSES;
        if (!cleared) (void)hipMemsetAsync(status, 0, radix_sort_control_words(n, lo, hi) * sizeof(uint32_t), s);
        SweepPlan plan;
        plan.passes = passes;
        int sh = lo;
        for (int p = 0; p < passes; p++) {
            plan.nbits[p] = (bits - (sh - lo) + (passes - p) - 1) / (passes - p);
            plan.shift[p] = sh;
            sh += plan.nbits[p];
        }
        for (int p = passes; p < MAX_PASSES; p++) plan.shift[p] = plan.nbits[p] = 0;
        if (!ghist_ready) sweep_hist_k<<<dim3(nblk), dim3(SORT_THREADS), 0, s>>>(keys[0], n, n_dev, plan, ghist);
        const size_t tile = sweep_tile_keys(n);
        const int min_items = sweep_min_items(n);
        // grid and status rows for the SMALLEST tile the kernel may choose (the count is on the device)
        const uint32_t nt = (uint32_t)div_up(n, sweep_min_tile_keys(n));
        const size_t lds = 2 * tile * sizeof(uint32_t);  // the tile's keys and values in digit order
        // (the attribute is per device: set it once on each device this process sorts on)
        static std::atomic<uint64_t> attr_set{0};
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        const uint64_t dev_bit = 1ull << (dev_id & 63);
        if (!(attr_set.load(std::memory_order_relaxed) & dev_bit)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_pass_k<512, 16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 * (int)sizeof(uint32_t));
            attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
        }
        const size_t prow = (size_t)nt + sweep_group_rows(nt);  // status rows of a pass: its tiles, then its groups
        for (int p = 0; p < passes; p++) {
            uint32_t* st_p = status + (size_t)p * prow * RADIX_MAX;
            uint32_t* gst_p = g_options.sort_lookback ? st_p + (size_t)nt * RADIX_MAX : nullptr;  // NULL: the chained look-back
            if (tile == 4096)
                sweep_pass_k<1024, 4><<<dim3(nt), dim3(1024), lds, s>>>(
                    keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, n_dev, plan.shift[p], plan.nbits[p],
                    ghist + (size_t)p * RADIX_MAX, st_p, g_options.sort_tickets ? ticket + p : nullptr, error, frame_error, 4, gst_p);
            else
                sweep_pass_k<512, 16><<<dim3(nt), dim3(512), lds, s>>>(
                    keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, n_dev, plan.shift[p], plan.nbits[p],
                    ghist + (size_t)p * RADIX_MAX, st_p, g_options.sort_tickets ? ticket + p : nullptr, error, frame_error, min_items, gst_p);
            cur ^= 1;
        }
        return cur;
    }
    for (int p = 0; p < passes; p++) {
        const int nbits = (bits - (shift - lo) + (passes - p) - 1) / (passes - p);
        const uint32_t mask = (1u << nbits) - 1u;
        radix_hist_k<<<dim3(nblk), dim3(SORT_THREADS), 0, s>>>(keys[cur], n, shift, mask, table, nblk);
        const size_t tab_n = (size_t)(mask + 1u) * nblk;
        exclusive_scan_u32(table, nullptr, table, tab_n, nullptr, scan_scratch, s);
        radix_scatter_k<<<dim3(nblk), dim3(SORT_THREADS), 0, s>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n,
                                                                  shift, nbits, table, nblk);
        cur ^= 1;
        shift += nbits;
    }
    return cur;
}

}  // namespace goi
