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
#include "common.h"

namespace goi {

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;  // 2048

constexpr int SORT_THREADS = 256;
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

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_k(const uint32_t* __restrict__ in,
                                                              const uint32_t* __restrict__ gather, size_t n,
                                                              uint32_t* __restrict__ partials) {
    __shared__ uint32_t sm[SCAN_THREADS / WAVE + 1];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + (size_t)k * SCAN_THREADS + threadIdx.x;
        if (i < n) sum += gather ? in[gather[i]] : in[i];
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
                                                             uint32_t* out, size_t n,
                                                             const uint32_t* __restrict__ partials) {
    __shared__ uint32_t sm[SCAN_THREADS / WAVE + 1];
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
    uint32_t tot;
    uint32_t run = block_exclusive_scan<SCAN_THREADS>(sum, sm, &tot) + partials[blockIdx.x];
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
    for (uint32_t d = threadIdx.x; d <= mask; d += SORT_THREADS)
        hist[(size_t)d * nblk + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
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

inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace

size_t scan_scratch_words(size_t n) { return div_up(n, SCAN_CHUNK) + 16; }

size_t sort_scratch_words(size_t n) {
    size_t nblk = div_up(n, SORT_TILE);
    size_t table = (size_t)RADIX_MAX * nblk;
    return table + scan_scratch_words(table) + 16;
}

void exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, size_t n, uint32_t* total,
                        uint32_t* scratch, hipStream_t s) {
    if (n == 0) {
        if (total) hipMemsetAsync(total, 0, sizeof(uint32_t), s);
        return;
    }
    const size_t nb = div_up(n, SCAN_CHUNK);
    scan_reduce_k<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, gather, n, scratch);
    scan_partials_k<<<dim3(1), dim3(1024), 0, s>>>(scratch, nb, total);
    scan_apply_k<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, gather, out, n, scratch);
}

int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], size_t n, int lo, int hi, uint32_t* scratch,
                     hipStream_t s) {
    int cur = 0;
    if (n == 0) return cur;
    const uint32_t nblk = (uint32_t)div_up(n, SORT_TILE);
    uint32_t* table = scratch;
    uint32_t* scan_scratch = scratch + (size_t)RADIX_MAX * nblk;
    // split [lo,hi) into the fewest passes of at most 8 bits, as evenly as possible
    const int bits = hi - lo;
    const int passes = (bits + 7) / 8;
    int shift = lo;
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
