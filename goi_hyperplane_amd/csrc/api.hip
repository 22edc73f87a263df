// extern "C" entry points of libgoi_raster.so (see include/goi_raster.h): workspace layout,
// stage orchestration on the caller's HIP stream, per-stage event timing.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.h"

namespace goi {

namespace {

thread_local std::string g_err;

int fail(const char* where, hipError_t e) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return -1;
}
int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

#define GOI_HIP(call)                                       \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess) return fail(#call, e__);     \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T>
inline void carve(char*& p, T*& out, size_t count) {
    p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p), 256));
    out = reinterpret_cast<T*>(p);
    p += count * sizeof(T);
}

// ---- per-stage timing -------------------------------------------------------------------------
struct StageEvents {
    int stage;
    hipEvent_t a, b;
};
// (process-wide and shared by every calling thread: the mask is atomic, the event lists sit behind a mutex.  Stage timing
// is a single-device measurement aid: events are created on whichever device is current when they are first needed.)
std::atomic<unsigned> g_profile_mask{0};  // bit i = record HIP events around GOI_STAGE_i
std::mutex g_profile_mu;
std::vector<StageEvents> g_events;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    {
        std::lock_guard<std::mutex> lk(g_profile_mu);
        if (!g_pool.empty()) {
            hipEvent_t e = g_pool.back();
            g_pool.pop_back();
            return e;
        }
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct StageTimer {
    hipStream_t s;
    bool on;
    StageEvents ev{};
    StageTimer(int stage, hipStream_t st) : s(st), on((g_profile_mask.load(std::memory_order_relaxed) >> stage) & 1u) {
        if (on) {
            ev.stage = stage;
            ev.a = get_event();
            ev.b = get_event();
            (void)hipEventRecord(ev.a, s);
        }
    }
    ~StageTimer() {
        if (on) {
            (void)hipEventRecord(ev.b, s);
            std::lock_guard<std::mutex> lk(g_profile_mu);
            g_events.push_back(ev);
        }
    }
};

int check_stage(const GoiRasterScene& sc, hipStream_t s, const char* name) {
    if (!sc.debug) return 0;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(name, e);
    return 0;
}

int validate(const GoiRasterScene* sc, bool need_sem, bool need_opacity = true) {
    if (!sc) return fail("scene is NULL");
    if (sc->P < 0 || sc->W <= 0 || sc->H <= 0) return fail("bad P/W/H");
    if (sc->S < 1 || sc->S > 32) return fail("semantic channels S must be in 1..32");
    if (sc->P == 0) return 0;
    if (need_opacity && !sc->opacities) return fail("opacities is NULL");
    if (!sc->means3D || !sc->viewmatrix || !sc->projmatrix || !sc->campos || !sc->bg)
        return fail("a required input pointer is NULL");
    if (need_sem && !sc->semantics) return fail("semantics is required (the reference dereferences it unconditionally)");
    if (sc->semantics && (sc->S & 3) == 0 && (reinterpret_cast<uintptr_t>(sc->semantics) & 15u) != 0)
        return fail("semantics must be 16-byte aligned when S is a multiple of 4 (its rows are moved as 16-byte words)");
    if ((sc->shs == nullptr) == (sc->colors_precomp == nullptr))
        return fail("Please provide excatly one of either SHs or precomputed colors!");
    // (an SH row whose 3 M floats are a whole number of 16-byte words -- M = 4, 8, 12, 16 -- is moved as such by both per-Gaussian
    // kernels: preprocess.hip, row16 / the backward's hoist)
    if (sc->shs && (3 * sc->M) % 4 == 0 && (reinterpret_cast<uintptr_t>(sc->shs) & 15u) != 0)
        return fail("shs must be 16-byte aligned when 3 M is a multiple of 4 (its rows are moved as 16-byte words)");
    if (((sc->scales == nullptr || sc->rotations == nullptr) && sc->cov3D_precomp == nullptr) ||
        ((sc->scales != nullptr || sc->rotations != nullptr) && sc->cov3D_precomp != nullptr))
        return fail("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    // M <= 16: the kernels hold at most the 16 coefficients of degree 3, and preprocess_bwd stages 256 x (3M+1) floats
    // of dL/dSH in LDS (50 KB at M = 16) -- a larger M would only fail at backward time
    if (sc->shs && (sc->D < 0 || sc->D > 3 || sc->M < (sc->D + 1) * (sc->D + 1) || sc->M > 16))
        return fail("SH degree must be 0..3 and (D+1)^2 <= M <= 16");
    return 0;
}

static inline size_t round_up_256(size_t n) { return (n + 255) & ~(size_t)255; }

// ---- read-back tickets ---------------------------------------------------------------------------
// num_rendered (and the "prefiltered" error flag) reach the host through a pinned copy of the frame's counters plus an
// event recorded right behind the copy.  A ticket is one such (pinned words, event) pair; they are pooled per DEVICE
// (events and pinned allocations belong to the device that was current when they were made) under a mutex, so
// concurrent callers on different streams / threads / GPUs each get their own.
constexpr int READBACK_HEAD_WORDS_MIN = 32;
struct Ticket {
    int dev = -1;
    uint32_t* pinned = nullptr;
    hipEvent_t ev = nullptr;
    bool busy = false;
    int capacity = 0;  // instances the frame's binning buffer holds (speculative frames); 0: exact frame
    bool head_only = false;  // only the first READBACK_HEAD_WORDS words were copied: num_rendered is counters[COUNTER_N]
    uint32_t* pinned_dev = nullptr;  // the same words as the device sees them (the pinned allocation is mapped)
    // STAMPED read-back (round 5): no copy and no event -- a wave of the forward blend stores the head words into the pinned
    // words and then, behind a system-scope fence, this use's sequence number into pinned[STAMP_WORD]; the host compares.
    // (An event recorded behind the blend cost the stream a ~5.6 us bubble per frame: rocprofv3 shows it as the only idle gap
    // of a training step, between render_fwd_k and the backward's first kernel.)
    uint32_t seq = 0;
    bool stamped = false;
    hipStream_t stream = nullptr;  // the stream the frame was enqueued on (the slow path of a blocking wait drains IT, nothing else)
};
constexpr int STAMP_WORD = HOST_STAMP_WORD;  // 33 (common.h: the blend kernels write it); the head is words 0..31, word 32 is a
                                            // num_rendered stripe of the full copy, which is never stamped
static_assert(STAMP_WORD >= READBACK_HEAD_WORDS_MIN && STAMP_WORD < COUNTER_WORDS, "stamp word inside the pinned block, outside the head");
// A read-back queued BEHIND the whole frame (speculative forward) finds num_rendered as one word (COUNTER_N, left by the
// listed-Gaussian compaction / the scan): 128 bytes travel instead of the 4 KB of striped partial counters -- which the
// runtime moved as three copy kernels per frame.
constexpr int READBACK_HEAD_WORDS = 32;
static_assert(COUNTER_N < READBACK_HEAD_WORDS && COUNTER_OVF < READBACK_HEAD_WORDS && NR_BASE >= READBACK_HEAD_WORDS, "head layout");
std::mutex g_ticket_mu;
std::vector<Ticket> g_tickets;
constexpr int MAX_TICKETS = 4096;

int ticket_acquire(int capacity) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail("hipGetDevice failed");
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    for (size_t i = 0; i < g_tickets.size(); i++)
        if (!g_tickets[i].busy && g_tickets[i].dev == dev) {
            g_tickets[i].busy = true;
            g_tickets[i].capacity = capacity;
            g_tickets[i].stamped = false;
            return (int)i;
        }
    if ((int)g_tickets.size() >= MAX_TICKETS)
        return fail("too many unresolved speculative forwards (goi_raster_ticket_result was never called for them)");
    Ticket t;
    t.dev = dev;
    GOI_HIP(hipHostMalloc(reinterpret_cast<void**>(&t.pinned), COUNTER_WORDS * sizeof(uint32_t),
                          hipHostMallocMapped | hipHostMallocCoherent));  // (coherent: a kernel's stores go straight to the host)
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&t.pinned_dev), t.pinned, 0) != hipSuccess) t.pinned_dev = nullptr;
    GOI_HIP(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    t.pinned[STAMP_WORD] = 0u;
    t.busy = true;
    t.capacity = capacity;
    g_tickets.push_back(t);
    return (int)g_tickets.size() - 1;
}

void ticket_release(int id) {
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    if (id >= 0 && id < (int)g_tickets.size()) g_tickets[id].busy = false;
}

// wait != 0: block until the frame's counters have arrived.  Returns 1 (done: *n = num_rendered), 0 (not yet; only
// when wait == 0) or -1 (error; the ticket is released).  A finished ticket is released.
int ticket_result(int id, int wait, long long* n, unsigned* frame_flags = nullptr) {
    Ticket t;
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        if (id < 0 || id >= (int)g_tickets.size() || !g_tickets[id].busy) return fail("invalid or already resolved ticket");
        t = g_tickets[id];
    }
    if (t.stamped) {
        // the blend's wave stores the head words, fences at system scope, then stores the stamp: an acquire load that sees
        // this use's sequence number sees the words
        auto arrived = [&]() { return __atomic_load_n(&t.pinned[STAMP_WORD], __ATOMIC_ACQUIRE) == t.seq; };
        if (!arrived()) {
            if (!wait) return 0;
            // spin politely; a frame is milliseconds.  After two seconds something is wrong with the device (or the frame
            // never ran): drain the FRAME'S STREAM and look once more -- never hang in here.  (Not the device: that would also
            // wait for every other stream, e.g. an RCCL collective whose peer needs this very host thread to make progress.)
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (!arrived()) {
                if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                    int cur = 0;
                    (void)hipGetDevice(&cur);
                    (void)hipSetDevice(t.dev);
                    const hipError_t e = hipStreamSynchronize(t.stream);
                    (void)hipSetDevice(cur);
                    if (e != hipSuccess || !arrived()) {
                        ticket_release(id);
                        return fail("the frame's counters never arrived (num_rendered read-back)", e);
                    }
                }
            }
        }
    } else if (wait) {
        hipError_t e = hipEventSynchronize(t.ev);
        if (e != hipSuccess) {
            ticket_release(id);
            return fail("hipEventSynchronize(num_rendered read-back)", e);
        }
    } else {
        hipError_t e = hipEventQuery(t.ev);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) {
            ticket_release(id);
            return fail("hipEventQuery(num_rendered read-back)", e);
        }
    }
    unsigned long long total = 0;  // 64-bit: 32 stripes of up to 2^32-1 each
    if (t.head_only)
        total = t.pinned[COUNTER_N];
    else
        for (int i = 0; i < NR_STRIPES; i++) total += t.pinned[NR_BASE + NR_STRIDE * i];
    const bool filtered = t.pinned[1] != 0;
    // (a read-back queued behind the whole frame also carries the verdict of both sorts)
    const bool missorted = t.head_only && (t.pinned[COUNTER_SORTERR] != 0 || (t.pinned[COUNTER_OVF] & OVF_MISSORTED) != 0);
    if (frame_flags) *frame_flags = t.head_only ? t.pinned[COUNTER_OVF] : 0u;
    ticket_release(id);
    if (filtered) return fail("Point is filtered although prefiltered is set. This shouldn't happen!");
    if (missorted)
        return fail("a look-back of the radix sort timed out (preempted or shared GPU?): the frame's lists are not sorted; its "
                    "backward writes zero gradients, its image must not be used");
    if (total > 0x7FFFFFFFull) return fail("num_rendered overflows int32");
    *n = (long long)total;
    return 1;
}

// Front half of the forward -- everything that does not depend on num_rendered: preprocess, the read-back of the
// counters (queued BEFORE the depth sort: preprocess has already summed num_rendered, so a host that waits for it wakes
// up while the GPU is still sorting), depth sort, scan.  The scan also leaves num_rendered in counters[COUNTER_N] for
// the kernels of the back half.
int enqueue_readback(const GeomView& g, int ticket, hipStream_t s, bool head_only = false) {
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    Ticket& t = g_tickets[ticket];
    t.head_only = head_only;
    GOI_HIP(hipMemcpyAsync(t.pinned, g.counters, (head_only ? READBACK_HEAD_WORDS : COUNTER_WORDS) * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, s));
    GOI_HIP(hipEventRecord(t.ev, s));
    return 0;
}

// ticket < 0: no read-back here (the speculative forward queues it behind the blend instead: nobody is waiting for it,
// and a device-to-host copy in the middle of the frame costs the stream a ~10 us bubble)
int enqueue_front(const GoiRasterScene& sc, GeomView& g, ImageView& im, int* radii, int ticket, const uint32_t** order_out,
                  hipStream_t s, const float* zcut = nullptr, uint32_t* zlearn = nullptr) {
    const int P = sc.P;
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    // one memset: the counters and, right behind them, the control words of the depth sort
    const size_t depth_ctrl = g_options.sort_variant == 1 ? radix_sort_control_words((size_t)P, 0, 32) : 0;
    // (sizes rounded up to 256 bytes: the runtime splits a memset of any other size into two fill kernels; the few
    // extra words are scratch that is written before it is read)
    GOI_HIP(hipMemsetAsync(g.blk_coarse, 0,
                           round_up_256((size_t)(reinterpret_cast<char*>(g.scratch + depth_ctrl) -
                                                 reinterpret_cast<char*>(g.blk_coarse))), s));
    {
        StageTimer t(GOI_STAGE_PREPROCESS, s);
        launch_preprocess_fwd(sc, g, radii, im.ranges, gx * gy, s, zcut, zlearn);  // also zeroes the tile ranges
    }
    if (check_stage(sc, s, "preprocess")) return -1;
    if (ticket >= 0 && enqueue_readback(g, ticket, s)) return -1;
    // Only the V LISTED Gaussians (tiles_touched > 0; half of the headline scene is culled) are depth-sorted: one small
    // pass compacts their (depth key, id) in id order -- the sort must stay stable -- and, reading the keys anyway, forms
    // the sort's digit histograms (its prologue kernel is skipped).  V lives on the device: grids cover P, the kernels stop
    // at counters[COUNTER_V]; the prefix sum over the depth order and emit walk the V listed Gaussians only.
    const bool onesweep = g_options.sort_variant == 1;
    const uint32_t* v_dev = g.counters + COUNTER_V;
    int order_idx;
    {
        StageTimer t(GOI_STAGE_DEPTH_SORT, s);
        launch_compact_listed(P, g, onesweep ? radix_sort_ghist(g.scratch, (size_t)P, 0, 32) : nullptr, /*pad=*/!onesweep, s);
        order_idx = radix_sort_pairs(g.sort_keys, g.sort_vals, (size_t)P, 0, 32, g.scratch, s, /*cleared=*/true,
                                     /*ghist_ready=*/onesweep, onesweep ? v_dev : nullptr, g.counters + COUNTER_SORTERR);
    }
    if (check_stage(sc, s, "depth sort")) return -1;
    *order_out = g.sort_vals[order_idx];
    {
        StageTimer t(GOI_STAGE_SCAN, s);
        exclusive_scan_u32(g.tiles_touched, *order_out, g.offsets, (size_t)P, nullptr, g.scratch, s, v_dev);
    }
    return 0;
}

// Back half: emit -> tile ranges -> tile sort, for a binning buffer that holds `cap` instances.  exact: cap IS
// num_rendered (known to the host); otherwise cap is a capacity and every kernel takes the count from
// counters[COUNTER_N], clamped to cap (an overflowed frame is truncated but memory-safe; the host redoes it).
int enqueue_back(const GoiRasterScene& sc, GeomView& g, ImageView& im, const BinView& bv_in, int cap, bool exact,
                 const uint32_t* order, const int* radii, const uint32_t** plist, hipStream_t s) {
    BinView bv = bv_in;  // (the caller hands bv_in.qmask to the blend itself)
    const int P = sc.P;
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    const uint32_t* n_dev = exact ? nullptr : g.counters + COUNTER_N;
    // Tile counting fused into emit (onesweep sort, tile grid small enough for an LDS histogram): the counts give
    // the tile ranges and the sort's digit histograms, so neither the keys nor the sorted keys are re-read for them.
    const bool counting = cap > 0 && g_options.sort_variant == 1 && emit_can_count_tiles(sc.W, sc.H);
    const int tile_bits = tile_key_bits((uint32_t)(gx * gy));
    {
        StageTimer t(GOI_STAGE_EMIT, s);
        if (counting) {
            // emit also clears the control words of the tile sort (status words, histograms, tickets): one launch less
            launch_emit_counting(P, sc.W, sc.H, g, order, radii, bv.keys[0], bv.vals[0], im.ranges, bv.scratch,
                                 radix_sort_control_words((size_t)cap, 0, tile_bits), (uint32_t)cap, s);
        }
        else if (cap > 0)
            launch_emit(P, sc.W, sc.H, g, order, radii, bv.keys[0], bv.vals[0], (uint32_t)cap, s);
    }
    if (check_stage(sc, s, "emit")) return -1;
    int fin;
    if (counting) {
        {
            StageTimer t(GOI_STAGE_RANGES, s);
            launch_tile_ranges_hist(sc.W, sc.H, im.ranges, radix_sort_ghist(bv.scratch, (size_t)cap, 0, tile_bits), g.counters, s);
        }
        StageTimer t(GOI_STAGE_TILE_SORT, s);
        fin = radix_sort_pairs(bv.keys, bv.vals, (size_t)cap, 0, tile_bits, bv.scratch, s, /*cleared=*/true,
                               /*ghist_ready=*/true, n_dev, g.counters + COUNTER_OVF);
    } else {
        {
            StageTimer t(GOI_STAGE_TILE_SORT, s);
            fin = radix_sort_pairs(bv.keys, bv.vals, (size_t)cap, 0, tile_bits, bv.scratch, s, false, false, n_dev,
                                   g.counters + COUNTER_OVF);
        }
        if (check_stage(sc, s, "tile sort")) return -1;
        StageTimer t(GOI_STAGE_RANGES, s);
        launch_ranges(cap, n_dev, bv.keys[fin], im.ranges, gx * gy, s);
    }
    if (check_stage(sc, s, "tile sort / ranges")) return -1;
    *plist = bv.vals[fin];
    return 0;
}

// The exact front end shared by goi_raster_forward and goi_raster_trace: the host waits for num_rendered (the one
// read-back of the reference, CR/rasterizer_impl.cu:285), sizes the binning workspace through the allocation callback
// and enqueues the back half while the GPU is still busy with the depth sort and the scan -- no idle gap on the device.
// Returns num_rendered (>= 0) and the final point list through *plist.
int geometry_and_binning(const GoiRasterScene& sc, GeomView& g, ImageView& im, goi_alloc_fn alloc, void* user,
                         int* radii, const uint32_t** plist, hipStream_t s, unsigned long long** qmask = nullptr) {
    const int ticket = ticket_acquire(0);
    if (ticket < 0) return -1;
    const uint32_t* order = nullptr;
    if (enqueue_front(sc, g, im, radii, ticket, &order, s)) {
        ticket_release(ticket);
        return -1;
    }
    long long n64 = 0;
    if (ticket_result(ticket, 1, &n64) < 0) return -1;
    const int N = (int)n64;
    const size_t need = goi_raster_binning_bytes(N);
    char* bin_mem = static_cast<char*>(alloc(user, need));
    if (!bin_mem && need > 0) return fail("binning allocation callback returned NULL");
    BinView bv;
    binning_layout(N, bin_mem, &bv);
    if (qmask) *qmask = bv.qmask;
    if (enqueue_back(sc, g, im, bv, N, /*exact=*/true, order, radii, plist, s)) return -1;
    return N;
}

}  // namespace

thread_local Options g_options;
static Options g_shared_options;
static std::mutex g_options_mu;
void refresh_options() {
    std::lock_guard<std::mutex> lk(g_options_mu);
    g_options = g_shared_options;
}

// Which ping-pong buffer holds the tile-sorted list: a pure function of the pass count, so the
// backward can recompute it instead of storing it.
static int tile_sort_result_index(int W, int H, int N) {
    if (N <= 0) return 0;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int passes = (tile_key_bits((uint32_t)(gx * gy)) + 7) / 8;
    return passes & 1;
}

size_t geom_layout(int P, char* base, GeomView* v) {
    char* p = base;
    const size_t n = (size_t)(P > 0 ? P : 1);
    GeomView tmp;
    GeomView& g = v ? *v : tmp;
    carve(p, g.rec, n);
    carve(p, g.cov3D, 6 * n);
    carve(p, g.tiles_touched, n);
    carve(p, g.clamped, n);
    carve(p, g.sort_keys[0], n);
    carve(p, g.sort_keys[1], n);
    carve(p, g.sort_vals[0], n);
    carve(p, g.sort_vals[1], n);
    g.bigq = g.sort_keys[1];  // (the raw depth keys are consumed by the compaction, the sort's result is in buffer 0)
    carve(p, g.offsets, n);
    carve(p, g.aux, n);
    carve(p, g.blk_agg, (n + PRE_BLOCK - 1) / PRE_BLOCK);
    // (blk_coarse, counters, sort control words: contiguous, ONE memset clears the three)
    carve(p, g.blk_coarse, (((n + PRE_BLOCK - 1) / PRE_BLOCK + COARSE_BLOCKS - 1) / COARSE_BLOCKS) * (size_t)COARSE_STRIDE);
    carve(p, g.counters, COUNTER_WORDS);  // directly in front of the sort scratch: one memset clears both
    g.scratch_words = sort_scratch_words(n) + scan_scratch_words(n);
    carve(p, g.scratch, g.scratch_words);
    return (size_t)(p - base) + 256;
}

size_t image_layout(int W, int H, char* base, ImageView* v) {
    char* p = base;
    ImageView tmp;
    ImageView& im = v ? *v : tmp;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    carve(p, im.n_contrib, (size_t)W * H);
    carve(p, im.ranges, (size_t)gx * gy);
    carve(p, im.qcost, (size_t)gx * gy * 4);
    carve(p, im.qorder, (size_t)gx * gy * 4 + 8);
    carve(p, im.qmask0, (size_t)gx * gy * 4);
    return (size_t)(p - base) + 256;
}

size_t bwd_scratch_layout(int N, int S, char* base, BwdScratchView* v) {
    char* p = base;
    const size_t n = (size_t)(N > 0 ? N : 1) * 4;
    BwdScratchView tmp;
    BwdScratchView& b = v ? *v : tmp;
    carve(p, b.rows, n * (size_t)bwd_row_floats(S));
    carve(p, b.flags, n);
    // big Gaussians (more than REDUCE_BIG_INST instances each): at most N / REDUCE_BIG_INST of them
    b.cap_big = n / 4 / REDUCE_BIG_INST + 2;
    carve(p, b.big_ctl, 8);
    carve(p, b.big_desc, b.cap_big);
    return (size_t)(p - base) + 256;
}

size_t binning_layout(int N, char* base, BinView* v) {
    char* p = base;
    const size_t n = (size_t)(N > 0 ? N : 1);
    BinView tmp;
    BinView& b = v ? *v : tmp;
    carve(p, b.keys[0], n);
    carve(p, b.keys[1], n);
    carve(p, b.vals[0], n);
    carve(p, b.vals[1], n);
    b.scratch_words = sort_scratch_words(n) + 8;
    carve(p, b.scratch, b.scratch_words);
    carve(p, b.qmask, 4 * (n / 64 + 2));
    return (size_t)(p - base) + 256;
}

}  // namespace goi

using namespace goi;

extern "C" {

int goi_raster_abi_version(void) { return GOI_RASTER_ABI_VERSION; }
const char* goi_raster_last_error(void) { return g_err.c_str(); }

// Sizes are computed with a base that already has the strictest alignment, so they are upper
// bounds for any 256-byte aligned buffer; +256 slack covers callers that hand over less.
size_t goi_raster_geom_bytes(int P) { return geom_layout(P, nullptr, nullptr) + 256; }
size_t goi_raster_image_bytes(int W, int H) { return image_layout(W, H, nullptr, nullptr) + 256; }
size_t goi_raster_binning_bytes(int N) { return binning_layout(N, nullptr, nullptr) + 256; }
size_t goi_raster_backward_scratch_bytes(int N, int S) { return bwd_scratch_layout(N, S, nullptr, nullptr) + 256; }

int goi_raster_forward(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer, goi_alloc_fn binning_alloc,
                       void* alloc_user, float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                       int* radii, void* stream) {
    refresh_options();
    if (validate(scene, true)) return -1;
    const GoiRasterScene& sc = *scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)sc.W * sc.H;
    if (sc.P == 0) {  // zero-filled outputs, nothing launched (DGR/rasterize_points.cu:84-85)
        GOI_HIP(hipMemsetAsync(out_color, 0, 3 * HW * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(out_semantic, 0, (size_t)sc.S * HW * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(out_depth, 0, HW * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(out_alpha, 0, HW * sizeof(float), s));
        return 0;
    }
    if (!geom_buffer || !image_buffer || !binning_alloc) return fail("workspace pointer / allocator is NULL");
    GeomView g;
    ImageView im;
    geom_layout(sc.P, static_cast<char*>(geom_buffer), &g);
    image_layout(sc.W, sc.H, static_cast<char*>(image_buffer), &im);
    const uint32_t* plist = nullptr;
    unsigned long long* qmask = nullptr;
    const int N = geometry_and_binning(sc, g, im, binning_alloc, alloc_user, radii, &plist, s, &qmask);
    if (N < 0) return -1;
    {
        StageTimer t(GOI_STAGE_BLEND_FWD, s);
        launch_render_fwd(sc, g, im, plist, out_color, out_semantic, out_depth, out_alpha, s, qmask);
    }
    if (check_stage(sc, s, "forward blend")) return -1;
    GOI_HIP(hipGetLastError());
    return N;
}

int goi_raster_forward_async(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer, void* binning_buffer,
                             int capacity, float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                             int* radii, void* stream) {
    return goi_raster_forward_async_cut(scene, geom_buffer, image_buffer, binning_buffer, capacity, out_color, out_semantic,
                                        out_depth, out_alpha, radii, nullptr, nullptr, stream);
}

int goi_raster_forward_async_cut(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer, void* binning_buffer,
                                 int capacity, float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                                 int* radii, const float* zcut_in, float* zcut_out, void* stream) {
    refresh_options();
    if (validate(scene, true)) return -1;
    const GoiRasterScene& sc = *scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sc.P == 0) return fail("goi_raster_forward_async: P == 0 has nothing to speculate on; use goi_raster_forward");
    if (sc.debug) return fail("goi_raster_forward_async: debug mode synchronises after every stage; use goi_raster_forward");
    if (g_options.sort_variant != 1) return fail("goi_raster_forward_async needs the onesweep sort (sort_variant 1)");
    if (capacity <= 0) return fail("goi_raster_forward_async: capacity must be positive");
    if (!geom_buffer || !image_buffer || !binning_buffer) return fail("workspace pointer is NULL");
    GeomView g;
    ImageView im;
    BinView bv;
    geom_layout(sc.P, static_cast<char*>(geom_buffer), &g);
    image_layout(sc.W, sc.H, static_cast<char*>(image_buffer), &im);
    binning_layout(capacity, static_cast<char*>(binning_buffer), &bv);
    const int ticket = ticket_acquire(capacity);
    if (ticket < 0) return -1;
    const uint32_t* order = nullptr;
    const uint32_t* plist = nullptr;
    uint32_t* zlearn = reinterpret_cast<uint32_t*>(zcut_out);  // (positive floats: the kernels take their maximum as integers)
    if (enqueue_front(sc, g, im, radii, /*ticket=*/-1, &order, s, zcut_in, zlearn) ||
        enqueue_back(sc, g, im, bv, capacity, /*exact=*/false, order, radii, &plist, s)) {
        ticket_release(ticket);
        return -1;
    }
    // The frame's counters reach the host without a copy of their own: a wave of the forward blend stores the 32 head words
    // into the ticket's pinned (device-mapped) words and then, behind a system-scope fence, the ticket's sequence number
    // (STAMP_WORD): the ticket resolves when that wave has run -- i.e. when the blend has STARTED, not when the frame's kernels
    // have finished (the counters are final before the blend starts; a caller that needs the images synchronises the stream as
    // for any other kernel).  The runtime moved a 128-byte device-to-host copy as two copy kernels per frame, and an event behind
    // the blend cost the stream a bubble.  A frame whose lists were CUT, or that learns a cut, may still raise
    // its flag inside the blend (the cut is checked there, whether or not zcut_out was given), so it keeps the copy behind the
    // kernel: the words a blend wave stores are a snapshot taken when the kernel STARTS.
    uint32_t* host_words = nullptr;
    uint32_t stamp = 0;
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        Ticket& tk = g_tickets[ticket];
        if (!zlearn && !zcut_in && tk.pinned_dev) {
            host_words = tk.pinned_dev;
            tk.seq = tk.seq + 1u ? tk.seq + 1u : 1u;  // (never 0: a fresh ticket's stamp word)
            stamp = tk.seq;
            tk.head_only = true;
            tk.stamped = true;  // no copy, no event: ticket_result compares pinned[STAMP_WORD] with seq
            tk.stream = s;
        }
    }
    {
        StageTimer t(GOI_STAGE_BLEND_FWD, s);
        launch_render_fwd(sc, g, im, plist, out_color, out_semantic, out_depth, out_alpha, s, bv.qmask, zcut_in, zlearn,
                          host_words, stamp);
    }
    if (host_words) {
    } else if (enqueue_readback(g, ticket, s, /*head_only=*/true)) {
        ticket_release(ticket);
        return -1;
    }
    if (hipGetLastError() != hipSuccess) {
        ticket_release(ticket);
        return fail("goi_raster_forward_async: a launch failed");
    }
    return ticket;
}

int goi_raster_ticket_result(int ticket, int wait, int* num_rendered) {
    return goi_raster_ticket_result2(ticket, wait, num_rendered, nullptr);
}

int goi_raster_ticket_result2(int ticket, int wait, int* num_rendered, unsigned* frame_flags) {
    long long n = 0;
    unsigned fl = 0;
    const int r = ticket_result(ticket, wait, &n, &fl);
    if (r == 1 && num_rendered) *num_rendered = (int)n;
    if (r == 1 && frame_flags) *frame_flags = fl;
    return r;
}

int goi_raster_forward_redo(const GoiRasterScene* scene, int num_rendered, void* geom_buffer, void* image_buffer,
                            void* binning_buffer, float* out_color, float* out_semantic, float* out_depth,
                            float* out_alpha, const int* radii, void* stream) {
    refresh_options();
    // (the back half of a frame reads P, S, W, H, the semantic rows and the background from the scene; everything else
    // comes from the geometry workspace of the first attempt, so only those fields are checked)
    if (!scene) return fail("scene is NULL");
    const GoiRasterScene& sc = *scene;
    if (sc.P <= 0 || sc.W <= 0 || sc.H <= 0 || sc.S < 1 || sc.S > 32) return fail("goi_raster_forward_redo: bad P/W/H/S");
    if (!sc.semantics || !sc.bg || !radii) return fail("goi_raster_forward_redo: semantics, bg and radii are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (num_rendered < 0) return fail("goi_raster_forward_redo: nothing to redo");
    if (!geom_buffer || !image_buffer || (num_rendered > 0 && !binning_buffer)) return fail("workspace pointer is NULL");
    GeomView g;
    ImageView im;
    BinView bv;
    geom_layout(sc.P, static_cast<char*>(geom_buffer), &g);
    image_layout(sc.W, sc.H, static_cast<char*>(image_buffer), &im);
    binning_layout(num_rendered, static_cast<char*>(binning_buffer), &bv);
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    // the geometry state of the first attempt (records, depth order, offsets) does not depend on the capacity and is
    // still in the geometry buffer; the per-tile counts the first emit accumulated are not wanted
    GOI_HIP(hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s));
    const uint32_t* order = g.sort_vals[depth_sort_result_index()];
    const uint32_t* plist = nullptr;
    if (enqueue_back(sc, g, im, bv, num_rendered, /*exact=*/true, order, radii, &plist, s)) return -1;
    {
        StageTimer t(GOI_STAGE_BLEND_FWD, s);
        launch_render_fwd(sc, g, im, plist, out_color, out_semantic, out_depth, out_alpha, s, bv.qmask);
    }
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_forward_reblend(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                               const void* cached_image_buffer, void* image_buffer, float* out_color, float* out_semantic,
                               float* out_depth, float* out_alpha, void* stream) {
    refresh_options();
    // (only the blend runs: it reads P, S, W, H, the semantic rows and the background from the scene; the Gaussian records,
    // the tile lists and the tile ranges are those of the frame that filled the workspaces)
    if (!scene) return fail("scene is NULL");
    const GoiRasterScene& sc = *scene;
    if (sc.P <= 0 || sc.W <= 0 || sc.H <= 0 || sc.S < 1 || sc.S > 32) return fail("goi_raster_forward_reblend: bad P/W/H/S");
    if (!sc.semantics || !sc.bg) return fail("goi_raster_forward_reblend: semantics and bg are required");
    if (R < 0) return fail("goi_raster_forward_reblend: bad R");
    if (!geom_buffer || !cached_image_buffer || !image_buffer || (R > 0 && !binning_buffer))
        return fail("workspace pointer is NULL");
    if (!out_color || !out_semantic || !out_depth || !out_alpha) return fail("output pointer is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GeomView g;
    ImageView im_old, im;
    BinView bv;
    geom_layout(sc.P, static_cast<char*>(const_cast<void*>(geom_buffer)), &g);
    image_layout(sc.W, sc.H, static_cast<char*>(const_cast<void*>(cached_image_buffer)), &im_old);
    image_layout(sc.W, sc.H, static_cast<char*>(image_buffer), &im);
    binning_layout(R, static_cast<char*>(const_cast<void*>(binning_buffer)), &bv);
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    // the new frame gets its own image state (n_contrib is written by the blend and read by ITS backward); the tile ranges
    // are the cached frame's
    GOI_HIP(hipMemcpyAsync(im.ranges, im_old.ranges, sizeof(uint2) * (size_t)gx * gy, hipMemcpyDeviceToDevice, s));
    const uint32_t* plist = bv.vals[tile_sort_result_index(sc.W, sc.H, R)];
    {
        // (the member masks of rounds >= 1 go into the CACHED binning workspace: they depend on geometry and tile lists only,
        // so every reblend of this camera writes the words that are already there)
        StageTimer t(GOI_STAGE_BLEND_FWD, s);
        launch_render_fwd(sc, g, im, plist, out_color, out_semantic, out_depth, out_alpha, s, bv.qmask);
    }
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_trace(const GoiRasterScene* scene, const float* img_sem, void* geom_buffer, void* image_buffer,
                     goi_alloc_fn binning_alloc, void* alloc_user, float* out_color, float* gau_sem, int* num_gsem,
                     int* radii, void* stream) {
    refresh_options();
    if (validate(scene, false)) return -1;
    const GoiRasterScene& sc = *scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)sc.W * sc.H;
    GOI_HIP(hipMemsetAsync(out_color, 0, 3 * HW * sizeof(float), s));
    if (sc.P == 0) return 0;
    if (!img_sem) return fail("img_sem is NULL");
    if (!geom_buffer || !image_buffer || !binning_alloc) return fail("workspace pointer / allocator is NULL");
    GOI_HIP(hipMemsetAsync(gau_sem, 0, (size_t)sc.P * sc.S * sizeof(float), s));
    GOI_HIP(hipMemsetAsync(num_gsem, 0, (size_t)sc.P * sizeof(int), s));
    GeomView g;
    ImageView im;
    geom_layout(sc.P, static_cast<char*>(geom_buffer), &g);
    image_layout(sc.W, sc.H, static_cast<char*>(image_buffer), &im);
    const uint32_t* plist = nullptr;
    const int N = geometry_and_binning(sc, g, im, binning_alloc, alloc_user, radii, &plist, s);
    if (N < 0) return -1;
    launch_trace_fwd(sc, img_sem, g, im, plist, out_color, gau_sem, num_gsem, s);
    if (check_stage(sc, s, "trace")) return -1;
    GOI_HIP(hipGetLastError());
    return N;
}

int goi_raster_backward(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                        const void* image_buffer, const int* radii, const float* out_alpha, const float* dL_dout_color,
                        const float* dL_dout_semantic, const float* dL_dout_depth, const float* dL_dout_alpha,
                        float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic,
                        float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                        float* dL_drot, void* scratch, void* stream) {
    return goi_raster_backward2(scene, R, geom_buffer, binning_buffer, image_buffer, radii, out_alpha, dL_dout_color,
                                dL_dout_semantic, dL_dout_depth, dL_dout_alpha, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                dL_dsemantic, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, scratch, nullptr,
                                stream);
}

int goi_raster_backward3(const GoiRasterScene* scene, int R, int scratch_instances, int flags, const void* geom_buffer,
                         const void* binning_buffer, const void* image_buffer, const int* radii, const float* out_alpha,
                         const float* dL_dout_color, const float* dL_dout_semantic, const float* dL_dout_depth,
                         const float* dL_dout_alpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                         float* dL_dsemantic, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, void* scratch, const int* prev_radii, void* stream) {
    refresh_options();
    if (validate(scene, true, false)) return -1;  // opacity lives in the forward's records
    const GoiRasterScene& sc = *scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t P = (size_t)sc.P;
    if (P == 0) return 0;
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail("workspace pointer is NULL");
    GeomView g;
    ImageView im;
    BinView bv;
    geom_layout(sc.P, const_cast<char*>(static_cast<const char*>(geom_buffer)), &g);
    image_layout(sc.W, sc.H, const_cast<char*>(static_cast<const char*>(image_buffer)), &im);
    const int fin = tile_sort_result_index(sc.W, sc.H, R);
    if (R > 0) binning_layout(R, const_cast<char*>(static_cast<const char*>(binning_buffer)), &bv);
    const bool rows_path = scratch != nullptr && (g_options.bwd_variant & 15) != 1;
    if (scratch_instances < 0 || scratch_instances > R) return fail("scratch_instances must be in 0..R (0: the scratch is laid out for R)");
    // Rs: instances the ROW SCRATCH holds rows for (the binning workspace keeps R); every slot index is below 4 x num_rendered <= 4 Rs
    const int Rs = scratch_instances > 0 ? scratch_instances : R;
    const bool accumulate = (flags & GOI_BACKWARD_ACCUMULATE) != 0;
    if (accumulate && (!rows_path || g_options.bwd_records != 1 || prev_radii || (sc.shs && !dL_dsh) || sc.debug))
        return fail("GOI_BACKWARD_ACCUMULATE needs the default backward (scratch given, bwd_variant 0 / 2, bwd_records 1), dL_dsh when "
                    "the colours are SH, prev_radii NULL and debug off");
    if (rows_path) {
        // atomic-free path: (quadrant, Gaussian) partial rows + validity bytes, then a fixed-order sum
        BwdScratchView scr;
        bwd_scratch_layout(Rs, sc.S, static_cast<char*>(scratch), &scr);
        {
            StageTimer t(GOI_STAGE_BLEND_BWD, s);
            if (R > 0) {
                // the validity bytes of the slots this frame can use are cleared by extra workgroups of the quadrant-order
                // launch (count on the device: 4 x num_rendered bytes, not 4 x capacity) -- or by a memset where that
                // launch does not exist
                if (!launch_quad_order(sc, im, s, scr.flags, g.counters + COUNTER_N, (uint32_t)Rs, scr.big_ctl)) {
                    GOI_HIP(hipMemsetAsync(scr.flags, 0, round_up_256((size_t)Rs * 4), s));  // (the layout ends with 256 spare bytes)
                    GOI_HIP(hipMemsetAsync(scr.big_ctl, 0, 8 * sizeof(uint32_t), s));
                }
                launch_render_bwd_rows(sc, g, im, bv.vals[fin], radii, out_alpha, dL_dout_color, dL_dout_semantic,
                                       dL_dout_depth, dL_dout_alpha, scr, s, bv.qmask);
            }
        }
        if (check_stage(sc, s, "backward blend")) return -1;
        StageTimer t(GOI_STAGE_PREPROCESS_BWD, s);
        if (g_options.bwd_records == 2 && bwd_row_floats(sc.S) == 32 && R > 0 && !accumulate) {
            // the per-Gaussian backward sums its Gaussians' rows itself; only the BIG Gaussians pass through reduce_big_k's records
            launch_reduce_big_only(sc, g, Rs, scr, s);
            launch_preprocess_bwd(sc, g, radii, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh,
                                  dL_dscale, dL_drot, s, scr.rows, dL_dopacity, dL_dsemantic, prev_radii, scr.flags, Rs);
        } else if (g_options.bwd_records) {
            // the sums stay in the row scratch (one record per listed Gaussian); preprocess_bwd_k writes the per-id outputs
            launch_reduce_rows(sc, g, Rs, scr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s, true);
            launch_preprocess_bwd(sc, g, radii, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh,
                                  dL_dscale, dL_drot, s, scr.rows, dL_dopacity, dL_dsemantic, prev_radii, nullptr, 0, accumulate);
        } else {
            launch_reduce_rows(sc, g, Rs, scr, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepth, s);
            launch_preprocess_bwd(sc, g, radii, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh,
                                  dL_dscale, dL_drot, s);
        }
    } else {
        // atomic path: the accumulated gradients start from zero
        GOI_HIP(hipMemsetAsync(dL_dmean2D, 0, 3 * P * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(dL_dconic, 0, 4 * P * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(dL_dopacity, 0, P * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(dL_dcolor, 0, 3 * P * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(dL_dsemantic, 0, (size_t)sc.S * P * sizeof(float), s));
        GOI_HIP(hipMemsetAsync(dL_ddepth, 0, P * sizeof(float), s));
        if (R > 0) {
            StageTimer t(GOI_STAGE_BLEND_BWD, s);
            launch_render_bwd_tile(sc, g, im, bv.vals[fin], out_alpha, dL_dout_color, dL_dout_semantic, dL_dout_depth,
                                   dL_dout_alpha, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic,
                                   dL_ddepth, s);
        }
        if (check_stage(sc, s, "backward blend")) return -1;
        StageTimer t(GOI_STAGE_PREPROCESS_BWD, s);
        launch_preprocess_bwd(sc, g, radii, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh,
                              dL_dscale, dL_drot, s);
    }
    if (check_stage(sc, s, "backward preprocess")) return -1;
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_backward2(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                         const void* image_buffer, const int* radii, const float* out_alpha, const float* dL_dout_color,
                         const float* dL_dout_semantic, const float* dL_dout_depth, const float* dL_dout_alpha,
                         float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic,
                         float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                         float* dL_drot, void* scratch, const int* prev_radii, void* stream) {
    return goi_raster_backward3(scene, R, 0, 0, geom_buffer, binning_buffer, image_buffer, radii, out_alpha, dL_dout_color,
                                dL_dout_semantic, dL_dout_depth, dL_dout_alpha, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                dL_dsemantic, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, scratch, prev_radii,
                                stream);
}

int goi_raster_backward_semantics(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                                  const void* image_buffer, const int* radii, const float* out_alpha,
                                  const float* dL_dout_semantic, float* dL_dsemantic, void* scratch, void* stream) {
    refresh_options();
    if (validate(scene, true, false)) return -1;
    const GoiRasterScene& sc = *scene;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sc.P == 0) return 0;
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail("workspace pointer is NULL");
    if (!scratch) return fail("goi_raster_backward_semantics needs the scratch of goi_raster_backward_scratch_bytes");
    if (!dL_dout_semantic || !dL_dsemantic || !out_alpha || !radii) return fail("a required pointer is NULL");
    GeomView g;
    ImageView im;
    BinView bv;
    geom_layout(sc.P, const_cast<char*>(static_cast<const char*>(geom_buffer)), &g);
    image_layout(sc.W, sc.H, const_cast<char*>(static_cast<const char*>(image_buffer)), &im);
    const int fin = tile_sort_result_index(sc.W, sc.H, R);
    if (R > 0) binning_layout(R, const_cast<char*>(static_cast<const char*>(binning_buffer)), &bv);
    BwdScratchView scr;  // same allocation as the full backward; rows are narrower here
    bwd_scratch_layout(R, sc.S, static_cast<char*>(scratch), &scr);
    const int row_floats = ((4 * ((sc.S + 3) / 4) + 15) / 16) * 16;
    {
        StageTimer t(GOI_STAGE_BLEND_BWD, s);
        if (R > 0) {
            if (!launch_quad_order(sc, im, s, scr.flags, g.counters + COUNTER_N, (uint32_t)R, scr.big_ctl)) {
                GOI_HIP(hipMemsetAsync(scr.flags, 0, round_up_256((size_t)R * 4), s));  // (the layout ends with 256 spare bytes)
                GOI_HIP(hipMemsetAsync(scr.big_ctl, 0, 8 * sizeof(uint32_t), s));
            }
            launch_render_bwd_sem(sc, g, im, bv.vals[fin], radii, out_alpha, dL_dout_semantic, scr.rows, scr.flags,
                                  row_floats, s, bv.qmask);
        }
    }
    if (check_stage(sc, s, "backward blend (semantics)")) return -1;
    StageTimer t(GOI_STAGE_PREPROCESS_BWD, s);
    launch_reduce_sem_rows(sc, g, R, scr, row_floats, dL_dsemantic, s);
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* present, void* stream) {
    (void)projmatrix;
    if (P < 0) return fail("bad P");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail("NULL pointer");
    launch_mark_visible(P, means3D, viewmatrix, present, static_cast<hipStream_t>(stream));
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_semantic_decode(const float* sem, int S, long long HW, const float* W, const float* bias, int n_codes,
                        const float* code_score, float thresh, float* sim_out, int* idx_out, uint8_t* bg_mask_out,
                        void* stream) {
    refresh_options();
    if (!sem || !W || !bias) return fail("goi_semantic_decode: NULL input");
    if (S < 1 || S > 32 || n_codes < 1 || HW < 0) return fail("goi_semantic_decode: need 1 <= S <= 32, n_codes >= 1");
    if (HW == 0) return 0;
    if (launch_semantic_decode(sem, S, HW, W, bias, n_codes, code_score, thresh, sim_out, idx_out, bg_mask_out,
                               static_cast<hipStream_t>(stream)) < 0)
        return fail("goi_semantic_decode: code book too large for LDS");
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_codebook_loss_partial_rows(void) { return codebook_loss_waves(); }

int goi_codebook_loss_rows(const float* sim_raw, const float* inv_gnorm, const float* sem, const float* W,
                           const float* bias, long long HW, int C, int S, float t, float* dsim, float* dsem,
                           float* partials, void* stream) {
    if (HW < 0) return fail("goi_codebook_loss_rows: bad HW");
    if (!sim_raw || !inv_gnorm || !sem || !W || !dsim || !dsem || !partials)
        return fail("goi_codebook_loss_rows: a required pointer is NULL");
    if (launch_codebook_rows(sim_raw, inv_gnorm, sem, W, bias, HW, C, S, t, dsim, dsem, partials,
                             static_cast<hipStream_t>(stream)) < 0)
        return fail("goi_codebook_loss_rows: supported sizes are 1 <= S <= 16, 1 <= C <= 512");
    GOI_HIP(hipGetLastError());
    return 0;
}

size_t goi_codebook_sim_workspace_bytes(void) { return codebook_sim_workspace_bytes(); }

int goi_codebook_sim(const float* g, const float* lut1, long long HW, int C, int D, float* sim, float* inv_gnorm,
                     void* workspace, void* stream) {
    if (!g || !lut1 || !sim || !inv_gnorm || !workspace) return fail("goi_codebook_sim: a required pointer is NULL");
    if (launch_codebook_sim(g, lut1, HW, C, D, sim, inv_gnorm, workspace, static_cast<hipStream_t>(stream)) < 0)
        return fail("goi_codebook_sim: supported shape is D = 256, C <= 304, C % 4 = 0");
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_codebook_dlut_partial_blocks(void) { return codebook_dlut_blocks(); }

int goi_codebook_dlut(const float* dsim, const float* g, long long HW, int C, int D, float* partial, void* stream) {
    if (!dsim || !g || !partial || HW < 0) return fail("goi_codebook_dlut: bad arguments");
    if (launch_codebook_dlut(dsim, g, HW, C, D, partial, static_cast<hipStream_t>(stream)) < 0)
        return fail("goi_codebook_dlut: supported shape is D = 256, 288 < C <= 304, HW % 4 = 0");
    GOI_HIP(hipGetLastError());
    return 0;
}

size_t goi_codebook_fused_workspace_bytes(long long HW) { return HW < 0 ? 0 : codebook_fused_workspace_bytes(HW); }
int goi_codebook_fused_partial_rows(void) { return codebook_fused_rows(); }

int goi_codebook_fused(const float* g, const float* lut1, const float* sem, const float* W, const float* bias, long long HW,
                       int C, int D, int S, float t, float* dsem, float* partials, float* dlut_partial, void* workspace,
                       void* stream) {
    if (!g || !lut1 || !sem || !W || !dsem || !partials || !dlut_partial || !workspace)
        return fail("goi_codebook_fused: a required pointer is NULL");
    if (launch_codebook_fused(g, lut1, sem, W, bias, HW, C, D, S, t, dsem, partials, dlut_partial, workspace,
                              static_cast<hipStream_t>(stream)) < 0)
        return fail("goi_codebook_fused: supported shape is D = 256, 288 < C <= 304, 1 <= S <= 16, HW % 4 = 0, HW < 2^25");
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* gcol,
                           float* dL_dsh, void* stream) {
    if (P <= 0 || V <= 0) return 0;
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1) || M > 16) return fail("goi_raster_sh_grad_from_views: need 0 <= D <= 3 and (D+1)^2 <= M <= 16");
    if (!means3D || !campos || !gcol || !dL_dsh) return fail("goi_raster_sh_grad_from_views: NULL pointer");
    launch_sh_grad_from_views(P, D, M, V, means3D, campos, gcol, dL_dsh, static_cast<hipStream_t>(stream));
    if (hipGetLastError() != hipSuccess) return fail("goi_raster_sh_grad_from_views: launch failed");
    return 0;
}

int goi_adam_step(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                  const unsigned char* nograd_mask, void* stream) {
    return goi_adam_step_guarded(groups, n_groups, beta1, beta2, eps, nograd_mask, nullptr, stream);
}

const uint32_t* goi_raster_truncated_flag(const void* geom_buffer, int P) {
    if (!geom_buffer || P <= 0) return nullptr;
    GeomView g;
    geom_layout(P, const_cast<char*>(static_cast<const char*>(geom_buffer)), &g);
    return g.counters + COUNTER_OVF;
}

int goi_adam_step_guarded(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                          const unsigned char* nograd_mask, const uint32_t* skip_flag, void* stream) {
    if (n_groups < 0 || n_groups > GOI_ADAM_MAX_GROUPS) return fail("goi_adam_step: n_groups must be 0..8");
    if (n_groups && !groups) return fail("goi_adam_step: groups is NULL");
    for (int i = 0; i < n_groups; i++) {
        const GoiAdamGroup& g = groups[i];
        if (g.numel < 0 || g.row_len < 1) return fail("goi_adam_step: bad numel / row_len");
        if (g.numel == 0) continue;
        if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq) return fail("goi_adam_step: a tensor pointer is NULL");
        if ((reinterpret_cast<uintptr_t>(g.param) | reinterpret_cast<uintptr_t>(g.grad) |
             reinterpret_cast<uintptr_t>(g.exp_avg) | reinterpret_cast<uintptr_t>(g.exp_avg_sq)) & 15)
            return fail("goi_adam_step: tensors must be 16-byte aligned");
    }
    launch_adam_step(groups, n_groups, beta1, beta2, eps, nograd_mask, skip_flag, static_cast<hipStream_t>(stream));
    GOI_HIP(hipGetLastError());
    return 0;
}

size_t goi_knn_workspace_bytes(int P) { return P > 0 ? knn_workspace_bytes(P) : 0; }

int goi_knn_dist2(int P, const float* points, float* mean_dist2, void* workspace, void* stream) {
    if (P < 0) return fail("goi_knn_dist2: bad P");
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !workspace) return fail("goi_knn_dist2: a required pointer is NULL");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail("goi_knn_dist2: workspace must be 256-byte aligned");
    launch_knn(P, points, mean_dist2, workspace, static_cast<hipStream_t>(stream));
    GOI_HIP(hipGetLastError());
    return 0;
}

void goi_raster_profile_enable(int on) { g_profile_mask.store(on ? ~0u : 0u); }

void goi_raster_profile_stages(unsigned stage_mask) { g_profile_mask.store(stage_mask); }

int goi_raster_set_option(const char* name, int value) {
    if (!name) return fail("option name is NULL");
    std::lock_guard<std::mutex> lk(g_options_mu);
    Options& g_options = g_shared_options;  // (calls already running keep their own snapshot)
    if (!strcmp(name, "fwd_variant")) g_options.fwd_variant = value;
    else if (!strcmp(name, "bwd_variant")) {
        if (value < 0 || value > 2) return fail("bwd_variant must be 0, 1 or 2");
        g_options.bwd_variant = value;
    }
    else if (!strcmp(name, "sort_variant")) g_options.sort_variant = value;
    else if (!strcmp(name, "sort_small")) g_options.sort_small = value;
    else if (!strcmp(name, "pre_shdma")) g_options.pre_shdma = value;
    else if (!strcmp(name, "sort_lookback")) g_options.sort_lookback = value;
    else if (!strcmp(name, "sort_tickets")) {
        // 0 is an EXPERIMENT that is only safe while the hardware starts workgroups in index order (HIP promises no such thing: a
        // tile could wait for a workgroup that never starts): refused unless the process opts in explicitly
        if (value == 0 && !getenv("GOI_UNSAFE_EXPERIMENTS")) return fail("sort_tickets 0 is an unsafe experiment: set GOI_UNSAFE_EXPERIMENTS=1 to allow it");
        g_options.sort_tickets = value;
    }
    else if (!strcmp(name, "cull_variant")) g_options.cull_variant = value;
    else if (!strcmp(name, "bwd_order")) {
        if (value < 0 || value > 8) return fail("bwd_order must be 0 .. 8");
        g_options.bwd_order = value;
    }
    else if (!strcmp(name, "decode_variant")) g_options.decode_variant = value;
    else if (!strcmp(name, "bwd_masks")) {
        if (value < 0 || value > 1) return fail("bwd_masks must be 0 or 1");
        g_options.bwd_masks = value;
    }
    else if (!strcmp(name, "bwd_records")) {
        if (value < 0 || value > 2) return fail("bwd_records must be 0, 1 or 2");
        g_options.bwd_records = value;
    }
    else return fail(std::string("unknown option ") + name);
    return 0;
}

int goi_raster_profile_collect(double* ms, int* calls) {
    std::vector<StageEvents> events;
    {
        std::lock_guard<std::mutex> lk(g_profile_mu);
        events.swap(g_events);
    }
    for (auto& ev : events) {
        GOI_HIP(hipEventSynchronize(ev.b));
        float t = 0.f;
        GOI_HIP(hipEventElapsedTime(&t, ev.a, ev.b));
        if (ms) ms[ev.stage] += (double)t;
        if (calls) calls[ev.stage] += 1;
        std::lock_guard<std::mutex> lk(g_profile_mu);
        g_pool.push_back(ev.a);
        g_pool.push_back(ev.b);
    }
    return 0;
}

int goi_raster_blend_stats(int P, int W, int H, int R, const void* geom_buffer, const void* binning_buffer,
                           const void* image_buffer, unsigned long long* counters, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P <= 0 || W <= 0 || H <= 0 || R < 0) return fail("goi_raster_blend_stats: bad P/W/H/R");
    if (!geom_buffer || !image_buffer || !counters || (R > 0 && !binning_buffer)) return fail("workspace pointer is NULL");
    GOI_HIP(hipMemsetAsync(counters, 0, GOI_BLEND_STATS_WORDS * sizeof(unsigned long long), s));
    if (R == 0) return 0;
    GeomView g;
    ImageView im;
    BinView bv;
    geom_layout(P, const_cast<char*>(static_cast<const char*>(geom_buffer)), &g);
    image_layout(W, H, const_cast<char*>(static_cast<const char*>(image_buffer)), &im);
    binning_layout(R, const_cast<char*>(static_cast<const char*>(binning_buffer)), &bv);
    launch_blend_stats(W, H, g, im, bv.vals[tile_sort_result_index(W, H, R)], bv.qmask, counters, s);
    GOI_HIP(hipGetLastError());
    return 0;
}

int goi_raster_debug_views(int P, int W, int H, int R, const void* geom_buffer, const void* binning_buffer,
                           const void* image_buffer, float* depths, float* means2D, float* conic_opacity, float* rgb,
                           uint32_t* tiles_touched, uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib,
                           void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P <= 0) return 0;
    GeomView g;
    ImageView im;
    geom_layout(P, const_cast<char*>(static_cast<const char*>(geom_buffer)), &g);
    image_layout(W, H, const_cast<char*>(static_cast<const char*>(image_buffer)), &im);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    // strided copies out of the 48-byte records
    const char* rec = reinterpret_cast<const char*>(g.rec);
    const size_t pitch = sizeof(GaussRec);
    if (means2D) GOI_HIP(hipMemcpy2DAsync(means2D, 8, rec + 0, pitch, 8, P, hipMemcpyDeviceToDevice, s));
    if (conic_opacity) {  // (a,b) then (c,o)
        GOI_HIP(hipMemcpy2DAsync(conic_opacity, 16, rec + 8, pitch, 8, P, hipMemcpyDeviceToDevice, s));
        GOI_HIP(hipMemcpy2DAsync(reinterpret_cast<char*>(conic_opacity) + 8, 16, rec + 16, pitch, 8, P,
                                 hipMemcpyDeviceToDevice, s));
    }
    if (depths) GOI_HIP(hipMemcpy2DAsync(depths, 4, rec + 44, pitch, 4, P, hipMemcpyDeviceToDevice, s));
    if (rgb) GOI_HIP(hipMemcpy2DAsync(rgb, 12, rec + 32, pitch, 12, P, hipMemcpyDeviceToDevice, s));
    if (tiles_touched)
        GOI_HIP(hipMemcpyAsync(tiles_touched, g.tiles_touched, sizeof(uint32_t) * P, hipMemcpyDeviceToDevice, s));
    if (ranges) GOI_HIP(hipMemcpyAsync(ranges, im.ranges, sizeof(uint2) * gx * gy, hipMemcpyDeviceToDevice, s));
    if (n_contrib)
        GOI_HIP(hipMemcpyAsync(n_contrib, im.n_contrib, sizeof(uint32_t) * (size_t)W * H, hipMemcpyDeviceToDevice, s));
    if (point_list && R > 0) {
        BinView bv;
        binning_layout(R, const_cast<char*>(static_cast<const char*>(binning_buffer)), &bv);
        const int fin = tile_sort_result_index(W, H, R);
        GOI_HIP(hipMemcpyAsync(point_list, bv.vals[fin], sizeof(uint32_t) * (size_t)R, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

}  // extern "C"
