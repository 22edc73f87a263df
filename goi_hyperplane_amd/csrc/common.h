// Internal declarations shared by the HIP translation units of libgoi_raster.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "goi_raster.h"

namespace goi {

constexpr int TILE = 16;          // tile edge; part of the numerical contract (CR/config.h:16-17)
constexpr int TILE_PIX = TILE * TILE;
constexpr int WAVE = 64;

// Per-Gaussian render record written by the forward preprocess: everything the blend kernels
// need about a Gaussian except its semantic row, in three 16-byte words (48 B, one gather).
//   q0 = (x, y, conic.a, conic.b)   q1 = (conic.c, opacity, hx, hy)   q2 = (r, g, b, depth)
// q0 and q1 are what the hit test of a candidate reads; q2 is what only a hit needs (and IS its first staged feature quad).
// hx / hy: half extents (pixels) of the axis-aligned box outside which alpha < 1/255 can be
// proven; < 0 when the Gaussian can never reach 1/255, +inf when no bound is known.
struct GaussRec {
    float4 q0, q1, q2;
};

// ---- workspace layouts (opaque to callers; forward -> backward of one call must agree) -------
struct GeomView {
    GaussRec* rec;            // [P]
    float* cov3D;             // [6P] (only when computed from scale/rotation)
    uint32_t* tiles_touched;  // [P]
    uint8_t* clamped;         // [P] bit0..2 = r,g,b clamped at 0
    uint32_t* sort_keys[2];   // [P] depth bits (ping-pong); preprocess leaves the RAW keys (by Gaussian id) in [1]
    uint32_t* sort_vals[2];   // [P] Gaussian ids (ping-pong); [final][0 .. V) = depth order of the V listed Gaussians
    uint32_t* offsets;        // [V] exclusive prefix of tiles_touched in DEPTH order: where emit puts a Gaussian's instances
    // [P] what emit and the backward need of a Gaussian besides its record, in ONE 16-byte gather (they were three):
    //   .x  the same prefix by Gaussian id (listed Gaussians only; written by emit) = its first row slot in the backward
    //   .y  its radius in pixels (the int radii[] of the API)
    //   .zw which tiles of the listed rectangle its ellipse reaches (cull_variant 2), row-major bits; TMASK_FULL: all of
    //       them (rectangles of more than 64 tiles, cull_variant < 2)
    uint4* aux;
    uint2* blk_agg;           // [ceil(P/256)] per preprocess block: (listed Gaussians, tiles touched)
    // [ceil(blocks / COARSE_BLOCKS)][COARSE_STRIDE] the same pair summed over COARSE_BLOCKS consecutive preprocess blocks, packed
    // (listed << 40 | tiles touched) in the first word of a 64-byte entry: one integer atomic per preprocess block; lets
    // compact_listed_k find its base rank from ~nblk / 32 + 31 words instead of all nblk pairs (3 M Gaussians: 49 -> 26 us)
    unsigned long long* blk_coarse;
    uint32_t* bigq;  // [P] emit's queue of big rectangles (ranks in the depth order), binning.hip; ALIASES sort_keys[1], which is
                     // free between the depth sort and the next frame's preprocess
    uint32_t* scratch;        // scan partials + radix histograms
    uint32_t* counters;       // [COUNTER_WORDS]: 1 = error flag, 2 = cull_variant of this forward, NR_BASE.. = num_rendered stripes
    size_t scratch_words;
};
struct ImageView {
    uint32_t* n_contrib;  // [HW]
    uint2* ranges;        // [T]
    uint32_t* qcost;      // [4T] per 8x8 quadrant: list positions its wave has to walk in the backward (max n_contrib)
    uint32_t* qorder;     // [quad_grid(4T)] launch slot -> quadrant, heaviest first inside each XCD's band (backward)
    // [4T] MEMBER mask of every quadrant's FIRST round of 64 list positions (bit j: the Gaussian at list position j
    // contributed to some pixel of the quadrant); later rounds live in BinView::qmask (their number depends on N).
    // Written by the forward blend, read by the backward blend: see member_mask_ptr.
    unsigned long long* qmask0;
};
// Backward scratch (goi_raster_backward_scratch_bytes): one 128-byte partial-gradient row per
// (emit-order instance, quadrant) and one validity byte per row.
// row = [sem 0..4*ceil(S/4)) | r g b depth | mean2D.x .y conic.a .b .c opacity | pad], 64-byte multiple
inline int bwd_row_floats(int S) { return ((4 * ((S + 3) / 4) + 4 + 6 + 15) / 16) * 16; }
#ifndef GOI_REDUCE_BIG_INST
#define GOI_REDUCE_BIG_INST 384
#endif
#ifndef GOI_REDUCE_HUGE_INST
#define GOI_REDUCE_HUGE_INST 2048
#endif
#ifndef GOI_REDUCE_DENSE_RATIO
#define GOI_REDUCE_DENSE_RATIO 10
#endif
// Instances above which a Gaussian's rows are summed by a workgroup (part) of its own (reduce_rows.hip, "BIG Gaussians"):
// REDUCE_BIG_INST on a frame with more than REDUCE_DENSE_RATIO instances per listed Gaussian, 1024 otherwise (the frame decides
// on the device, from its own counters); 64 quarter waves beyond REDUCE_HUGE_INST instances, 16 up to there.
constexpr uint32_t REDUCE_BIG_INST = GOI_REDUCE_BIG_INST, REDUCE_HUGE_INST = GOI_REDUCE_HUGE_INST,
                   REDUCE_DENSE_RATIO = GOI_REDUCE_DENSE_RATIO;
struct BwdScratchView {
    float* rows;     // [4N][bwd_row_floats(S)]: slot = (emit-order instance) * 4 + quadrant
    uint8_t* flags;  // [4N] validity bytes
    // Gaussians with more than 1024 instances get a workgroup of their own (reduce_rows.hip, "BIG Gaussians"):
    uint32_t* big_ctl;  // [8]: word 1 = HUGE Gaussians registered, word 2 = the other big ones (zeroed with the validity bytes)
    uint4* big_desc;    // [cap_big] (first instance, instances, -, Gaussian id): huge ones from the front, the others from the back
    size_t cap_big;
};
size_t bwd_scratch_layout(int N, int S, char* base, BwdScratchView* v);

struct BinView {
    uint32_t* keys[2];  // [N] tile ids (ping-pong)
    uint32_t* vals[2];  // [N] Gaussian ids (ping-pong)
    uint32_t* scratch;
    size_t scratch_words;
    unsigned long long* qmask;  // [4 (N / 64 + 2)] member masks of the rounds >= 1 (member_mask_ptr)
};

// Where the forward blend leaves, and the backward blend finds, the 64-bit MEMBER mask of round r (list positions
// 64 r .. 64 r + 63) of quadrant q of a tile whose list starts at x0: round 0 in the image state (one word per quadrant),
// round r >= 1 at word (x0 / 64 + r) of the binning state -- tile t's rounds 1 .. ceil(len / 64) - 1 end at word
// floor((x0 + len - 1) / 64) < floor(x1 / 64) + 1, the first word of the next tile: no two tiles share a word, and the
// array needs N / 64 + 1 words per quadrant whatever the tile grid is.
__host__ __device__ inline unsigned long long* member_mask_ptr(unsigned long long* qmask0, unsigned long long* qmask, int tile,
                                                                int q, uint32_t x0, int r) {
    return r == 0 ? qmask0 + ((size_t)tile * 4 + q) : qmask + (((size_t)(x0 >> 6) + (size_t)r) * 4 + q);
}

size_t geom_layout(int P, char* base, GeomView* v);
size_t image_layout(int W, int H, char* base, ImageView* v);
size_t binning_layout(int N, char* base, BinView* v);

// Tuning / experiment switches (goi_raster_set_option); defaults are the shipped configuration.
struct Options {
    int fwd_variant = 1;  // 0: one candidate per loop trip, 1: two candidates per trip (default); experiments (render_fwd.hip): 2 the
                          // 16 pixels x 4 Gaussians mapping, 3 scalar-operand features (S = 16), 4 fp32 outer-product MFMA
    int bwd_variant = 0;  // low 4 bits: 0 atomic-free wave-per-quadrant backward (needs scratch) with the split-f16
                          // MFMA flush (fp32-grade), 2 the same with the exact-fp32 flush, 1 workgroup-per-tile + atomics
    int sort_variant = 1;  // 0: histogram / scan / scatter per pass, 1: onesweep (decoupled look-back, default)
    int sort_lookback = 1;  // onesweep: 1 grouped look-back (two round trips: group aggregates, group totals), 0 the chained decoupled look-back
    int sort_tickets = 1;  // onesweep: 1 tiles are dealt by a ticket counter (a tile only waits for tiles that are running), 0 tile = workgroup index (experiment)
    int sort_small = 0;    // 1: sorts of up to 2 M keys also take the adaptive 512 x (2..16) tile (scan_sort.hip) instead of 1024 x 4
    int decode_variant = 1;  // semantic decode, S <= 16: 1 split-bf16 MFMA contraction, 2 pixel blocks per operand fetch (2: 4 blocks, 3: 1 block; bit-identical), 0 fp32 MFMA
    int cull_variant = 2;  // 0: a Gaussian is listed in every tile of its 3-sigma rectangle (the reference's lists),
                           // 1: only in the tiles its exact contribution box touches, 2: only in the tiles its contribution
                           // ELLIPSE reaches (rectangles of up to 64 tiles).  Same images and gradients, bit for bit.
    int pre_shdma = 0;     // preprocess_fwd_k, SH colours with M = 16: 1 the wave moves the SH rows of its visible lanes to LDS by
                           // DMA (EXPERIMENT, measured: 67 vs 63.5 us at 1 M, 190 vs 193 us at 3 M), 0 every lane fetches its own
                           // row (default).  Bit-identical.
    int bwd_order = 1;     // backward blend: v >= 1 the quadrants of each XCD's band are launched longest-first (their cost is
                           // known from the forward's n_contrib; cost classes of 2^(3+v) list positions), 0 in tile order.
                           // Same rows, same gradients.
    int bwd_masks = 1;     // atomic-free backward blend: 1 walks the MEMBER masks the forward blend left (no candidate tests, no
                           // evaluation of pairs that contribute nowhere; default), 0 tests every candidate of its list against
                           // the quadrant itself.  Same rows, same gradients, bit for bit.
    int bwd_records = 1;   // atomic-free backward: 2 EXPERIMENT the per-Gaussian backward sums its Gaussians' rows itself (no record, no
                           // reduce_rows_k; 128-byte rows), 1 the per-Gaussian sums stay in the row scratch as records and
                           // preprocess_bwd_k writes every per-id output (default), 0 reduce_rows_k writes six per-id arrays
                           // (and zeros for the unlisted Gaussians) that preprocess_bwd_k reads back.  Same gradients, bit for bit.
};
// The switches an entry point works with are a per-THREAD snapshot taken when the call starts (refresh_options):
// goi_raster_set_option changes the process-wide set under a mutex, and a call that is already running on another host
// thread keeps the values it started with -- every stage of one forward or backward sees one consistent set.
extern thread_local Options g_options;
void refresh_options();

// ---- device-wide primitives (scan_sort.hip) ----------------------------------------------------
size_t scan_scratch_words(size_t n);
size_t sort_scratch_words(size_t n);
// out[i] = sum_{j<i} f(j), f(j) = gather ? in[gather[j]] : in[j]; *total (device, may be NULL) = sum.
// n_dev (may be NULL): the count lives on the device, `n` is a capacity (grids cover n; min(*n_dev, n) elements are scanned).
void exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, size_t n, uint32_t* total,
                        uint32_t* scratch, hipStream_t s, const uint32_t* n_dev = nullptr);
// Stable LSD radix sort of (key,val) pairs on key bits [lo, hi).  Data start in keys[0]/vals[0];
// returns the index (0/1) of the buffers holding the sorted result.
// Onesweep control words (status, global digit histograms, tickets, error) sit at the start of `scratch`:
//   cleared     : the caller has zeroed the first radix_sort_control_words(n, lo, hi) words (e.g. together with
//                 neighbouring counters in one memset) -- otherwise the sort clears them itself;
//   ghist_ready : after that clear the caller has written the per-pass global digit histograms
//                 ([pass][256] words at radix_sort_ghist(...)), so the sort's histogram kernel is skipped.
//   n_dev       : (onesweep only) the element count lives on the DEVICE and `n` is a capacity: grids and control words
//                 are sized for n, the kernels sort min(*n_dev, n) elements (speculative forward, api.hip).
//   frame_error : (onesweep only) a device word that gets bit 1 OR-ed in when a look-back spin runs out of its budget (a
//                 preempted or wedged GPU): the sort carries on -- it must never hang the device -- but its output is
//                 garbage and the caller must not use the frame (COUNTER_SORTERR / COUNTER_OVF).
int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], size_t n, int lo, int hi, uint32_t* scratch,
                     hipStream_t s, bool cleared = false, bool ghist_ready = false, const uint32_t* n_dev = nullptr,
                     uint32_t* frame_error = nullptr);
size_t radix_sort_control_words(size_t n, int lo, int hi);
uint32_t* radix_sort_ghist(uint32_t* scratch, size_t n, int lo, int hi);

// ---- stages ---------------------------------------------------------------------------------------
void launch_preprocess_fwd(const GoiRasterScene& sc, const GeomView& g, int* radii, uint2* ranges, int n_tiles,
                           hipStream_t s, const float* zcut = nullptr, uint32_t* zlearn = nullptr);
// Listed Gaussians (tiles_touched > 0) compacted in id order into sort_keys[0] / sort_vals[0] (the depth sort's input),
// counters[COUNTER_V / COUNTER_N], and -- ghist != NULL -- the four digit histograms of the compacted keys (the onesweep
// sort's prologue).  pad: entries [V, P) get key 0xFFFFFFFF (a sort
// that cannot take its count from the device sorts all P).
void launch_compact_listed(int P, const GeomView& g, uint32_t* ghist, bool pad, hipStream_t s);
constexpr int COARSE_BLOCKS = 32, COARSE_STRIDE = 8;  // GeomView::blk_coarse
// the onesweep sort's global digit histograms: [SORT_GH_COPIES][SORT_MAX_PASSES][256] words; writers that flush with atomics
// spread over the copies (copy = workgroup % SORT_GH_COPIES), a pass adds them up (scan_sort.hip)
constexpr int SORT_MAX_PASSES = 4, SORT_GH_COPIES = 8;
constexpr int PRE_BLOCK = 256;  // Gaussians per workgroup of preprocess_fwd_k = granularity of blk_agg / blk_pre
// cap: instances keys[] / vals[] can hold (instances past it are dropped: only an overflowed speculative frame has any)
void launch_emit(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                 uint32_t* vals, uint32_t cap, hipStream_t s);
bool emit_can_count_tiles(int W, int H);
void launch_emit_counting(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                          uint32_t* vals, uint2* ranges, uint32_t* clear, size_t clear_words, uint32_t cap, hipStream_t s);
void launch_tile_ranges_hist(int W, int H, uint2* ranges, uint32_t* ghist, uint32_t* counters, hipStream_t s);
void launch_ranges(int N, const uint32_t* n_dev, const uint32_t* sorted_keys, uint2* ranges, int T, hipStream_t s);
// zcut / zlearn: the speculative depth cut-off (goi_raster_forward_async_cut): per tile, the cut this frame's lists were built
// with (or NULL) and what the frame learns for the camera's next visit (float bits, max over the tile's quadrants; or NULL)
void launch_render_fwd(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s,
                       unsigned long long* qmask = nullptr, const float* zcut = nullptr, uint32_t* zlearn = nullptr,
                       uint32_t* host_words = nullptr,  // host_words: pinned, device-mapped words that get counters[0 .. 32) ...
                       uint32_t stamp = 0);  // ... and, behind a system-scope fence, `stamp` in word HOST_STAMP_WORD (api.hip: tickets)
constexpr int HOST_STAMP_WORD = 33;
// fwd_variant 2 (experiment): the 16 pixels x 4 Gaussians mapping of the forward blend (render_fwd_g4.hip); S <= 16, no depth cut
void launch_render_fwd_g4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                          float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s,
                          unsigned long long* qmask, uint32_t* host_words, uint32_t stamp);
// lane utilisation of the blend kernels counted from a forward's member masks / n_contrib (blend_stats.hip): out[GOI_BLEND_STATS_WORDS]
void launch_blend_stats(int W, int H, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                        const unsigned long long* qmask, unsigned long long* out, hipStream_t s);
void launch_trace_fwd(const GoiRasterScene& sc, const float* img_sem, const GeomView& g, const ImageView& im,
                      const uint32_t* point_list, float* out_color, float* gau_sem, int* num_gsem, hipStream_t s);
// launch order of the backward's quadrant waves (render_bwd.hip): im.qcost -> im.qorder
bool quad_order_enabled(int W, int H);
// clear_flags != NULL: extra workgroups of the same launch zero the first 4 * min(*n_dev, cap) validity bytes of the backward
// scratch (what a memset of 4 * cap bytes did) and the 8 control words at clear_ctl (BwdScratchView::big_ctl).  Returns false -- nothing launched, nothing cleared -- where the quadrant
// order is not used (quad_order_enabled).
bool launch_quad_order(const GoiRasterScene& sc, const ImageView& im, hipStream_t s, uint8_t* clear_flags = nullptr,
                       const uint32_t* n_dev = nullptr, uint32_t cap = 0, uint32_t* clear_ctl = nullptr);
// atomic-free backward blend: partial rows + flags into the scratch (render_bwd.hip)
void launch_render_bwd_rows(const GoiRasterScene& sc, const GeomView& g, const ImageView& im,
                            const uint32_t* point_list, const int* radii, const float* out_alpha, const float* dL_dpix,
                            const float* dL_dsem, const float* dL_ddepth, const float* dL_dalpha,
                            const BwdScratchView& scr, hipStream_t s, const unsigned long long* qmask = nullptr);
// feature-gradient-only backward blend (render_bwd_sem.hip) and its row reduction: dL/dsemantics only
void launch_render_bwd_sem(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                           const int* radii, const float* out_alpha, const float* dL_dsem, float* rows, uint8_t* flags,
                           int row_floats, hipStream_t s, const unsigned long long* qmask = nullptr);
void launch_reduce_big_only(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, hipStream_t s);
void launch_reduce_sem_rows(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, int row_floats,
                            float* dL_dsemantic, hipStream_t s);
// sums every Gaussian's partial rows (fixed order) into the six blend-gradient arrays (writes all P rows) -- or, `records`,
// into one record per listed Gaussian, left in the row scratch over the Gaussian's first slot (the arrays may be NULL then)
void launch_reduce_rows(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, float* dL_dmean2D,
                        float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic, float* dL_ddepth,
                        hipStream_t s, bool records = false);
void launch_render_bwd_tile(const GoiRasterScene& sc, const GeomView& g, const ImageView& im,
                            const uint32_t* point_list, const float* out_alpha, const float* dL_dpix,
                            const float* dL_dsem, const float* dL_ddepth, const float* dL_dalpha, float* dL_dmean2D,
                            float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic,
                            float* dL_ddepths, hipStream_t s);
void launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* gcol,
                               float* dL_dsh, hipStream_t s);
// record_rows != NULL: the blend gradients are the per-Gaussian records reduce_rows left in the row scratch, and the kernel
// writes dL_dmean2D, dL_dcolor, dL_dopacity and dL_dsemantic itself (dL_dconic / dL_ddepth are then neither read nor written)
void launch_preprocess_bwd(const GoiRasterScene& sc, const GeomView& g, const int* radii, float* dL_dmean2D,
                           const float* dL_dconic, float* dL_dcolor, const float* dL_ddepth, float* dL_dmean3D,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, hipStream_t s,
                           const float* record_rows = nullptr, float* dL_dopacity = nullptr, float* dL_dsemantic = nullptr,
                           const int* prev_radii = nullptr,  // prev_radii: BwdArgs (rows that already hold zeros)
                           const uint8_t* row_flags = nullptr, int N_cap = 0,  // row_flags: the kernel sums the rows itself (bwd_records 2)
                           bool accumulate = false);  // add to the outputs instead of writing them (record path only: BwdArgs::accumulate)
int launch_semantic_decode(const float* sem, int S, long long HW, const float* W, const float* bias, int n_codes,
                           const float* code_score, float thresh, float* sim_out, int* idx_out, uint8_t* bg_mask_out,
                           hipStream_t s);
int codebook_loss_waves();
int launch_codebook_rows(const float* sim, const float* inv_gnorm, const float* sem, const float* W, const float* bias,
                         long long HW, int C, int S, float t, float* dsim, float* dsem, float* partials, hipStream_t s);
size_t codebook_sim_workspace_bytes();
int launch_codebook_sim(const float* g, const float* l1, long long HW, int C, int D, float* sim, float* inv_gnorm,
                        void* workspace, hipStream_t s);
int codebook_dlut_blocks();
size_t codebook_fused_workspace_bytes(long long HW);
int codebook_fused_rows();
int launch_codebook_fused(const float* g, const float* l1, const float* sem, const float* W, const float* bias, long long HW,
                          int C, int D, int S, float t, float* dsem, float* partials, float* dl1_partial, void* workspace,
                          hipStream_t s);
int launch_codebook_dlut(const float* dsim, const float* g, long long HW, int C, int D, float* partial, hipStream_t s);
int launch_adam_step(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                     const uint8_t* nograd_mask, const uint32_t* skip_flag, hipStream_t s);
size_t knn_workspace_bytes(int P);
int launch_knn(int P, const float* points, float* mean_dist2, void* workspace, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

// Tile rectangle of a Gaussian (restates getRect, CR/auxiliary.h:46-56: float divide, truncation).
__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1) {
    x0 = min(gx, max(0, (int)((px - r) / TILE)));
    y0 = min(gy, max(0, (int)((py - r) / TILE)));
    x1 = min(gx, max(0, (int)((px + r + TILE - 1) / TILE)));
    y1 = min(gy, max(0, (int)((py + r + TILE - 1) / TILE)));
}

// Tile rectangle actually LISTED for a Gaussian.  The reference lists it in every tile of the 3-sigma
// rectangle above; a tile the exact contribution box (hx, hy: alpha < 1/255 is certain outside, GaussRec)
// does not touch can never receive a contribution from it, so with `cull` those tiles are dropped --
// the per-pixel sequence of contributing Gaussians, hence every output and gradient, is unchanged, while
// emit, the tile sort and the list walks of the blend kernels shrink with the instance count (x0.62 on the
// headline scene).  hx < 0: the Gaussian never reaches 1/255 (no tile); +inf: no bound known.
__device__ __forceinline__ void listed_rect(float px, float py, int r, float hx, float hy, bool cull, int gx, int gy,
                                            int& x0, int& y0, int& x1, int& y1) {
    tile_rect(px, py, r, gx, gy, x0, y0, x1, y1);
    if (!cull) return;
    if (hx < 0.f || hy < 0.f) {
        x1 = x0;
        y1 = y0;
        return;
    }
    if (hx < 1e30f) {  // tile t holds pixels 16t .. 16t+15; it is touched iff 16t <= px+hx and 16t+15 >= px-hx
        x0 = max(x0, (int)ceilf((px - hx - (float)(TILE - 1)) / TILE));
        x1 = min(x1, (int)floorf((px + hx) / TILE) + 1);
    }
    if (hy < 1e30f) {
        y0 = max(y0, (int)ceilf((py - hy - (float)(TILE - 1)) / TILE));
        y1 = min(y1, (int)floorf((py + hy) / TILE) + 1);
    }
    x1 = max(x1, x0);
    y1 = max(y1, y0);
}
// Tiles of a Gaussian's listed rectangle that its contribution ellipse reaches (cull_variant 2): a 64-bit mask, bit
// (row * width + column) of the rectangle.  The tile instance of tile (tx, ty) is then the mask's set bits below that bit.
__host__ __device__ inline unsigned long long aux_mask(const uint4& a) { return (unsigned long long)a.z | ((unsigned long long)a.w << 32); }
constexpr unsigned long long TMASK_FULL = ~0ull;
__device__ __forceinline__ uint32_t tile_instance(unsigned long long mask, int tx, int ty, int x0, int y0, int x1) {
    const uint32_t bit = (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
    if (mask == TMASK_FULL) return bit;
    return (uint32_t)__popcll(mask & ((1ull << bit) - 1ull));
}
// position (0-based, from bit 0) of the k-th set bit of mask (k < popcount(mask))
__device__ __forceinline__ int select_bit(unsigned long long mask, int k) {
    uint32_t m = (uint32_t)mask;
    int base = 0;
    int c = __popc(m);
    if (k >= c) {
        k -= c;
        base = 32;
        m = (uint32_t)(mask >> 32);
    }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        c = __popc(m & ((1u << sh) - 1u));
        if (k >= c) {
            k -= c;
            m >>= sh;
            base += sh;
        }
    }
    return base;
}
// num_rendered is accumulated by preprocess in NR_STRIPES partial counters, one per 128-byte line (a single hot
// counter serialises 4 K block atomics: +130 us); word NR_BASE + NR_STRIDE * i, the host adds them up
constexpr int NR_STRIPES = 32, NR_STRIDE = 32, NR_BASE = 32;
constexpr int COUNTER_WORDS = NR_BASE + NR_STRIPES * NR_STRIDE;
constexpr int COUNTER_CULL = 2;  // GeomView::counters[COUNTER_CULL]: the forward's cull_variant, read by emit / backward
constexpr int COUNTER_N = 3;     // num_rendered as ONE device word (the scan's total): what the tile sort, the ranges pass and
                                 // the backward's row reduction read when the host sized the frame from a capacity
constexpr int COUNTER_V = 5;     // number of LISTED Gaussians (tiles_touched > 0): the depth sort, its prefix sum and emit work on these only
constexpr int COUNTER_OVF = 4;   // 1: this frame's instance list was TRUNCATED (num_rendered > the binning capacity of a
                                 // speculative forward).  Written by emit; every backward kernel reads it and, if set, produces
                                 // ZERO gradients: a truncated frame must never reach the optimiser (the reference sizes its
                                 // buffers from the true count, CR/rasterizer_impl.cu:283-289, and cannot truncate)

constexpr uint32_t OVF_TRUNCATED = 1u, OVF_MISSORTED = 2u, OVF_CUT_TOO_TIGHT = 4u;  // bits of counters[COUNTER_OVF]
constexpr int COUNTER_BIGQ = 7;     // emit: number of BIG rectangles queued for emit_big_k (GeomView::bigq); tile_ranges_hist_k, which
                                    // runs behind both, puts it back to 0 (a redone frame emits again from the same workspace)
constexpr int COUNTER_SORTERR = 6;  // != 0: a look-back of the DEPTH sort timed out (scan_sort.hip): emit folds it into
                                    // COUNTER_OVF (the tile sort, which runs behind emit, sets bit 1 of COUNTER_OVF itself), so
                                    // a mis-sorted frame back-propagates zeros like a truncated one, and the read-back fails

// The depth sort runs ceil(32/8) = 4 ping-pong passes from buffer 0, so its result is in buffer 0.
inline int depth_sort_result_index() { return ((32 + 7) / 8) & 1; }

// Number of key bits that cover every tile id < n_tiles (the reference sorts on
// getHigherMsb(T) = floor(log2 T) + 1 tile bits, CR/rasterizer_impl.cu:35-50,304; any bit count
// that covers the ids gives the same order).
inline int tile_key_bits(uint32_t n_tiles) { return n_tiles <= 1 ? 1 : 32 - __builtin_clz(n_tiles); }

}  // namespace goi
