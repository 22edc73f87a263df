/*
 * include/goi_raster.h -- C ABI of libgoi_raster.so, the MI355X (gfx950) differentiable Gaussian
 * rasterizer with a per-Gaussian semantic-feature channel.
 *
 * This is the drop-in boundary for the reference's rasterizer hot path.  Each entry point replaces
 * one C++ entry of the reference (paths relative to submodules/diff-gaussian-rasterization/):
 *
 *   goi_raster_forward      <- CudaRasterizer::Rasterizer::forward   cuda_rasterizer/rasterizer.h:20-46,
 *                              called from RasterizeGaussiansCUDA    rasterize_points.cu:35-123
 *   goi_raster_backward     <- CudaRasterizer::Rasterizer::backward  cuda_rasterizer/rasterizer.h:75-111,
 *                              called from RasterizeGaussiansBackwardCUDA rasterize_points.cu:213-306
 *   goi_raster_trace        <- CudaRasterizer::Rasterizer::trace     cuda_rasterizer/rasterizer.h:48-73,
 *                              called from TraceGaussiansCUDA        rasterize_points.cu:125-211
 *   goi_raster_mark_visible <- CudaRasterizer::Rasterizer::markVisible cuda_rasterizer/rasterizer.h:13-18,
 *                              called from markVisible               rasterize_points.cu:308-327
 *   goi_raster_*_bytes      <- required<GeometryState/ImageState/BinningState>() cuda_rasterizer/rasterizer_impl.h:67-73
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors in the Python binding);
 *     no torch types cross this boundary; inputs are never written;
 *   - an absent optional input is NULL (the reference's "empty tensor => nullptr", rasterize_points.cu:98-111);
 *   - all arrays are contiguous fp32 unless stated; matrices are 16 floats in the reference's
 *     transposed (row-vector) convention (cuda_rasterizer/auxiliary.h:58-77);
 *   - the three workspaces are opaque bytes sized by goi_raster_*_bytes(); the binning workspace
 *     size depends on num_rendered, which is only known mid-call, so the caller passes an
 *     allocation callback (the reference's std::function<char*(size_t)>, rasterize_points.cu:27-33);
 *   - outputs need not be initialised by the caller; every element is written;
 *   - `stream` is a hipStream_t (NULL = the default stream); all work is enqueued on it; the only
 *     host synchronisation is the read-back of num_rendered inside forward/trace (the reference
 *     has the same one, cuda_rasterizer/rasterizer_impl.cu:285); goi_raster_forward_async has none;
 *   - entry points may be called concurrently from several host threads on different streams / devices
 *     (read-back tickets are pooled per device under a mutex; last_error is per thread); the
 *     goi_raster_set_option switches are process-wide but every call works with the snapshot it took when it
 *     started (a switch flipped by another thread never changes a call half-way); the stage profile is process-wide;
 *   - return value: >= 0 on success (forward/trace: num_rendered), < 0 on error with the message
 *     available from goi_raster_last_error() (thread-local).
 *   - supported S (semantic channels): any 1..32; fast paths are instantiated for 10 and 16.
 *   - alignment: device pointers as torch / hipMalloc hand them out (256 bytes) are always fine.  What the kernels actually
 *     assume: `semantics` 16-byte aligned when S is a multiple of 4 (its rows are moved as 16-byte words, by LDS-DMA in the
 *     backward), the workspaces 256-byte aligned, everything else 4 bytes.
 */
#ifndef GOI_RASTER_H
#define GOI_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6: + goi_raster_backward3 (row scratch sized by the frame's count instead of its capacity), goi_raster_blend_stats
 * 5: + goi_raster_forward_async_cut, goi_raster_ticket_result2 (speculative depth cut-off of the tile lists), goi_raster_backward2; the binning and
 *    backward-scratch workspaces grew (member masks; descriptors of big Gaussians): sizes come from goi_raster_*_bytes as ever
 * 4: + goi_raster_truncated_flag, goi_adam_step_guarded; a truncated speculative frame back-propagates ZERO gradients
 * (3: + goi_raster_forward_reblend, goi_codebook_sim, goi_codebook_fused; 2: + the asynchronous forward; additions only) */
#define GOI_RASTER_ABI_VERSION 6

typedef struct GoiRasterScene {
    int P;                       /* number of Gaussians */
    int D;                       /* active SH degree, 0..3 */
    int M;                       /* SH coefficients per colour channel held in shs (0 if shs == NULL) */
    int S;                       /* semantic channels */
    int W, H;                    /* image width, height */
    const float* bg;             /* [3] */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL; 16-byte aligned when 3 M is a multiple of 4 (rows move as 16-byte words) */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* semantics;      /* [P,S] (forward/backward) */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] or NULL */
    float scale_modifier;
    const float* rotations;      /* [P,4] (r,x,y,z) or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* viewmatrix;     /* [16] */
    const float* projmatrix;     /* [16] */
    const float* campos;         /* [3] */
    float tan_fovx, tan_fovy;
    int prefiltered;
    int debug;                   /* synchronise and check after every stage */
} GoiRasterScene;

/* Device-memory allocation callback: must return a device pointer to at least `bytes` bytes that
 * stays valid until the matching backward has run (NULL = failure). */
typedef void* (*goi_alloc_fn)(void* user, size_t bytes);

/* Thread safety: entry points may be called from several host threads (one per stream); the last-error
 * string and the pinned read-back buffer are per thread.  The profiling hooks and goi_raster_set_option touch
 * process-wide state and are meant for single-threaded measurement runs. */
int goi_raster_abi_version(void);
const char* goi_raster_last_error(void);

size_t goi_raster_geom_bytes(int P);
size_t goi_raster_image_bytes(int W, int H);
size_t goi_raster_binning_bytes(int num_rendered);
size_t goi_raster_backward_scratch_bytes(int num_rendered, int S);

/* Forward: color[3,H,W], semantic[S,H,W], depth[H,W], alpha[H,W], radii[P] (int32). */
int goi_raster_forward(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer,
                       goi_alloc_fn binning_alloc, void* alloc_user,
                       float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                       int* radii, void* stream);

/* ---- Speculative forward: the same frame WITHOUT the host round trip -----------------------------------------------
 * (no counterpart in the reference: CudaRasterizer::Rasterizer::forward blocks on num_rendered at
 * cuda_rasterizer/rasterizer_impl.cu:285 to size the binning workspace; SURVEY.md section 7 "no host sync").
 *
 * goi_raster_forward_async enqueues the WHOLE frame and returns at once.  The caller supplies the binning workspace
 * up front, sized by goi_raster_binning_bytes(capacity) for a capacity it believes to be >= num_rendered (e.g. twice
 * the largest count seen so far); on the device every kernel takes the true count from the geometry workspace and
 * clamps it to `capacity`.  The return value is a TICKET (>= 0) for the asynchronous read-back of the count:
 *
 *   goi_raster_ticket_result(ticket, wait, &n): 1 = the count has arrived (n = num_rendered; the ticket is released),
 *       0 = not yet (only with wait == 0), -1 = error (e.g. the "prefiltered" trap; ticket released).
 *   n <= capacity : the frame is exactly what goi_raster_forward would have produced (bit-identical outputs; the
 *       tile lists are identical, only the workspace is larger).  Pass R = capacity to goi_raster_backward.
 *   n >  capacity : OVERFLOW.  The frame was rendered from the first `capacity` instances in emit (depth) order --
 *       memory-safe, but not the right image.  The device knows (emit sets the frame's "truncated" word in the
 *       geometry workspace, goi_raster_truncated_flag): goi_raster_backward / _backward_semantics on such a frame write
 *       ZERO into every gradient, so a truncated frame never trains anything even if the host has not looked at the
 *       count yet (the reference sizes its buffers from the true count and cannot truncate, CR/rasterizer_impl.cu:283-289);
 *       goi_adam_step_guarded can skip the optimiser step of that view on the device as well.
 *       goi_raster_forward_redo re-runs emit -> tile sort -> blend from the geometry state that is still in the
 *       workspace, into a binning buffer of goi_raster_binning_bytes(n) bytes: afterwards outputs and workspaces are
 *       bit-identical to goi_raster_forward's, and R = n.
 *
 * goi_raster_forward_redo reads only P, S, W, H, semantics and bg from `scene` (the other fields may be NULL).
 * Nothing here waits for the device unless asked to (wait != 0).  Every ticket must be resolved exactly once.
 * A resolved ticket means the COUNT has arrived -- it is final before the frame's blend kernel starts -- not that the frame's
 * kernels have finished: the outputs are ordered on `stream` like any other kernel's.  A blocking wait spins on host memory;
 * after two seconds it synchronises the frame's stream (not the device) and fails loudly rather than hang.
 * Not available with scene->debug (which synchronises after every stage) or for P == 0. */
int goi_raster_forward_async(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer, void* binning_buffer,
                             int capacity, float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                             int* radii, void* stream);
int goi_raster_ticket_result(int ticket, int wait, int* num_rendered);

/* SPECULATIVE DEPTH CUT-OFF of the tile lists (no counterpart in the reference, which lists every Gaussian in every tile of its
 * rectangle, CR/rasterizer_impl.cu:70-111, although a pixel stops at T < 1e-4, CR/forward.cu:352-357: on an opaque scene three
 * quarters of the instances lie behind their tile's saturation front and are emitted, sorted and never looked at).
 *
 * goi_raster_forward_async_cut is goi_raster_forward_async with two per-TILE arrays (ceil(W/16) * ceil(H/16) floats, row-major):
 *   zcut_out (or NULL): LEARNT by this frame -- for every tile, the view depth up to which its list is worth listing the next
 *       time the same camera is rendered: the depth of the list entry a quarter (+ 64 positions) beyond the last position any
 *       pixel of the tile looked at; +inf for a tile in which some pixel reached the end of its list unsaturated.
 *   zcut_in (or NULL): APPLIED to this frame -- a Gaussian deeper than zcut_in[t] is not listed in tile t (rectangles of up to
 *       64 tiles, cull_variant 2: the cut lives in the ellipse tile masks).  Hand in what an earlier frame of the SAME camera
 *       learnt (never an uninitialised array).
 * A frame whose pixels all saturate inside their cut lists is bit-identical to the uncut frame: outputs, n_contrib, every
 * gradient (the dropped instances were never looked at).  If a pixel of a cut tile reaches the end of its list unsaturated, or
 * looks at an entry DEEPER than its tile's cut (a rectangle of more than 64 tiles has no tile mask and stays listed at every
 * depth: the Gaussians that were dropped in between would have come first), the
 * cut was TOO TIGHT for this frame (the geometry has moved since it was learnt): the forward blend raises bit 2 of the frame's
 * flag word (goi_raster_truncated_flag; bit 0 = the instance list was truncated, bit 1 = a sort timed out), the frame's
 * backward writes ZERO gradients like a truncated frame's, and goi_raster_ticket_result2 hands the word to the host, which
 * renders the frame again without a cut (goi_raster_forward) and forgets what the camera had learnt.
 * goi_raster_ticket_result2: as goi_raster_ticket_result, plus the frame's flag word (0 for a frame of goi_raster_forward). */
int goi_raster_forward_async_cut(const GoiRasterScene* scene, void* geom_buffer, void* image_buffer, void* binning_buffer,
                                 int capacity, float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                                 int* radii, const float* zcut_in, float* zcut_out, void* stream);
int goi_raster_ticket_result2(int ticket, int wait, int* num_rendered, unsigned* frame_flags);
#define GOI_FRAME_TRUNCATED 1u
#define GOI_FRAME_MISSORTED 2u
#define GOI_FRAME_CUT_TOO_TIGHT 4u
/* Device pointer (inside the geometry workspace of a frame over P Gaussians) of the frame's "truncated" word: non-zero
 * iff the frame's instance list did not fit its binning capacity.  Written by every forward (0 for goi_raster_forward
 * and after goi_raster_forward_redo); read on the device by the backward kernels and, if handed over, by
 * goi_adam_step_guarded.  NULL for P <= 0. */
const uint32_t* goi_raster_truncated_flag(const void* geom_buffer, int P);
int goi_raster_forward_redo(const GoiRasterScene* scene, int num_rendered, void* geom_buffer, void* image_buffer,
                            void* binning_buffer, float* out_color, float* out_semantic, float* out_depth,
                            float* out_alpha, const int* radii, void* stream);

/* The blend of a frame whose geometry and tile lists are already in workspaces filled by an earlier goi_raster_forward /
 * _async / _redo of the SAME camera over the SAME Gaussian geometry (positions, covariances, opacities, colours): only the
 * semantic rows and the background may have changed.  Runs the forward blend alone (no preprocess, sort or emit), writes the
 * four outputs and a fresh image workspace (`image_buffer`, goi_raster_image_bytes; the tile ranges are copied from
 * `cached_image_buffer`), and leaves geometry and binning workspaces untouched, so the result is what goi_raster_forward
 * would produce for the current semantics, bit for bit, and goi_raster_backward* can follow with (geom, binning, the NEW
 * image workspace, R).  R as for goi_raster_backward.  Reads only P, S, W, H, semantics and bg from `scene`.  What it cannot
 * check is the premise: the caller vouches that nothing but the semantics changed (goi_hyperplane_amd: opt-in geometry cache,
 * DESIGN.md 7c).  No counterpart in the reference, which redoes the whole frame (CR/rasterizer_impl.cu:198-344). */
int goi_raster_forward_reblend(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                               const void* cached_image_buffer, void* image_buffer, float* out_color, float* out_semantic,
                               float* out_depth, float* out_alpha, void* stream);

/* Backward of the forward that filled the three workspaces.  R = the instance count the binning workspace was laid out
 * for: goi_raster_forward's return value, or the `capacity` of a goi_raster_forward_async frame (the kernels read the
 * true count from the geometry workspace), or num_rendered after goi_raster_forward_redo.
 * Any of the four upstream gradients dL_dout_* may be NULL (= zero).  dL_dconic is [P,4] (x: a, y: b, z: unused, w: c), dL_dsh [P,M,3] (may be NULL when M == 0).
 * dL_dconic and dL_ddepth are WORKSPACES (CR/backward.cu hands them from its render pass to its preprocess pass; no caller of
 * the reference reads them): with a scratch buffer and option "bwd_records" 1 (the default) the blend gradients travel
 * between the two passes inside the scratch instead and these two arrays are left unwritten.
 * FACTORED mode: dL_dsh == NULL while the scene has SH colours.  dL/dSH is not formed (192 of the 300 bytes of
 * gradient per Gaussian at degree 3); dL_dcolor returns the colour gradient with the forward's clamp mask applied
 * (CR/backward.cu:44-47), the factor g of dL/dSH[k] = basis_k(view direction) * g -- see goi_raster_sh_grad_from_views. */
int goi_raster_backward(const GoiRasterScene* scene, int R,
                        const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                        const int* radii, const float* out_alpha,
                        const float* dL_dout_color, const float* dL_dout_semantic,
                        const float* dL_dout_depth, const float* dL_dout_alpha,
                        float* dL_dmean2D /*[P,3]*/, float* dL_dconic /*[P,4]*/, float* dL_dopacity /*[P]*/,
                        float* dL_dcolor /*[P,3]*/, float* dL_dsemantic /*[P,S]*/, float* dL_ddepth /*[P]*/,
                        float* dL_dmean3D /*[P,3]*/, float* dL_dcov3D /*[P,6]*/, float* dL_dsh /*[P,M,3]*/,
                        float* dL_dscale /*[P,3]*/, float* dL_drot /*[P,4]*/,
                        void* scratch /* goi_raster_backward_scratch_bytes(R, S) bytes, uninitialised; NULL selects
                                         the float-atomic accumulation path (not bit-reproducible) */,
                        void* stream);

/* goi_raster_backward with one more pointer (ABI 5).  prev_radii (or NULL = goi_raster_backward): the radii array of the
 * backward that LAST WROTE these very output buffers -- all eleven of them, e.g. views of one allocation a binding keeps and
 * reuses once its consumers have let go of it -- provided nothing else has written to them since.  A Gaussian with
 * prev_radii == 0 that is invisible in this frame as well already has zeros in every output row, and nothing is written for it:
 * on the headline scene half of the Gaussians are invisible in any one view and the dense gradient tensors of the reference's
 * interface (rasterize_points.cu:252-262: eleven zero-filled [P, ..] tensors per call) are 170 MB of zeros per step.  The
 * results are those of goi_raster_backward bit for bit.  (Honoured on the default path -- row records; the other backward
 * variants write every row.)  csrc/torch_binding.cpp keeps such a pool and checks refcount and version counter before a reuse. */
int goi_raster_backward2(const GoiRasterScene* scene, int R, const void* geom_buffer, const void* binning_buffer,
                         const void* image_buffer, const int* radii, const float* out_alpha, const float* dL_dout_color,
                         const float* dL_dout_semantic, const float* dL_dout_depth, const float* dL_dout_alpha,
                         float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic,
                         float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                         float* dL_drot, void* scratch, const int* prev_radii, void* stream);

/* goi_raster_backward2 with the row scratch laid out for FEWER instances than the binning workspace (ABI 6).  `R` stays what the
 * binning buffer was laid out for (a speculative frame: its capacity); `scratch_instances` (0 = R) is what the scratch --
 * goi_raster_backward_scratch_bytes(scratch_instances, S) bytes -- holds rows for.  It must be >= the frame's num_rendered: a
 * caller that has READ the count of a speculative frame by the time it enqueues the backward (the loss usually sits in between)
 * passes the count and needs half the scratch of a frame sized by its capacity (headroom 2: 2.1 instead of 4.3 GB on the
 * headline view, 6.4 instead of 12.9 GB at 3 M Gaussians).  Same results bit for bit.
 * `flags` bit 0, ACCUMULATE: the eleven output arrays already hold the gradients of earlier views of the same batch (written by a
 * call without the bit, then possibly added to by calls with it) and this view's gradients are ADDED: the rows of a Gaussian that
 * is visible in this view are read, added to and written back, all other rows are left alone -- the sum over the K views of a batch
 * costs each view its visible rows once more instead of a dense [P, 75 + S] addition per view.  Default (record) path with dL_dsh
 * formed or no SH at all; prev_radii must be NULL.  (Not in the reference: its loop back-propagates one view per optimiser step,
 * train.py:96-198; dist.backward_views uses it for multi-view batches.) */
#define GOI_BACKWARD_ACCUMULATE 1
int goi_raster_backward3(const GoiRasterScene* scene, int R, int scratch_instances, int flags, const void* geom_buffer,
                         const void* binning_buffer, const void* image_buffer, const int* radii, const float* out_alpha,
                         const float* dL_dout_color, const float* dL_dout_semantic, const float* dL_dout_depth,
                         const float* dL_dout_alpha, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                         float* dL_dsemantic, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, void* scratch, const int* prev_radii, void* stream);

/* Feature-gradient-only backward: dL/dsemantics [P,S] from dL/d(semantic map) alone, bit-identical to
 * the dL_dsemantic of goi_raster_backward and about 3x cheaper.  For the reference's default training
 * configuration, where only the semantic features are optimised (arguments/__init__.py:85-90,
 * scene/gaussian_model.py:185-246).  Same workspaces, `R` and scratch as goi_raster_backward. */
int goi_raster_backward_semantics(const GoiRasterScene* scene, int R, const void* geom_buffer,
                                  const void* binning_buffer, const void* image_buffer, const int* radii,
                                  const float* out_alpha, const float* dL_dout_semantic, float* dL_dsemantic,
                                  void* scratch, void* stream);

/* Trace: scene->semantics is ignored; img_sem[S,H,W] is scattered onto the Gaussians it meets
 * with alpha > 0.005.  out_color[3,H,W], gau_sem[P,S], num_gsem[P] (int32), radii[P]. */
int goi_raster_trace(const GoiRasterScene* scene, const float* img_sem, void* geom_buffer, void* image_buffer,
                     goi_alloc_fn binning_alloc, void* alloc_user,
                     float* out_color, float* gau_sem, int* num_gsem, int* radii, void* stream);

/* present[P] (bytes, 0/1): view-space z > 0.2. */
int goi_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* present, void* stream);

/* ---- semantic head, inference decode ----------------------------------------
 * Replaces, per pixel, gui/main.py:364-386 of the reference with scene/semantic_model.py used as one
 * Linear(S -> n_codes, bias):  idx = argmax_c (W[c,:] . f + b[c]);  sim = code_score[idx];
 * sim < thresh -> background (sim = 0).  `sem` is the rasterizer's semantic output [S, HW]
 * (channel-major, no permute); W is [n_codes, S] row-major (torch Linear.weight); code_score[n_codes]
 * is the host-folded tail (LUT lookup -> L2 normalise -> LinearSVM/VLM score).  Any of sim_out[HW],
 * idx_out[HW] (int32), bg_mask_out[HW] (bytes) may be NULL.  S <= 32. */
int goi_semantic_decode(const float* sem, int S, long long HW, const float* W, const float* bias, int n_codes,
                        const float* code_score, float thresh, float* sim_out, int* idx_out, uint8_t* bg_mask_out,
                        void* stream);

/* ---- measurement hooks (bench.py): per-stage HIP-event timing on the launch stream ---------- */
enum {
    GOI_STAGE_PREPROCESS = 0,   /* forward per-Gaussian kernel */
    GOI_STAGE_DEPTH_SORT,       /* radix sort of Gaussians by depth */
    GOI_STAGE_SCAN,             /* prefix sum of tiles_touched + num_rendered read-back */
    GOI_STAGE_EMIT,             /* (tile, Gaussian) instance emission */
    GOI_STAGE_TILE_SORT,        /* stable radix sort of instances by tile */
    GOI_STAGE_RANGES,           /* per-tile [start,end) */
    GOI_STAGE_BLEND_FWD,        /* forward alpha blend */
    GOI_STAGE_BLEND_BWD,        /* backward alpha blend */
    GOI_STAGE_PREPROCESS_BWD,   /* cov2D + projection + SH + cov3D backward */
    GOI_STAGE_COUNT
};
/* on != 0: record events around every stage of subsequent calls (adds event overhead). */
void goi_raster_profile_enable(int on);
/* Same, restricted to the stages whose bit (1u << GOI_STAGE_x) is set; 0 turns profiling off.  Timing
 * one stage costs two event records per call instead of two per stage. */
void goi_raster_profile_stages(unsigned stage_mask);
/* Synchronises the recorded events and ADDS each stage's elapsed milliseconds and launch count
 * since the last reset into ms[GOI_STAGE_COUNT] / calls[GOI_STAGE_COUNT]; then resets. */
int goi_raster_profile_collect(double* ms, int* calls);

/* Tuning / experiment switches; the defaults are the shipped configuration.
 *   "fwd_variant"  1 (default) two candidates per loop trip in the forward blend, 0 one; 2 the EXPERIMENTAL 16 pixels x 4 list
 *                  entries mapping (csrc/render_fwd_g4.hip; S <= 16, frames without a depth cut): same n_contrib, alpha, member
 *                  masks and gradients bit for bit, channel sums differ by one fp32 association; 1.7x slower;
 *                  3 EXPERIMENT (S = 16): a contributing Gaussian's feature row reaches the packed FMAs as scalar operands through
 *                  scalar loads instead of broadcast LDS reads (bit-identical; 1.08x slower); 4 EXPERIMENT (S <= 28): the channel
 *                  sums as fp32 outer products on v_mfma_f32_32x32x1_2b_f32 (decisions and gradients bit-identical; 1.16x slower)
 *                  -- DESIGN.md section 9, profiles/r06_blend_bounds.txt
 *   "bwd_variant"  0 (default) atomic-free backward; the per-Gaussian sums over pixels run at the 16-bit matrix rate on
 *                  split operands that keep fp32 accuracy (two f16 planes of exactly scaled values, all four partial
 *                  products, fp32 accumulation: indistinguishable from the fp32 chain at the noise level of two builds of
 *                  the reference -- profiles/r04_flush_equivalence*.json; bit-reproducible); 2 the same with exact-fp32
 *                  MFMA (one fp32 FMA chain per output); 1 workgroup-per-tile backward with float atomics (what
 *                  scratch = NULL selects)
 *   "sort_variant" 1 (default) onesweep radix sort, 0 histogram / scan / scatter per pass
 *   "sort_lookback" 1 (default) onesweep passes of up to 640 tiles find their prefix with the GROUPED look-back (a tile adds up the
 *                  aggregates of its group, then the totals of the groups in front: two round trips), 0 always the chained
 *                  decoupled look-back; same order either way
 *   "sort_tickets" 1 (default) a onesweep workgroup takes its tile from a ticket counter, so that a tile only ever waits for tiles
 *                  that are running whatever order the hardware starts workgroups in; 0 EXPERIMENT: tile = workgroup index
 *                  (measures what the tickets cost: 5 us of the headline step, 37 us at 3 M Gaussians; NOT safe as a default:
 *                  HIP promises no dispatch order)
 *   "sort_small"   0 (default) sorts of up to 2 M keys use 1024 x 4-key tiles; 1 they take the adaptive 512 x (2..16) tile that
 *                  larger sorts choose from the device-side count (slower for them: DESIGN.md 8.2); same order either way
 *   "pre_shdma"    0 (default) every lane of the per-Gaussian forward fetches its own degree-3 SH row; 1 EXPERIMENT: the wave moves
 *                  the rows of its visible Gaussians to LDS by DMA (coalesced, culled rows never requested): measured slower at
 *                  1 M Gaussians (67 vs 63.5 us), level at 3 M.  Bit-identical
 *   "cull_variant" 2 (default) a Gaussian is listed only in the tiles its contribution ellipse (alpha >= 1/255) reaches,
 *                  1 in the tiles its axis-aligned contribution box touches, 0 in the reference's 3-sigma squares.
 *                  Identical images; gradients equal up to the order of one fp32 sum
 *   "decode_variant" (goi_semantic_decode, S <= 16) 1 (default) contraction as three bf16 MFMAs on exact 3-way splits of
 *                  the fp32 operands (fp32 accuracy), two 16-pixel blocks per code-book operand fetch; 2 / 3 the same
 *                  with four / one block per fetch (bit-identical results, slower); 0 fp32 MFMA
 *   "bwd_order"    1 (default) the backward's quadrant waves are launched longest-first inside each XCD's band (their
 *                  cost is known from the forward; cost classes of 16 list positions), 2 .. 4 the same with classes of 32 ..
 *                  128 positions (closer to tile order: less HBM traffic, less balance), 0 in tile order; same gradients
 *   "bwd_records"  1 (default) the per-Gaussian sums of the atomic-free backward stay in the scratch as one record per
 *                  listed Gaussian and the per-Gaussian pass writes every per-id output; 0 they go through six per-id
 *                  arrays (dL_dconic, dL_ddepth, ... and zeros for the Gaussians that are not listed); same gradients, bit for bit;
 *                  2 EXPERIMENT (128-byte rows: S = 5 .. 20): the per-Gaussian pass sums its Gaussians' rows itself -- no record,
 *                  no reduce_rows_k; bit-identical, measured slower (285 -> 437 us on the headline view: DESIGN.md 9.3)
 * Thread safety: the set is changed under a mutex; an entry point snapshots it when it starts. */
int goi_raster_set_option(const char* name, int value);

/* dL/dSH [P,M,3] of V views from the factors goi_raster_backward leaves in FACTORED mode: means3D [P,3], the V camera
 * centres campos [V,3] and the clamp-masked colour gradients gcol [V,P,3]:
 *     dL_dsh[g][k] = sum_v basis_k(normalize(means3D[g] - campos[v])) * gcol[v][g]
 * with the basis of CR/backward.cu:49-109 (degree D, (D+1)^2 <= M <= 16; coefficients above (D+1)^2 get 0), views added
 * in index order: bit-identical to adding the per-view dL_dsh arrays of goi_raster_backward in that order.  Lets a
 * data-parallel job exchange 12 bytes per Gaussian and view (all-gather) instead of all-reducing 192. */
int goi_raster_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* gcol,
                           float* dL_dsh, void* stream);

/* ---- simple_knn._C.distCUDA2 (submodules/simple-knn/ext.cpp:15-17, spatial.cu:15-26,
 * simple_knn.cu:170-221): mean squared distance of every point to its 3 nearest OTHER points,
 * mean_dist2[i] = (d0 + d1 + d2) / 3 in fp32.  points [P,3] and mean_dist2 [P] are device pointers;
 * workspace holds goi_knn_workspace_bytes(P) bytes of device memory (256-byte aligned).  With fewer
 * than 4 points the missing neighbours count as FLT_MAX, as in the reference.  Asynchronous on
 * `stream`; no host read-back. */
size_t goi_knn_workspace_bytes(int P);
int goi_knn_dist2(int P, const float* points, float* mean_dist2, void* workspace, void* stream);

/* ---- training losses of the semantic head, row pass (train.py:142-163; see csrc/codebook_loss.hip).
 * Inputs: sim_raw [HW][C] = <g_p, LUT_c/|LUT_c|> with g NOT normalised, inv_gnorm [HW] = 1/|g_p|,
 * sem [S][HW] (the rasterizer's channel-major feature map), decoder W [C][S] and bias [C] (or NULL),
 * anneal factor t (1 or 2).  Outputs: dsim [HW][C] = dL/dsim_raw, dsem [S][HW] = dL/dsem, and
 * partials [goi_codebook_loss_partial_rows()][C*(S+1)+4]: per persistent wave the dL/dW rows (S
 * values then dL/db per code) followed by the sums over its pixels of (sum_c (P-label)^2, max sim,
 * entropy, sim at the decoder's argmax); the caller adds the rows up.  L = lab + sl + 0.3 sl1 + recc
 * with upstream gradient 1.  1 <= S <= 16, 1 <= C <= 512.  All pointers are device pointers. */
int goi_codebook_loss_partial_rows(void);
int goi_codebook_loss_rows(const float* sim_raw, const float* inv_gnorm, const float* sem, const float* W,
                           const float* bias, long long HW, int C, int S, float t, float* dsim, float* dsem,
                           float* partials, void* stream);

/* sim_raw [HW][C] = g^T * L1^T and inv_gnorm [HW] = 1/|g_p| in one pass over g [D][HW] (the channel-major ground-truth
 * map), L1 [C][D] the row-normalised code book: the dense code-book x feature contraction of train.py:147-149 on the bf16
 * matrix rate with split (hi + lo) operands, fp32 accumulation (csrc/codebook_loss.hip: codebook_sim_k; products to 2^-16,
 * sim to ~1e-6).  workspace: goi_codebook_sim_workspace_bytes() device bytes.  Supported shape: D = 256, C <= 304, C % 4 = 0;
 * returns < 0 otherwise (use a library GEMM then). */
size_t goi_codebook_sim_workspace_bytes(void);
int goi_codebook_sim(const float* g, const float* lut1, long long HW, int C, int D, float* sim, float* inv_gnorm,
                     void* workspace, void* stream);

/* dL/dL1 [C][D] = dsim^T * g^T as a split-K fp32 MFMA GEMM over the pixel axis (csrc/codebook_loss.hip):
 * dsim [HW][C] (from goi_codebook_loss_rows), g [D][HW] (the channel-major ground-truth map).  Writes
 * partial [goi_codebook_dlut_partial_blocks()][304][D]; the caller sums over the first axis and keeps
 * rows < C.  Supported shape: D = 256, 288 < C <= 304, HW % 4 = 0; returns < 0 otherwise (use a
 * library GEMM then). */
int goi_codebook_dlut_partial_blocks(void);
int goi_codebook_dlut(const float* dsim, const float* g, long long HW, int C, int D, float* partial, void* stream);

/* The three steps above as one call with no [HW][C] fp32 matrix in memory (csrc/codebook_loss.hip: decoder_stats_k,
 * codebook_simgrad_k, decoder_gd_k, codebook_dlut2_k): g [D][HW], lut1 [C][D] (rows normalised), sem [S][HW], W [C][S], bias [C] or NULL, t as in
 * goi_codebook_loss_rows.  Writes dsem [S][HW], partials [goi_codebook_fused_partial_rows()][C*(S+1)+4] (same row format
 * as goi_codebook_loss_rows) and dlut_partial [goi_codebook_dlut_partial_blocks()][304][D]; the caller sums both over the
 * first axis.  workspace: goi_codebook_fused_workspace_bytes(HW) device bytes (dL/dsim as two bf16 planes: 4 bytes per
 * (pixel, code), plus 64 B of records per pixel).  Supported shape: D = 256, 288 < C <= 304, 1 <= S <= 16, HW % 4 = 0,
 * HW < 2^25; anything else returns -1 and the caller uses the three separate entry points.  Reference: train.py:142-163. */
size_t goi_codebook_fused_workspace_bytes(long long HW);
int goi_codebook_fused_partial_rows(void);
int goi_codebook_fused(const float* g, const float* lut1, const float* sem, const float* W, const float* bias, long long HW,
                       int C, int D, int S, float t, float* dsem, float* partials, float* dlut_partial, void* workspace,
                       void* stream);

/* ---- fused Adam step over the Gaussian parameter groups (scene/gaussian_model.py:163-253:
 * torch.optim.Adam(lr=0.0, eps=1e-15) over xyz / f_dc / f_rest / semantics / opacity / scaling /
 * rotation; train.py:193 optimizer.step()) with the optional per-Gaussian gradient mask of
 * gui/main.py:480-513 (clear_noralative_gs_grad: rows with mask != 0 see a zero gradient).
 * One launch for up to GOI_ADAM_MAX_GROUPS tensors.  All pointers are device pointers to fp32,
 * 16-byte aligned; numel = P * row_len. */
#define GOI_ADAM_MAX_GROUPS 8
typedef struct GoiAdamGroup {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long numel;
    int row_len;      /* elements per Gaussian (mask granularity); >= 1 */
    float step_size;  /* lr / (1 - beta1^t), rounded to fp32 once */
    float bc2_sqrt;   /* sqrt(1 - beta2^t) */
} GoiAdamGroup;
int goi_adam_step(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                  const unsigned char* nograd_mask /*[P] or NULL*/, void* stream);
/* The same with a device-side guard: if *skip_flag (a device word, e.g. goi_raster_truncated_flag of the view the
 * gradients came from) is non-zero when the kernel runs, nothing is updated -- parameters and both moments keep their
 * values (the host-side step count of the caller is the caller's business).  skip_flag == NULL: goi_adam_step. */
int goi_adam_step_guarded(const GoiAdamGroup* groups, int n_groups, double beta1, double beta2, double eps,
                          const unsigned char* nograd_mask /*[P] or NULL*/, const uint32_t* skip_flag, void* stream);

/* Lane utilisation of the blend kernels, counted on the device from what the forward of a frame left in its workspaces
 * (member masks, n_contrib, per-quadrant walk lengths): a diagnostic of the execution mapping, not part of the reference's
 * interface (its blend loops, one thread per pixel: CR/forward.cu:330-372, CR/backward.cu:523-589).  R / the three buffers:
 * as for goi_raster_backward of the same frame.  counters: DEVICE array of GOI_BLEND_STATS_WORDS 64-bit words, cleared
 * and filled on `stream`:
 *   0 quadrants (8x8 pixels = one wave) that composited anything      1 rounds of 64 list positions they walk
 *   2 list positions in those rounds up to the quadrant's last contributor
 *   3 of those, candidates passing the forward's quadrant hit test (the pairs the forward blend evaluates)
 *   4 MEMBER pairs: (quadrant, Gaussian) with a contribution to some pixel (the pairs the backward evaluates and flushes)
 *   5 live lanes: (pixel, Gaussian) contributions = lanes doing useful work, summed over the member pairs
 *   6 / 7 / 8 (block, Gaussian) pairs with a live lane were a wave split into 8x4 / 4x4 / 2x2 pixel blocks
 *   9 member pairs without a live lane (0 by construction)   10 pixels inside the image   11 sum of n_contrib */
#define GOI_BLEND_STATS_WORDS 16
int goi_raster_blend_stats(int P, int W, int H, int R, const void* geom_buffer, const void* binning_buffer,
                           const void* image_buffer, unsigned long long* counters, void* stream);

/* Inspection of the opaque workspaces (tests only): copies device -> caller DEVICE buffers.
 * Any pointer may be NULL.  point_list is in final sorted order. */
int goi_raster_debug_views(int P, int W, int H, int R, const void* geom_buffer, const void* binning_buffer,
                           const void* image_buffer,
                           float* depths /*[P]*/, float* means2D /*[P,2]*/, float* conic_opacity /*[P,4]*/,
                           float* rgb /*[P,3]*/, uint32_t* tiles_touched /*[P]*/, uint32_t* point_list /*[R]*/,
                           uint32_t* ranges /*[T,2]*/, uint32_t* n_contrib /*[H*W]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif
