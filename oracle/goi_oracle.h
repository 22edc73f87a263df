/*
 * oracle/goi_oracle.h -- C ABI of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The oracle is a plain C++ restatement, on the host CPU, of the arithmetic of the
 * reference's tile-based differentiable Gaussian rasterizer with a semantic channel
 * (submodules/diff-gaussian-rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped HIP path (goi_hyperplane_amd/csrc) never links, calls or falls back to it.
 *
 * PARITY STATUS: see the header of goi_oracle.cpp ("parity partially pinned").
 */
#ifndef GOI_ORACLE_H
#define GOI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GoiOracleScene {
    int P;               /* number of Gaussians */
    int D;               /* active SH degree 0..3 */
    int M;               /* SH coefficients per channel held in `shs` (0 if shs == NULL) */
    int S;               /* semantic channels */
    int W, H;            /* image size */
    const float* bg;             /* [3] */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* semantics;      /* [P,S] */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] or NULL */
    float scale_modifier;
    const float* rotations;      /* [P,4] or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* viewmatrix;     /* [16] transposed (row-vector) convention */
    const float* projmatrix;     /* [16] */
    const float* campos;         /* [3] */
    float tan_fovx, tan_fovy;
    int prefiltered;
} GoiOracleScene;

/* Opaque forward state (the oracle's own GeometryState/BinningState/ImageState). */
typedef struct GoiOracleState GoiOracleState;

GoiOracleState* goi_oracle_state_new(void);
void goi_oracle_state_free(GoiOracleState*);

/* Forward. Outputs: color[3,H,W], semantic[S,H,W], depth[H,W], alpha[H,W], radii[P].
 * fragile[H,W] (may be NULL): 1 where some (pixel, Gaussian) pair sat within `fragile_eps`
 * (relative) of one of the discontinuous blend guards.  Returns num_rendered (>=0). */
int goi_oracle_forward(const GoiOracleScene* sc, GoiOracleState* st,
                       float* out_color, float* out_semantic, float* out_depth, float* out_alpha,
                       int* radii, uint8_t* fragile, float fragile_eps, int num_threads);

/* Backward of the forward that filled `st`. All outputs are float32, caller-zeroed or not
 * (the oracle overwrites).  dL_dsh may be NULL when M == 0. */
int goi_oracle_backward(const GoiOracleScene* sc, const GoiOracleState* st,
                        const float* out_alpha,
                        const float* dL_dpix, const float* dL_dpixsem,
                        const float* dL_dpix_depth, const float* dL_dalphas,
                        float* dL_dmean2D /*[P,3]*/, float* dL_dconic /*[P,4]*/,
                        float* dL_dopacity /*[P]*/, float* dL_dcolor /*[P,3]*/,
                        float* dL_dsemantic /*[P,S]*/, float* dL_ddepth /*[P]*/,
                        float* dL_dmean3D /*[P,3]*/, float* dL_dcov3D /*[P,6]*/,
                        float* dL_dsh /*[P,M,3]*/, float* dL_dscale /*[P,3]*/,
                        float* dL_drot /*[P,4]*/, int num_threads);

/* Trace (image -> Gaussian feature scatter), deterministic-sum restatement.
 * img_sem[S,H,W]; outputs out_color[3,H,W], gau_sem[P,S], num_gsem[P]. */
int goi_oracle_trace(const GoiOracleScene* sc, GoiOracleState* st, const float* img_sem,
                     float* out_color, float* gau_sem, int* num_gsem, int* radii, int num_threads);

void goi_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix,
                             const float* projmatrix, uint8_t* present);

/* Introspection of the forward state (for stage-by-stage parity of the HIP path). */
int goi_oracle_state_counts(const GoiOracleState* st, int* P, int* N, int* T);
/* Copies; any pointer may be NULL. Sizes: per-Gaussian arrays P, point_list N, ranges 2*T, n_contrib HW. */
void goi_oracle_state_get(const GoiOracleState* st,
                          float* depths, float* means2D /*[P,2]*/, float* conic_opacity /*[P,4]*/,
                          float* rgb /*[P,3]*/, float* cov3D /*[P,6]*/, uint8_t* clamped /*[P,3]*/,
                          uint32_t* tiles_touched, uint32_t* point_list, uint64_t* point_list_keys,
                          uint32_t* ranges, uint32_t* n_contrib);

#ifdef __cplusplus
}
#endif
#endif
