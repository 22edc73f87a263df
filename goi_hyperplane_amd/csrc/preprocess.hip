// Per-Gaussian kernels: forward preprocess (project, EWA cov2D, conic, radius, tile rectangle,
// SH -> RGB, render record) and its backward (conic -> cov3D/mean, projection, depth, SH, cov3D ->
// scale/rotation).  Compiled with -ffp-contract=off: these stages are HBM-bound streaming, and
// keeping the reference's operation order makes radii / tile rectangles / depth keys integer-exact.
//
// Behaviour restated from the reference (paths relative to submodules/diff-gaussian-rasterization/):
//   forward : cuda_rasterizer/forward.cu:155-256 (+ :20-71 SH, :74-113 cov2D, :118-152 cov3D),
//             cuda_rasterizer/auxiliary.h:41-56,139-164
//   backward: cuda_rasterizer/backward.cu:144-274 (cov2D), :346-412 (projection/depth),
//             :20-139 (SH), :278-341 (cov3D)
// Layout is this library's own: one 48-byte GaussRec per Gaussian instead of five arrays, depth
// keys for the per-Gaussian depth sort, 3 clamp bits in one byte.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gmath.h"
#include "row_sum.h"

namespace goi {

namespace {

__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ M3 rotation_from_quat(float r, float x, float y, float z) {
    return make_m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

struct Cov2D {
    M3 T, Vrk, W;
    V3 t;
    float txtz, tytz, limx, limy;
    M3 cov;
};

__device__ __forceinline__ void ewa_cov2d(V3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                          const float* cov3D, const float* view, Cov2D& c) {
    V3 t = xform_point_4x3(mean, view);
    c.limx = 1.3f * tan_fovx;
    c.limy = 1.3f * tan_fovy;
    c.txtz = t.x / t.z;
    c.tytz = t.y / t.z;
    t.x = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * t.z;
    t.y = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * t.z;
    c.t = t;
    M3 J = make_m3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0.f, 0.f, 0.f);
    c.W = make_m3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c.T = mul(c.W, J);
    c.Vrk = make_m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    c.cov = mul(mul(transpose(c.T), transpose(c.Vrk)), c.T);
}

struct PreArgs {
    int P, D, M, W, H, gx, gy, prefiltered, cull;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    const float* view_p;    // device, [16]
    const float* proj_p;    // device, [16]
    const float* campos_p;  // device, [3]
    // speculative depth cut-off of the tile lists (api.hip, goi_raster_forward_async_cut): zcut[t] = view depth beyond which
    // tile t's list was never walked when this camera was rendered last (NULL: no cut); zlearn[t]: what this frame learns
    // (cleared here, written by the forward blend)
    const float* zcut;
    uint32_t* zlearn;
};

// The camera arrays are tiny, uniform device arrays: every thread reads them through the scalar
// cache into registers once.
struct Camera {
    float view[16], proj[16], campos[3];
};
__device__ __forceinline__ Camera load_camera(const float* __restrict__ v, const float* __restrict__ p,
                                              const float* __restrict__ c) {
    Camera cam;
#pragma unroll
    for (int i = 0; i < 16; i++) cam.view[i] = v[i];
#pragma unroll
    for (int i = 0; i < 16; i++) cam.proj[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; i++) cam.campos[i] = c[i];
    return cam;
}

// SH rows by LDS-DMA (SHDMA: SH colours with M = 16, the degree-3 layout).  A lane's colour is a sequential fp32 sum over its
// own 192-byte row (the reference's order: the clamp bits depend on it), so the arithmetic stays per lane -- but a lane FETCHING
// its row (sixteen 12-byte or twelve 16-byte loads at a 192-byte lane stride) makes every load instruction of the wave touch 64
// different lines, and with 20+ waves per CU a line is fetched from L2 again for most of them: SH -> RGB was 6-12 us of a wave's
// ~20.  Now the WAVE moves the rows of its visible lanes: global_load_lds_dwordx4 sends 16 bytes per lane from
// memory straight to LDS (no registers), consecutive lanes taking consecutive 16-byte quads of consecutive VISIBLE rows (ranked
// by a ballot; a culled Gaussian's row is never requested), so an instruction reads ~5 rows as ~10 whole lines; the requests
// are issued as soon as the visibility is known and travel under the contribution box / ellipse tile mask work; then every
// visible lane reads its row from LDS (row stride 13 quads = 52 dwords: eight lanes of a ds_read_b128 cover the 32 banks once)
// and runs the same statement sequence as before: bit-identical colours and clamp bits.  LDS holds SH_ROWS_CAP rows per wave
// (7.3 KB: five waves per SIMD, what the registers allow); the ~1 in 6 waves with more visible lanes take a second trip.
#ifndef GOI_PRE_SH_ROWS
#define GOI_PRE_SH_ROWS 36
#endif
constexpr int SH_ROWS_CAP = GOI_PRE_SH_ROWS;
constexpr int SH_ROW_QUADS = 13;  // 12 quads of coefficients + 1 of padding

template <typename SHK_T>
__device__ __forceinline__ V3 sh_to_rgb_sum(int D, const V3& dir, SHK_T SHK) {
    V3 res = kSH0 * SHK(0);
    if (D > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        res = res - kSH1 * y * SHK(1) + kSH1 * z * SHK(2) - kSH1 * x * SHK(3);
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + kSH2[0] * xy * SHK(4) + kSH2[1] * yz * SHK(5) + kSH2[2] * (2.0f * zz - xx - yy) * SHK(6) +
                  kSH2[3] * xz * SHK(7) + kSH2[4] * (xx - yy) * SHK(8);
            if (D > 2) {
                res = res + kSH3[0] * y * (3.0f * xx - yy) * SHK(9) + kSH3[1] * xy * z * SHK(10) +
                      kSH3[2] * y * (4.0f * zz - xx - yy) * SHK(11) +
                      kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHK(12) +
                      kSH3[4] * x * (4.0f * zz - xx - yy) * SHK(13) + kSH3[5] * z * (xx - yy) * SHK(14) +
                      kSH3[6] * x * (xx - 3.0f * yy) * SHK(15);
            }
        }
    }
    return res + V3{0.5f, 0.5f, 0.5f};
}

__device__ __forceinline__ void sh_dma16(const void* global_src, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(global_src, lds_base, 16, 0, 0);
#endif
}

#ifndef GOI_PRE_MINBLOCKS
#define GOI_PRE_MINBLOCKS 1
#endif
template <bool SHDMA>
__global__ __launch_bounds__(256, GOI_PRE_MINBLOCKS) void preprocess_fwd_k(const PreArgs args, GaussRec* __restrict__ rec,
                                                        float* __restrict__ cov3D_out,
                                                        uint32_t* __restrict__ tiles_touched,
                                                        uint8_t* __restrict__ clamped, uint32_t* __restrict__ raw_key,
                                                        uint4* __restrict__ aux,
                                                        uint2* __restrict__ blk_agg,
                                                        unsigned long long* __restrict__ blk_coarse, int* __restrict__ radii,
                                                        uint32_t* __restrict__ counters, uint2* __restrict__ ranges,
                                                        int n_tiles) {
    __shared__ __attribute__((aligned(16))) float4 s_sh[SHDMA ? 4 : 1][SHDMA ? SH_ROWS_CAP * SH_ROW_QUADS : 1];
    __shared__ uint8_t s_rank[SHDMA ? 4 : 1][64];  // rank among the wave's visible lanes -> lane
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    // the tile ranges start from zero (emit accumulates per-tile counts into them): cleared here for free
    for (int t = gtid; t < n_tiles; t += gridDim.x * blockDim.x) ranges[t] = make_uint2(0u, 0u);
    if (args.zlearn)
        for (int t = gtid; t < n_tiles; t += gridDim.x * blockDim.x) args.zlearn[t] = 0u;
    if (gtid == 0) counters[COUNTER_CULL] = (uint32_t)args.cull;  // emit and the backward list the same rectangles
    // every lane reaches the wave reduction at the end: lanes past P redo the last Gaussian and write nothing
    const bool live = gtid < args.P;
    const int idx = live ? gtid : args.P - 1;
    const Camera cam = load_camera(args.view_p, args.proj_p, args.campos_p);
    struct : PreArgs {
        const float *view, *proj, *campos;
    } a;
    static_cast<PreArgs&>(a) = args;
    a.view = cam.view;
    a.proj = cam.proj;
    a.campos = cam.campos;
    int my_radius_i = 0;
    unsigned long long tmask_v = TMASK_FULL;
    uint32_t touched = 0, key = 0xFFFFFFFFu;
    uint8_t clamp_bits = 0;

    const V3 p = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    const V3 p_view = xform_point_4x3(p, a.view);
    // ---- phase A: projection, covariance, conic, radius, 3-sigma rectangle: is the Gaussian rendered at all?
    bool vis = false;
    float pix = 0.f, piy = 0.f, con_a = 0.f, con_b = 0.f, con_c = 0.f, my_radius = 0.f;
    do {
        if (p_view.z <= 0.2f) {
            if (a.prefiltered) atomicOr(&counters[1], 1u);  // the reference traps here
            break;
        }
        const float4 p_hom = xform_point_4x4(p, a.proj);
        const float p_w = 1.0f / (p_hom.w + 0.0000001f);
        const float projx = p_hom.x * p_w, projy = p_hom.y * p_w;

        float cov3D[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = a.cov3D_precomp[(size_t)6 * idx + i];
        } else {
            M3 S = make_m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
            S.c[0][0] = a.scale_modifier * a.scales[3 * idx];
            S.c[1][1] = a.scale_modifier * a.scales[3 * idx + 1];
            S.c[2][2] = a.scale_modifier * a.scales[3 * idx + 2];
            const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];  // (r,x,y,z), not normalised
            M3 R = rotation_from_quat(q.x, q.y, q.z, q.w);
            M3 Mm = mul(S, R);
            M3 Sg = mul(transpose(Mm), Mm);
            cov3D[0] = Sg.c[0][0]; cov3D[1] = Sg.c[0][1]; cov3D[2] = Sg.c[0][2];
            cov3D[3] = Sg.c[1][1]; cov3D[4] = Sg.c[1][2]; cov3D[5] = Sg.c[2][2];
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D_out[(size_t)6 * idx + i] = cov3D[i];
        }
        Cov2D c2;
        ewa_cov2d(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.view, c2);
        const float cx = c2.cov.c[0][0] + 0.3f, cy = c2.cov.c[0][1], cz = c2.cov.c[1][1] + 0.3f;
        const float det = cx * cz - cy * cy;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        con_a = cz * det_inv;
        con_b = -cy * det_inv;
        con_c = cx * det_inv;
        const float mid = 0.5f * (cx + cz);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        pix = ndc_to_pix(projx, a.W);
        piy = ndc_to_pix(projy, a.H);
        int x0, y0, x1, y1;
        tile_rect(pix, piy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        vis = live;  // (a lane past P has redone the last Gaussian up to here: it must neither rank among the wave's rows -- its
                     // "row" would lie beyond the array -- nor store anything)
    } while (false);

    // ---- (SHDMA) the wave requests the SH rows of its visible lanes: rank r of the visible lanes -> row r of the wave's LDS area
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long vm = 0ull;
    int nvis = 0, rank = 0;
    const int wave_first = blockIdx.x * 256 + wv * 64;
    auto request_rows = [&](int r0) {  // rows of ranks [r0, r0 + SH_ROWS_CAP): quad k of rank r travels to slot 13 r + k
        const int nslots = min(SH_ROWS_CAP, nvis - r0) * SH_ROW_QUADS;
#pragma unroll
        for (int i = 0; i < (SH_ROWS_CAP * SH_ROW_QUADS + 63) / 64; i++) {
            if (i * 64 >= nslots) break;  // (wave-uniform)
            const int sl = i * 64 + lane;
            const int r = (sl * 5042) >> 16, q = sl - SH_ROW_QUADS * r;  // sl / 13 (exact below 1100), sl % 13
            if (sl < nslots && q < SH_ROW_QUADS - 1) {
                const int src = s_rank[wv][r0 + r];
                sh_dma16(a.shs + ((size_t)(wave_first + src) * 48 + 4 * q), &s_sh[wv][i * 64]);
            }
        }
    };
    if constexpr (SHDMA) {
        vm = __ballot(vis);
        nvis = __popcll(vm);
        rank = __popcll(vm & ((1ull << lane) - 1ull));
        if (vis) s_rank[wv][rank] = (uint8_t)lane;
        __builtin_amdgcn_wave_barrier();
        if (nvis > 0) request_rows(0);
    }

    // ---- phase B: contribution box, listed rectangle, ellipse tile mask (nothing here needs the colour)
    float hx = -1.f, hy = -1.f, o = 0.f;
    if (vis) {
        // Box outside which alpha = min(0.99, o*exp(power)) < 1/255 is certain (see DESIGN.md,
        // "exact contribution box"): the ellipse power >= -tau of the COMPUTED conic, tau inflated.
        o = a.opacities[idx];
        if (o >= 1.0f / 255.0f) {
            // In single precision, every rounding pushed OUTWARD (the box only has to contain the region; its size
            // decides nothing but how many tiles and quadrants are looked at): the double-precision log / sqrt / divide
            // this replaces were software routines of a few hundred instructions on the critical path of every visible lane.
            //   tau = 1.01 ln(255 o) + 0.01, rounded up;  det = a c - b b by Kahan's difference of products (within
            //   1.5 ulp even when the two products cancel: a needle), rounded down;  h = sqrt(2 tau c / det), rounded up.
            const float tau = fmaf(1.01f, logf(255.0f * o), 0.01f) * 1.000002f + 1e-6f;
            const float bb = con_b * con_b;
            const float detc = (fmaf(con_a, con_c, -bb) + fmaf(-con_b, con_b, bb));
            const float det_lo = detc - 4e-7f * fabsf(detc);
            if (det_lo > 0.f && con_a > 0.f && con_c > 0.f) {
                hx = sqrtf(2.0f * tau * con_c / det_lo) * 1.000002f + 1e-3f;
                hy = sqrtf(2.0f * tau * con_a / det_lo) * 1.000002f + 1e-3f;
                if (!(hx == hx) || !(hy == hy)) hx = hy = __builtin_inff();
            } else {
                hx = hy = __builtin_inff();
            }
        }
        my_radius_i = (int)my_radius;
        int x0, y0, x1, y1;
        listed_rect(pix, piy, my_radius_i, hx, hy, a.cull != 0, a.gx, a.gy, x0, y0, x1, y1);
        touched = (uint32_t)((y1 - y0) * (x1 - x0));  // may be 0 for a visible Gaussian (radius stays > 0)
        // cull_variant 2: of that rectangle, only the tiles the contribution ELLIPSE  1/2 d^T C d <= tau  reaches (the
        // box over-covers by 1 - pi/4 for a round Gaussian, by most of its area for an elongated diagonal one: x0.80
        // instances on the headline scene).  Exact for a convex set, one tile row at a time: over the row's pixel-centre
        // band y in [16 t, 16 t + 15], clipped to the ellipse's own extent, the ellipse spans x in [L, R] with
        //     R(dy) = (-b dy + sqrt(2 a tau - det dy^2)) / a   (concave: its maximum over the band is at the band's
        // point nearest to the ellipse's rightmost point dy = -(b/c) hx), L likewise (convex, leftmost point); the row's
        // tiles are those whose pixel-centre range [16 t, 16 t + 15] meets [L, R].  tau carries the box's inflation (1 % +
        // 0.01: the blend kernels' evaluation error of `power` can never move a contributing pixel outside), every
        // rounding here is pushed outward, and a NaN keeps the whole row.  A tile that is dropped can not receive a
        // contribution; the per-pixel sequences of contributing Gaussians, hence all outputs and gradients, are untouched.
        if (a.cull >= 2 && touched > 0 && touched <= 64 && hx < 1e30f && hy < 1e30f) {
            const int w = x1 - x0;
            const float tau = (fmaf(1.01f, logf(255.0f * o), 0.01f) * 1.000002f + 1e-6f) * 1.0001f + 1e-4f;
            const float bb = con_b * con_b;
            const float det = fmaxf((fmaf(con_a, con_c, -bb) + fmaf(-con_b, con_b, bb)) * (1.f - 4e-7f), 0.f);
            const float inv_a = 1.f / con_a;
            const float dyR = -(con_b / con_c) * hx, dyL = -dyR;  // where the ellipse is rightmost / leftmost
            unsigned long long mask = 0ull;
            for (int ry = 0; ry < y1 - y0; ry++) {
                const float lo = fmaxf((float)((y0 + ry) * TILE) - piy, -hy), hi = fminf((float)((y0 + ry) * TILE + TILE - 1) - piy, hy);
                if (!(lo <= hi)) {
                    if (lo == lo && hi == hi) continue;  // the band misses the ellipse's extent
                }
                const float d1 = fminf(fmaxf(dyR, lo), hi), d2 = fminf(fmaxf(dyL, lo), hi);
                const float s1 = sqrtf(fmaxf(2.f * con_a * tau - det * d1 * d1, 0.f));
                const float s2 = sqrtf(fmaxf(2.f * con_a * tau - det * d2 * d2, 0.f));
                float R = (-con_b * d1 + s1) * inv_a, L = (-con_b * d2 - s2) * inv_a;
                R += 1e-3f + 4e-6f * fabsf(R);
                L -= 1e-3f + 4e-6f * fabsf(L);
                // columns t with 16 t <= pix + R and 16 t + 15 >= pix + L
                int c0 = (int)ceilf((pix + L - (float)(TILE - 1)) / TILE), c1 = (int)floorf((pix + R) / TILE) + 1;
                if (!(R == R) || !(L == L)) {
                    c0 = x0;
                    c1 = x1;
                }
                c0 = max(c0, x0);
                c1 = min(c1, x1);
                if (c1 > c0) mask |= ((c1 - c0 >= 64) ? ~0ull : ((1ull << (c1 - c0)) - 1ull)) << (ry * w + (c0 - x0));
            }
            // speculative depth cut-off: a tile whose list was not needed beyond depth zcut[t] the last time this camera was
            // rendered does not list a deeper Gaussian (the forward blend raises the frame's flag if a pixel of such a tile
            // turns out to want more: api.hip).  Eight tiles per trip, their cut depths requested together (one load round
            // trip for the typical 3 x 3 rectangle; tile by tile this loop cost the kernel 28 us).
            if (a.zcut) {
                unsigned long long todo = mask;
                while (todo) {
                    int bits[8];
                    float zc[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        bits[i] = todo ? __builtin_ctzll(todo) : -1;
                        if (todo) todo &= todo - 1;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int bsafe = max(bits[i], 0);
                        zc[i] = a.zcut[(size_t)(y0 + bsafe / w) * a.gx + (x0 + bsafe % w)];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (bits[i] >= 0 && zc[i] < p_view.z) mask &= ~(1ull << bits[i]);
                }
            }
            tmask_v = mask;
            touched = (uint32_t)__popcll(mask);
        }
        key = __float_as_uint(p_view.z);
    }

    // everything that does not depend on the colour leaves NOW: the values are dead while the SH rows are in registers
    if (vis) {
        float4* r4 = reinterpret_cast<float4*>(rec + idx);
        r4[0] = make_float4(pix, piy, con_a, con_b);
        r4[1] = make_float4(con_c, o, hx, hy);       // what the hit test needs beside q0: fetched for every CANDIDATE
    }
    if (live) {
        radii[idx] = my_radius_i;
        tiles_touched[idx] = touched;
        raw_key[idx] = key;  // depth bits by Gaussian id; compact_listed_k keeps the listed ones for the depth sort
        aux[idx] = make_uint4(0u, (uint32_t)my_radius_i, (uint32_t)tmask_v, (uint32_t)(tmask_v >> 32));  // (.x: emit)
    }

    // ---- phase C: the colour
    float cr = 0.f, cg = 0.f, cb = 0.f;
    V3 dir = {0.f, 0.f, 0.f};
    if (vis && !a.colors_precomp) {
        const V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
        dir = p - campos;
        dir = dir / sqrtf(dot3(dir, dir));
    }
    auto finish_colour = [&](const V3& res) {
        clamp_bits = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
        cr = fmaxf(res.x, 0.0f);
        cg = fmaxf(res.y, 0.0f);
        cb = fmaxf(res.z, 0.0f);
    };
    if constexpr (SHDMA) {
        for (int r0 = 0; r0 < nvis; r0 += SH_ROWS_CAP) {  // (wave-uniform: one trip for five waves in six)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows are in LDS
            __builtin_amdgcn_wave_barrier();
            if (vis && rank >= r0 && rank < r0 + SH_ROWS_CAP) {
                const float4* row = &s_sh[wv][(rank - r0) * SH_ROW_QUADS];
                float rowf[48];
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const float4 v = row[i];
                    rowf[4 * i] = v.x;
                    rowf[4 * i + 1] = v.y;
                    rowf[4 * i + 2] = v.z;
                    rowf[4 * i + 3] = v.w;
                }
                finish_colour(sh_to_rgb_sum(a.D, dir, [&](int k) { return V3{rowf[3 * k], rowf[3 * k + 1], rowf[3 * k + 2]}; }));
            }
            __builtin_amdgcn_wave_barrier();
            if (r0 + SH_ROWS_CAP < nvis) request_rows(r0 + SH_ROWS_CAP);
        }
    } else if (vis) {
        if (a.colors_precomp) {
            cr = a.colors_precomp[3 * idx];
            cg = a.colors_precomp[3 * idx + 1];
            cb = a.colors_precomp[3 * idx + 2];
        } else {
            // A degree-3 row (M = 16: 192 bytes, 16-byte aligned) is fetched as TWELVE 16-byte loads instead of sixteen 12-byte ones
            // (this instance only runs when the LDS-DMA path above is switched off, GOI_OPTIONS pre_shdma=0: the A/B of it).
            float rowf[48];
            const bool row16 = a.M == 16;
            if (row16) {
                const float4* r4 = reinterpret_cast<const float4*>(a.shs + (size_t)idx * 48);
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const float4 v = r4[i];
                    rowf[4 * i] = v.x;
                    rowf[4 * i + 1] = v.y;
                    rowf[4 * i + 2] = v.z;
                    rowf[4 * i + 3] = v.w;
                }
            }
            const V3* shg = reinterpret_cast<const V3*>(a.shs) + (size_t)idx * a.M;
            finish_colour(sh_to_rgb_sum(a.D, dir, [&](int k) { return row16 ? V3{rowf[3 * k], rowf[3 * k + 1], rowf[3 * k + 2]} : shg[k]; }));
        }
    }
    // what only a HIT needs (r, g, b, depth: the first staged feature quad as it is)
    if (vis) reinterpret_cast<float4*>(rec + idx)[2] = make_float4(cr, cg, cb, p_view.z);
    if (live) clamped[idx] = clamp_bits;
    // num_rendered = sum of tiles_touched: order-independent, so it is formed HERE (one integer atomic per wave)
    // instead of falling out of the prefix sum after the depth sort -- the host can read it while the sort runs
    __shared__ uint32_t s_wsum[4], s_wvis[4];
    uint32_t wsum = live ? touched : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, d, 64);
    const uint32_t wvis = (uint32_t)__popcll(__ballot(live && touched > 0));  // LISTED Gaussians of this wave
    if ((threadIdx.x & 63) == 0) {
        s_wsum[threadIdx.x >> 6] = wsum;
        s_wvis[threadIdx.x >> 6] = wvis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bsum = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        if (bsum) atomicAdd(&counters[NR_BASE + NR_STRIDE * (blockIdx.x % NR_STRIPES)], bsum);
        // the block's aggregate: compact_listed_k ranks the listed Gaussians with it
        const uint32_t bvis = s_wvis[0] + s_wvis[1] + s_wvis[2] + s_wvis[3];
        blk_agg[blockIdx.x] = make_uint2(bvis, bsum);
        // ... and the sum over COARSE_BLOCKS consecutive blocks (compact_listed_k's base rank), packed: one integer atomic
        if (bvis) atomicAdd(&blk_coarse[(size_t)(blockIdx.x / COARSE_BLOCKS) * COARSE_STRIDE], ((unsigned long long)bvis << 40) | bsum);
    }
}

struct BwdArgs {
    int P, D, M, W, H;
    const float* means3D;
    const float* shs;
    const float* scales;
    const float* rotations;
    const float* cov3D;  // precomputed or the forward's
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    const float* view_p;
    const float* proj_p;
    const float* campos_p;
    // radii of the backward that LAST WROTE these very output buffers (nothing has touched them since), or NULL: a Gaussian that
    // was invisible then and is invisible now already has zeros in every output row -- nothing is written for it (half of the
    // headline scene: 170 MB of zeros per step).  goi_raster_backward2, csrc/torch_binding.cpp: the gradient-buffer pool.
    const int* prev_radii;
    // ACCUMULATE (goi_raster_backward3, flags bit 0; the record path only): the outputs already hold the gradients of earlier
    // views of the same batch -- a visible Gaussian's rows are read, added to and written back, an invisible one's are left
    // alone.  The sum of K views then costs each view its visible rows once more instead of a dense [P, 75 + S] addition
    // (dist.backward_views).
    int accumulate;
};

// WITH_DSH: dL/dSH is formed ([P,M,3], staged through LDS).  Otherwise (the caller passed dL_dsh = NULL with SH
// colours: "factored" mode of goi_raster_backward) the kernel writes the clamp-masked colour gradient g back to
// dL_dcolor instead: dL/dSH[k] = basis_k(view direction) * g is then formed elsewhere (goi_raster_sh_grad_from_views).
// FROM_ROWS: the blend gradients of a Gaussian come from its RECORD in the row scratch (reduce_rows_k<.., RECORD>: the summed
// row over the Gaussian's first slot, goff[id] * 4) instead of six per-id arrays, and this kernel writes the per-id outputs
// the reduction used to write -- dL/dmean2D, dL/dcolour, dL/dopacity, dL/dsemantics -- itself: zeros for a Gaussian that is
// not listed, coalesced either way.  (dL_dconic / dL_ddepth are not written on that path: they were only ever this
// kernel's inputs.)
struct RecArgs {
    const float* rows;              // row scratch
    const uint4* aux;               // [P] .x: first emit-order instance of a listed Gaussian
    const uint32_t* tiles_touched;  // [P] 0: not listed (no record)
    int row_floats, S, nch;         // nch = padded semantic channels + 4 (see render_bwd.hip: BwdCfg)
    float* dL_dopacity;
    float* dL_dsemantic;
    // SRC == 2 (bwd_records 2): the kernel sums the rows itself
    const uint8_t* flags;           // validity bytes of the row slots
    const uint32_t* n_dev;          // the frame's counters from COUNTER_N on (instances, listed Gaussians: the BIG threshold)
    uint32_t N_cap;                 // slot capacity the scratch was laid out for
};
// Rows of the dL/dSH staging tile = visible Gaussians a workgroup takes through the chain at a time: 224 x (3 M + 1) floats =
// 43.9 KB at M = 16, three workgroups per CU with the index lists (the kernel's 143 VGPRs allow three as well).
#ifndef GOI_PBWD_ROWS
#define GOI_PBWD_ROWS 224
#endif
#ifndef GOI_PBWD_BLOCKS
#define GOI_PBWD_BLOCKS 3
#endif
#ifndef GOI_PBWD_HOIST
#define GOI_PBWD_HOIST 1
#endif
constexpr int BWD_TILE_ROWS_DEFAULT = GOI_PBWD_ROWS;
#ifndef GOI_PBWD_INFLIGHT
#define GOI_PBWD_INFLIGHT 16  // rows a quarter wave requests back to back in the in-kernel row sum (reduce_rows_k: 32)
#endif
#ifndef GOI_PBWD_ROWS_FUSED
#define GOI_PBWD_ROWS_FUSED 208
#endif
constexpr int BWD_TILE_ROWS_FUSED = GOI_PBWD_ROWS_FUSED;  // (SRC == 2: ten more floats of LDS per row; 208 rows keep three workgroups per CU)
constexpr int BWD_REC_FLOATS = 10;  // r, g, b, depth, mean2D x, y, conic a, b, c, opacity
constexpr int BWD_BLOCKS_PER_CU = GOI_PBWD_BLOCKS;
constexpr bool BWD_HOIST = GOI_PBWD_HOIST != 0;

// SRC: where the blend gradients of a Gaussian come from -- 0 the six per-id arrays, 1 its RECORD in the row scratch
// (reduce_rows_k<.., RECORD>), 2 the kernel SUMS THE ROWS ITSELF (bwd_records 2; 128-byte rows): before a tile's chain, the
// workgroup's sixteen quarter waves walk the tile's Gaussians, each summing one Gaussian's rows exactly as reduce_rows_k does
// (same function, same slot order: bit-identical sums; slot range two Gaussians ahead, validity bytes one ahead); the 16
// semantic sums leave for dL/dsemantics at once, the ten values the chain needs go to LDS.  No record is written or read back,
// no second kernel walks the listed Gaussians; only the BIG ones (reduce_big_k) still pass through a record.
template <bool WITH_DSH, int SRC>
__global__ __launch_bounds__(256, GOI_PBWD_BLOCKS) void preprocess_bwd_k(const BwdArgs args, const int* __restrict__ radii,
                                                        const uint32_t* __restrict__ counters,
                                                        const uint8_t* __restrict__ clamped,
                                                        float* dL_dmean2D,
                                                        const float* __restrict__ dL_dconic,
                                                        float* dL_dcolor,
                                                        const float* __restrict__ dL_ddepth, const RecArgs ra,
                                                        float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
                                                        float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
                                                        float* __restrict__ dL_drot) {
    // The work of this kernel is per VISIBLE Gaussian (~600 instructions of chain rule, a 128-byte record, 192 bytes of SH in and
    // 368 bytes of gradients out); an invisible one needs zeros in its output rows, or -- when the rows still hold the zeros of
    // the backward that last wrote the buffers (BwdArgs::prev_radii) -- nothing at all.  With one thread per Gaussian id every wave
    // ran the whole chain at the scene's visible fraction of its lanes (51 % on the headline view, 8 % on a close-up of a 3 M
    // scene, where the kernel took 264 us for 242 k visible Gaussians), and every block walked all 256 rows of its dL/dSH tile.
    // Now a PERSISTENT workgroup walks segments of 256 ids (segment = blockIdx.x, + gridDim.x, ...): it CLASSIFIES a segment's ids
    // (visible / needs zeros / nothing to do) with ballots and a scan of the wave counts, writes the zeros at once, and APPENDS the
    // visible ids to a pending list; whenever BWD_TILE_ROWS of them are pending (or the input is exhausted) thread t runs the chain
    // for the t-th pending id -- dense lanes whatever the visible fraction -- and the block streams out exactly the tile rows
    // that exist.  Per Gaussian the arithmetic is the statement sequence it always was (-ffp-contract=off): bit-identical
    // gradients, whichever block and lane a Gaussian lands on.
    // dL/dSH (192 B per Gaussian at degree 3) still leaves through an LDS tile (odd row stride: no bank conflicts) as
    // contiguous rows instead of 48 stores at a 192-byte lane stride.
    constexpr bool FROM_ROWS = SRC != 0;
    constexpr int BWD_TILE_ROWS = SRC == 2 ? BWD_TILE_ROWS_FUSED : BWD_TILE_ROWS_DEFAULT;
    extern __shared__ float s_dsh[];  // [BWD_TILE_ROWS][3 M + 1] when dL_dsh
    __shared__ uint32_t s_vis[BWD_TILE_ROWS + 256];  // pending visible ids
    __shared__ uint16_t s_zero[256];
    __shared__ uint32_t s_src[SRC == 1 ? BWD_TILE_ROWS : 1];  // (SRC 1) the tile row's record: first slot of a listed Gaussian, ~0u: none
    __shared__ float s_rec[SRC == 2 ? BWD_TILE_ROWS * BWD_REC_FLOATS : 1];  // (SRC 2) the chain's ten inputs of every tile row
    __shared__ int s_wcnt[2][4];
    const Camera cam = load_camera(args.view_p, args.proj_p, args.campos_p);
    struct : BwdArgs {
        const float *view, *proj, *campos;
    } a;
    static_cast<BwdArgs&>(a) = args;
    a.view = cam.view;
    a.proj = cam.proj;
    a.campos = cam.campos;
    const int w = 3 * a.M;
    const int nseg = (args.P + 255) / 256;
    const bool truncated = counters[COUNTER_OVF] != 0;  // (a truncated frame is treated as if nothing were visible: all gradients zero)
    int npend = 0;  // (block-uniform)
    for (int seg = blockIdx.x; seg < nseg || npend > 0; seg += gridDim.x) {
    int nzero = 0;
    const int base = seg * 256;
    if (seg < nseg) {
        const int gtid = base + threadIdx.x;
        const bool live = gtid < args.P;
        const bool vis_t = live && radii[gtid] > 0 && !truncated;
        // rows that already hold zeros (see BwdArgs::prev_radii) are not written again
        const bool keep_t = live && !vis_t && args.prev_radii != nullptr && args.prev_radii[gtid] == 0;
        const bool zero_t = live && !vis_t && !keep_t && !args.accumulate;
        const int wv = threadIdx.x >> 6;
        const unsigned long long bv = __ballot(vis_t), bz = __ballot(zero_t);
        if ((threadIdx.x & 63) == 0) {
            s_wcnt[0][wv] = __popcll(bv);
            s_wcnt[1][wv] = __popcll(bz);
        }
        __syncthreads();
        int ov = 0, oz = 0;
        for (int i = 0; i < wv; i++) {
            ov += s_wcnt[0][i];
            oz += s_wcnt[1][i];
        }
        const int nv = s_wcnt[0][0] + s_wcnt[0][1] + s_wcnt[0][2] + s_wcnt[0][3];
        nzero = s_wcnt[1][0] + s_wcnt[1][1] + s_wcnt[1][2] + s_wcnt[1][3];
        const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
        if (vis_t) s_vis[npend + ov + __popcll(bv & lt)] = (uint32_t)gtid;
        if (zero_t) s_zero[oz + __popcll(bz & lt)] = (uint16_t)threadIdx.x;
        npend += nv;
        __syncthreads();
    }
    // ---- zeros for the invisible Gaussians whose rows do not hold them yet
    if (nzero > 0) {
        if ((int)threadIdx.x < nzero) {
            const int idx = base + s_zero[threadIdx.x];
            if constexpr (FROM_ROWS) {
                ra.dL_dopacity[idx] = 0.f;
                dL_dmean2D[3 * idx] = dL_dmean2D[3 * idx + 1] = dL_dmean2D[3 * idx + 2] = 0.f;
                dL_dcolor[3 * idx] = dL_dcolor[3 * idx + 1] = dL_dcolor[3 * idx + 2] = 0.f;
            }
            dL_dmean3D[3 * idx] = dL_dmean3D[3 * idx + 1] = dL_dmean3D[3 * idx + 2] = 0.f;
#pragma unroll
            for (int i = 0; i < 6; i++) dL_dcov3D[(size_t)6 * idx + i] = 0.f;
            dL_dscale[3 * idx] = dL_dscale[3 * idx + 1] = dL_dscale[3 * idx + 2] = 0.f;
            reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (FROM_ROWS) {  // dL/dsemantics: S floats per row
            for (int i = threadIdx.x; i < nzero * ra.S; i += 256) {
                const int r = i / ra.S;
                ra.dL_dsemantic[(size_t)(base + s_zero[r]) * ra.S + (i - r * ra.S)] = 0.f;
            }
        }
        if constexpr (WITH_DSH) {
            for (int i = threadIdx.x; i < nzero * w; i += 256) {
                const int r = i / w;
                dL_dsh[(size_t)(base + s_zero[r]) * w + (i - r * w)] = 0.f;
            }
        }
    }
    // ---- the pending visible Gaussians, BWD_TILE_ROWS at a time: thread t runs the chain for the t-th of them
    const bool last_trip = seg + (int)gridDim.x >= nseg;  // no further segment for this block: drain the list
    while (npend >= BWD_TILE_ROWS || (last_trip && npend > 0)) {
    const int nrows = min(BWD_TILE_ROWS, npend);
    const bool visible = (int)threadIdx.x < nrows;
    const int idx = visible ? (int)s_vis[threadIdx.x] : 0;
    // Everything a Gaussian's chain reads is REQUESTED here, before anything waits.  The kernel is latency bound: its waves
    // spent two thirds of their time in s_waitcnt (profiles/r04_c_pmc_summary.txt) walking ~10 dependent round trips -- each
    // input was fetched where the chain first needs it.  The SH rows (192 B, the widest input) are fetched by the whole block,
    // consecutive lanes taking consecutive 16-byte pieces of a row, into the tile the dL/dSH rows leave through (a thread reads
    // its coefficients from its tile row before it overwrites them with their gradients); the small per-Gaussian inputs go to
    // registers: two round trips per chunk (slot -> record; everything else beside the first).
    if constexpr (WITH_DSH && BWD_HOIST) {
        if ((w & 3) == 0) {
            const int w4 = w >> 2;
            for (int i = threadIdx.x; i < nrows * w4; i += 256) {
                const int r = i / w4, q = i - r * w4;
                const float4 v = reinterpret_cast<const float4*>(a.shs + (size_t)s_vis[r] * w)[q];
                float* dst = s_dsh + r * (w + 1) + 4 * q;
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
        } else {
            for (int i = threadIdx.x; i < nrows * w; i += 256) {
                const int r = i / w;
                s_dsh[r * (w + 1) + (i - r * w)] = a.shs[(size_t)s_vis[r] * w + (i - r * w)];
            }
        }
    }
    if constexpr (SRC == 2) {
        // ---- the tile's rows: quarter wave qw sums the rows of tile rows qw, qw + 16, ... (software pipeline as in reduce_rows_k)
        constexpr int RF = 32;  // 128-byte rows (the launcher takes this mode for them only)
        const int qw = threadIdx.x >> 4, e = threadIdx.x & 15, quarter = (threadIdx.x & 63) >> 4;
        const uint32_t* flags32 = reinterpret_cast<const uint32_t*>(ra.flags);
        const uint32_t N = min(ra.N_cap, ra.n_dev[0]);  // (a truncated frame has no visible Gaussians: never gets here)
        const uint32_t Vl = ra.n_dev[COUNTER_V - COUNTER_N];
        const uint32_t big_inst = (N > REDUCE_DENSE_RATIO * Vl) ? REDUCE_BIG_INST : 1024u;  // (reduce_rows.hip: find_big_k)
        const int rounds = (nrows + 15) >> 4;  // (block-uniform)
        struct Meta {
            uint32_t idx, off0, cnt;
        };
        auto load_meta = [&](int k) {
            const int r = qw + 16 * k;
            Meta m{0u, 0u, 0u};
            if (k < rounds && r < nrows) {
                m.idx = s_vis[r];
                m.cnt = ra.tiles_touched[m.idx];          // instances = slots / 4 of a listed Gaussian, 0: not listed
                m.off0 = ra.aux[m.idx].x;                 // its first emit-order instance
                if (m.cnt == 0u) m.off0 = 0u;
            }
            return m;
        };
        auto load_first = [&](const Meta& m) { return (m.cnt <= big_inst && (uint32_t)e < m.cnt) ? flags32[m.off0 + e] : 0u; };
        Meta cur = load_meta(0), nxt = load_meta(1);
        uint32_t w_cur = load_first(cur);
#pragma unroll 1
        for (int k = 0; k < rounds; k++) {
            const Meta nn = load_meta(k + 2);
            const uint32_t w_nxt = load_first(nxt);
            const int r = qw + 16 * k;
            const bool big = cur.cnt > big_inst;
            float sum[2] = {0.f, 0.f};
            sum_instances<2, false, GOI_PBWD_INFLIGHT>(ra.rows, flags32, (size_t)cur.off0, big ? 0u : cur.cnt, w_cur, quarter, e, sum, sum);
            if (r < nrows) {
                if (big) {  // its record, left by reduce_big_k over the first slot it owns
                    const float2 t = reinterpret_cast<const float2*>(ra.rows + (size_t)cur.off0 * 4 * RF)[e];
                    sum[0] = t.x;
                    sum[1] = t.y;
                }
                const int nsem = ra.nch - 4;  // lane e holds row elements 2 e, 2 e + 1
                const int el = 2 * e;
                if (el < nsem) {
                    float* dst = ra.dL_dsemantic + (size_t)cur.idx * ra.S;
                    if (el + 1 < ra.S && (ra.S & 1) == 0) *reinterpret_cast<float2*>(dst + el) = make_float2(sum[0], sum[1]);
                    else {
                        if (el < ra.S) dst[el] = sum[0];
                        if (el + 1 < ra.S) dst[el + 1] = sum[1];
                    }
                } else if (el < nsem + BWD_REC_FLOATS) {
                    *reinterpret_cast<float2*>(&s_rec[r * BWD_REC_FLOATS + (el - nsem)]) = make_float2(sum[0], sum[1]);
                }
            }
            cur = nxt;
            nxt = nn;
            w_cur = w_nxt;
        }
        __syncthreads();
    }
    const V3 mean_in = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float cov3D_in[6];
#pragma unroll
    for (int i = 0; i < 6; i++) cov3D_in[i] = a.cov3D[(size_t)6 * idx + i];
    const float4 rot_in = a.scales ? reinterpret_cast<const float4*>(a.rotations)[idx] : make_float4(1.f, 0.f, 0.f, 0.f);
    const V3 scale_in = a.scales ? V3{a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]} : V3{0.f, 0.f, 0.f};
    const uint8_t clamp_in = a.shs ? clamped[idx] : (uint8_t)0;
    V3 gmean = {0, 0, 0};
    float gcov[6] = {0, 0, 0, 0, 0, 0};
    V3 gscale = {0, 0, 0};
    float4 grot = make_float4(0, 0, 0, 0);
    // ---- the blend gradients: from the per-id arrays, or from the Gaussian's record
    float in_conic[3] = {0.f, 0.f, 0.f}, in_m2d[2] = {0.f, 0.f}, in_depth = 0.f;
    V3 in_col = {0.f, 0.f, 0.f};
    if constexpr (SRC == 2) {
        if (visible) {
            const float* rc = &s_rec[threadIdx.x * BWD_REC_FLOATS];
            in_col = V3{rc[0], rc[1], rc[2]};
            in_depth = rc[3];
            in_m2d[0] = rc[4];
            in_m2d[1] = rc[5];
            in_conic[0] = rc[6];
            in_conic[1] = rc[7];
            in_conic[2] = rc[8];
            ra.dL_dopacity[idx] = rc[9];
            dL_dmean2D[3 * idx] = in_m2d[0];
            dL_dmean2D[3 * idx + 1] = in_m2d[1];
            dL_dmean2D[3 * idx + 2] = 0.f;
            dL_dcolor[3 * idx] = in_col.x;  // (factored SH mode overwrites it with the clamp-masked gradient below)
            dL_dcolor[3 * idx + 1] = in_col.y;
            dL_dcolor[3 * idx + 2] = in_col.z;
        }
    } else if constexpr (SRC == 1) {
        const bool listed = visible && ra.tiles_touched[idx] != 0;
        const uint32_t slot = listed ? ra.aux[idx].x : 0u;
        const float* rec = ra.rows + (size_t)slot * 4 * ra.row_floats;
        const int nsem = ra.nch - 4;
        float opa = 0.f;
        if (listed) {
            const float4 cd = *reinterpret_cast<const float4*>(rec + nsem);        // r, g, b, depth
            const float4 mc = *reinterpret_cast<const float4*>(rec + ra.nch);      // mean2D x, y, conic a, b
            const float2 co = *reinterpret_cast<const float2*>(rec + ra.nch + 4);  // conic c, opacity
            in_col = V3{cd.x, cd.y, cd.z};
            in_depth = cd.w;
            in_m2d[0] = mc.x;
            in_m2d[1] = mc.y;
            in_conic[0] = mc.z;
            in_conic[1] = mc.w;
            in_conic[2] = co.x;
            opa = co.y;
        }
        if (visible) {
            s_src[threadIdx.x] = listed ? slot : 0xFFFFFFFFu;
            if (a.accumulate) {
                ra.dL_dopacity[idx] += opa;
                dL_dmean2D[3 * idx] += in_m2d[0];
                dL_dmean2D[3 * idx + 1] += in_m2d[1];
                dL_dcolor[3 * idx] += in_col.x;
                dL_dcolor[3 * idx + 1] += in_col.y;
                dL_dcolor[3 * idx + 2] += in_col.z;
            } else {
                ra.dL_dopacity[idx] = opa;
                dL_dmean2D[3 * idx] = in_m2d[0];
                dL_dmean2D[3 * idx + 1] = in_m2d[1];
                dL_dmean2D[3 * idx + 2] = 0.f;
                dL_dcolor[3 * idx] = in_col.x;  // (factored SH mode overwrites it with the clamp-masked gradient below)
                dL_dcolor[3 * idx + 1] = in_col.y;
                dL_dcolor[3 * idx + 2] = in_col.z;
            }
        }
    } else if (visible) {
        in_conic[0] = dL_dconic[4 * idx];
        in_conic[1] = dL_dconic[4 * idx + 1];
        in_conic[2] = dL_dconic[4 * idx + 3];
        in_m2d[0] = dL_dmean2D[3 * idx];
        in_m2d[1] = dL_dmean2D[3 * idx + 1];
        in_depth = dL_ddepth[idx];
        in_col = V3{dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    }
    const bool live = visible;  // (the chain below writes through `live`)
    V3* dsh = WITH_DSH ? reinterpret_cast<V3*>(s_dsh + (size_t)threadIdx.x * (3 * a.M + 1)) : nullptr;
    auto put = [&](int k, const V3& v) {
        if constexpr (WITH_DSH) dsh[k] = v;
    };

    if constexpr (WITH_DSH && BWD_HOIST) __syncthreads();  // the SH rows are in the tile
    if (visible) {
        const V3 mean = mean_in;
        // ---- conic -> cov2D -> cov3D and the covariance path of the mean gradient
        {
            float cov3D[6];
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_in[i];
            const float dca = in_conic[0], dcb = in_conic[1], dcc = in_conic[2];
            Cov2D c;
            ewa_cov2d(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.view, c);
            const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
            const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
            const auto& T = c.T.c;
            const auto& Vrk = c.Vrk.c;
            const auto& Wm = c.W.c;
            const float ca = c.cov.c[0][0] + 0.3f, cb = c.cov.c[0][1], cc = c.cov.c[1][1] + 0.3f;
            const float denom = ca * cc - cb * cb;
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            if (denom2inv != 0) {
                dL_da = denom2inv * (-cc * cc * dca + 2 * cb * cc * dcb + (denom - ca * cc) * dcc);
                dL_dc = denom2inv * (-ca * ca * dcc + 2 * ca * cb * dcb + (denom - ca * cc) * dca);
                dL_db = denom2inv * 2 * (cb * cc * dca - (denom + 2 * cb * cb) * dcb + ca * cb * dcc);
                gcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
                gcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
                gcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
                gcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
                          2 * T[1][0] * T[1][1] * dL_dc;
                gcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
                          2 * T[1][0] * T[1][2] * dL_dc;
                gcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
                          2 * T[1][1] * T[1][2] * dL_dc;
            }
            const float r0a = T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2];
            const float r0b = T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2];
            const float r0c = T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2];
            const float r1a = T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2];
            const float r1b = T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2];
            const float r1c = T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2];
            const float dL_dT00 = 2 * r0a * dL_da + r1a * dL_db;
            const float dL_dT01 = 2 * r0b * dL_da + r1b * dL_db;
            const float dL_dT02 = 2 * r0c * dL_da + r1c * dL_db;
            const float dL_dT10 = 2 * r1a * dL_dc + r0a * dL_db;
            const float dL_dT11 = 2 * r1b * dL_dc + r0b * dL_db;
            const float dL_dT12 = 2 * r1c * dL_dc + r0c * dL_db;
            const float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
            const float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
            const float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
            const float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
            const float tz = 1.f / c.t.z;
            const float tz2 = tz * tz;
            const float tz3 = tz2 * tz;
            const float dL_dtx = x_grad_mul * -a.focal_x * tz2 * dL_dJ02;
            const float dL_dty = y_grad_mul * -a.focal_y * tz2 * dL_dJ12;
            const float dL_dtz = -a.focal_x * tz2 * dL_dJ00 - a.focal_y * tz2 * dL_dJ11 +
                                 (2 * a.focal_x * c.t.x) * tz3 * dL_dJ02 + (2 * a.focal_y * c.t.y) * tz3 * dL_dJ12;
            gmean = xform_vec_4x3_t({dL_dtx, dL_dty, dL_dtz}, a.view);
        }
        // ---- projection and depth paths of the mean gradient
        {
            const float* proj = a.proj;
            const float* view = a.view;
            const float4 m_hom = xform_point_4x4(mean, proj);
            const float m_w = 1.0f / (m_hom.w + 0.0000001f);
            const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
            const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
            const float d2x = in_m2d[0], d2y = in_m2d[1];
            V3 g1;
            g1.x = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
            g1.y = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
            g1.z = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
            gmean = gmean + g1;
            const float mul3 = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
            const float dd = in_depth;
            V3 g2;
            g2.x = (view[2] - view[3] * mul3) * dd;
            g2.y = (view[6] - view[7] * mul3) * dd;
            g2.z = (view[10] - view[11] * mul3) * dd;
            gmean = gmean + g2;
        }
        // ---- SH backward (colour gradient -> SH coefficients and view-direction path of the mean)
        if (a.shs) {
            const V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
            const V3 dir_orig = mean - campos;
            const V3 dir = dir_orig / sqrtf(dot3(dir_orig, dir_orig));
            // (the tile row holds the Gaussian's SH coefficients until the put() pass at the end of this block overwrites them)
            const V3* sh = (WITH_DSH && BWD_HOIST) ? reinterpret_cast<const V3*>(s_dsh + (size_t)threadIdx.x * (3 * a.M + 1))
                                    : reinterpret_cast<const V3*>(a.shs) + (size_t)idx * a.M;
            const uint8_t cl = clamp_in;
            V3 g = in_col;
            g.x *= (cl & 1) ? 0.f : 1.f;
            g.y *= (cl & 2) ? 0.f : 1.f;
            g.z *= (cl & 4) ? 0.f : 1.f;
            if constexpr (!WITH_DSH) {
                if (live) {
                    dL_dcolor[3 * idx] = g.x;
                    dL_dcolor[3 * idx + 1] = g.y;
                    dL_dcolor[3 * idx + 2] = g.z;
                }
            }
            V3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
            const float x = dir.x, y = dir.y, z = dir.z;
            if (a.D > 0) {
                dRGBdx = -kSH1 * sh[3];
                dRGBdy = -kSH1 * sh[1];
                dRGBdz = kSH1 * sh[2];
                if (a.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    dRGBdx = dRGBdx + (kSH2[0] * y * sh[4] + kSH2[2] * 2.f * -x * sh[6] + kSH2[3] * z * sh[7] +
                                       kSH2[4] * 2.f * x * sh[8]);
                    dRGBdy = dRGBdy + (kSH2[0] * x * sh[4] + kSH2[1] * z * sh[5] + kSH2[2] * 2.f * -y * sh[6] +
                                       kSH2[4] * 2.f * -y * sh[8]);
                    dRGBdz = dRGBdz + (kSH2[1] * y * sh[5] + kSH2[2] * 2.f * 2.f * z * sh[6] + kSH2[3] * x * sh[7]);
                    if (a.D > 2) {
                        dRGBdx = dRGBdx + (kSH3[0] * sh[9] * 3.f * 2.f * xy + kSH3[1] * sh[10] * yz +
                                           kSH3[2] * sh[11] * -2.f * xy + kSH3[3] * sh[12] * -3.f * 2.f * xz +
                                           kSH3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                                           kSH3[5] * sh[14] * 2.f * xz + kSH3[6] * sh[15] * 3.f * (xx - yy));
                        dRGBdy = dRGBdy + (kSH3[0] * sh[9] * 3.f * (xx - yy) + kSH3[1] * sh[10] * xz +
                                           kSH3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) +
                                           kSH3[3] * sh[12] * -3.f * 2.f * yz + kSH3[4] * sh[13] * -2.f * xy +
                                           kSH3[5] * sh[14] * -2.f * yz + kSH3[6] * sh[15] * -3.f * 2.f * xy);
                        dRGBdz = dRGBdz + (kSH3[1] * sh[10] * xy + kSH3[2] * sh[11] * 4.f * 2.f * yz +
                                           kSH3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) +
                                           kSH3[4] * sh[13] * 4.f * 2.f * xz + kSH3[5] * sh[14] * (xx - yy));
                    }
                }
            }
            const V3 dL_ddir = {dot3(dRGBdx, g), dot3(dRGBdy, g), dot3(dRGBdz, g)};
            const V3 v = dir_orig;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            V3 dm;
            dm.x = ((+sum2 - v.x * v.x) * dL_ddir.x - v.y * v.x * dL_ddir.y - v.z * v.x * dL_ddir.z) * invsum32;
            dm.y = (-v.x * v.y * dL_ddir.x + (sum2 - v.y * v.y) * dL_ddir.y - v.z * v.y * dL_ddir.z) * invsum32;
            dm.z = (-v.x * v.z * dL_ddir.x - v.y * v.z * dL_ddir.y + (sum2 - v.z * v.z) * dL_ddir.z) * invsum32;
            gmean = gmean + dm;
            if constexpr (WITH_DSH) {
                // dL/dSH[k] = basis_k(dir) * g into the tile row, AFTER the last read of the coefficients it replaces
                const int ncoef = (a.D + 1) * (a.D + 1);
                for (int k = ncoef; k < a.M; k++) put(k, V3{0, 0, 0});
                put(0, kSH0 * g);
                if (a.D > 0) {
                    put(1, (-kSH1 * y) * g);
                    put(2, (kSH1 * z) * g);
                    put(3, (-kSH1 * x) * g);
                    if (a.D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        put(4, (kSH2[0] * xy) * g);
                        put(5, (kSH2[1] * yz) * g);
                        put(6, (kSH2[2] * (2.f * zz - xx - yy)) * g);
                        put(7, (kSH2[3] * xz) * g);
                        put(8, (kSH2[4] * (xx - yy)) * g);
                        if (a.D > 2) {
                            put(9, (kSH3[0] * y * (3.f * xx - yy)) * g);
                            put(10, (kSH3[1] * xy * z) * g);
                            put(11, (kSH3[2] * y * (4.f * zz - xx - yy)) * g);
                            put(12, (kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g);
                            put(13, (kSH3[4] * x * (4.f * zz - xx - yy)) * g);
                            put(14, (kSH3[5] * z * (xx - yy)) * g);
                            put(15, (kSH3[6] * x * (xx - 3.f * yy)) * g);
                        }
                    }
                }
            }
        }
        // ---- cov3D -> scale / rotation
        if (a.scales) {
            const float4 q = rot_in;
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const M3 R = rotation_from_quat(r, x, y, z);
            M3 S = make_m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
            const V3 s = a.scale_modifier * scale_in;
            S.c[0][0] = s.x;
            S.c[1][1] = s.y;
            S.c[2][2] = s.z;
            const M3 Mm = mul(S, R);
            const M3 dSig = make_m3(gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                                    0.5f * gcov[2], 0.5f * gcov[4], gcov[5]);
            M3 M2 = Mm;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) M2.c[i][j] = Mm.c[i][j] * 2.0f;
            const M3 dL_dM = mul(M2, dSig);
            const M3 Rt = transpose(R);
            M3 dMt = transpose(dL_dM);
            gscale.x = dot3(column(Rt, 0), column(dMt, 0));
            gscale.y = dot3(column(Rt, 1), column(dMt, 1));
            gscale.z = dot3(column(Rt, 2), column(dMt, 2));
#pragma unroll
            for (int j = 0; j < 3; j++) {
                dMt.c[0][j] *= s.x;
                dMt.c[1][j] *= s.y;
                dMt.c[2][j] *= s.z;
            }
            const auto& A = dMt.c;
            grot.x = 2 * z * (A[0][1] - A[1][0]) + 2 * y * (A[2][0] - A[0][2]) + 2 * x * (A[1][2] - A[2][1]);
            grot.y = 2 * y * (A[1][0] + A[0][1]) + 2 * z * (A[2][0] + A[0][2]) + 2 * r * (A[1][2] - A[2][1]) -
                     4 * x * (A[2][2] + A[1][1]);
            grot.z = 2 * x * (A[1][0] + A[0][1]) + 2 * r * (A[2][0] - A[0][2]) + 2 * z * (A[1][2] + A[2][1]) -
                     4 * y * (A[2][2] + A[0][0]);
            grot.w = 2 * r * (A[0][1] - A[1][0]) + 2 * x * (A[2][0] + A[0][2]) + 2 * y * (A[1][2] + A[2][1]) -
                     4 * z * (A[1][1] + A[0][0]);
        }
    }
    if (visible) {
        if (a.accumulate) {  // (requested together, added, written back)
            const float m0 = dL_dmean3D[3 * idx], m1 = dL_dmean3D[3 * idx + 1], m2 = dL_dmean3D[3 * idx + 2];
            float c6[6];
#pragma unroll
            for (int i = 0; i < 6; i++) c6[i] = dL_dcov3D[(size_t)6 * idx + i];
            const float s0 = dL_dscale[3 * idx], s1 = dL_dscale[3 * idx + 1], s2 = dL_dscale[3 * idx + 2];
            const float4 r4 = reinterpret_cast<const float4*>(dL_drot)[idx];
            gmean = gmean + V3{m0, m1, m2};
#pragma unroll
            for (int i = 0; i < 6; i++) gcov[i] += c6[i];
            gscale = gscale + V3{s0, s1, s2};
            grot = make_float4(grot.x + r4.x, grot.y + r4.y, grot.z + r4.z, grot.w + r4.w);
        }
        dL_dmean3D[3 * idx] = gmean.x;
        dL_dmean3D[3 * idx + 1] = gmean.y;
        dL_dmean3D[3 * idx + 2] = gmean.z;
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[(size_t)6 * idx + i] = gcov[i];
        dL_dscale[3 * idx] = gscale.x;
        dL_dscale[3 * idx + 1] = gscale.y;
        dL_dscale[3 * idx + 2] = gscale.z;
        reinterpret_cast<float4*>(dL_drot)[idx] = grot;
    }
    if constexpr (WITH_DSH || FROM_ROWS) __syncthreads();
    if constexpr (SRC == 1) {
        // dL/dsemantics of the chunk's rows: S floats from the Gaussian's record (zeros for a visible Gaussian without tiles).
        // Four lanes per row at S = 16 (one float4 each): an instruction moves 16 records' 64-byte pieces in and 16 rows out.
        if ((ra.S & 3) == 0) {
            const int S4 = ra.S >> 2;
            for (int i = threadIdx.x; i < nrows * S4; i += 256) {
                const int r = i / S4, sub = i - r * S4;
                const uint32_t o = s_src[r];
                const float4 v = o != 0xFFFFFFFFu ? *reinterpret_cast<const float4*>(ra.rows + (size_t)o * 4 * ra.row_floats + 4 * sub)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                float4* dst = reinterpret_cast<float4*>(ra.dL_dsemantic + (size_t)s_vis[r] * ra.S + 4 * sub);
                if (a.accumulate) {
                    const float4 p = *dst;
                    *dst = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
                } else {
                    *dst = v;
                }
            }
        } else {
            for (int i = threadIdx.x; i < nrows * ra.S; i += 256) {
                const int r = i / ra.S, ch = i - r * ra.S;
                const uint32_t o = s_src[r];
                const float v = o != 0xFFFFFFFFu ? ra.rows[(size_t)o * 4 * ra.row_floats + ch] : 0.f;
                float* dst = &ra.dL_dsemantic[(size_t)s_vis[r] * ra.S + ch];
                *dst = a.accumulate ? *dst + v : v;
            }
        }
    }
    if constexpr (WITH_DSH) {
        // the tile's rows -> the dL/dSH rows of their Gaussians: consecutive lanes take consecutive floats of a row (conflict-free
        // LDS reads, 192 contiguous bytes per row in memory); (row, column) advance without a division
        const int dr = 256 / w, dc = 256 - dr * w;
        int r = (int)threadIdx.x / w, col = (int)threadIdx.x - r * w;
        while (r < nrows) {
            float* dst = &dL_dsh[(size_t)s_vis[r] * w + col];
            *dst = a.accumulate ? *dst + s_dsh[r * (w + 1) + col] : s_dsh[r * (w + 1) + col];
            r += dr;
            col += dc;
            if (col >= w) {
                col -= w;
                r++;
            }
        }
    }
    // the ids that are still pending move to the front of the list (at most 255 of them: one per thread)
    __syncthreads();
    const int rest = npend - nrows;
    const uint32_t moved = (int)threadIdx.x < rest ? s_vis[nrows + threadIdx.x] : 0u;
    __syncthreads();
    if ((int)threadIdx.x < rest) s_vis[threadIdx.x] = moved;
    npend = rest;
    __syncthreads();
    }  // chunks of pending visible Gaussians
    }  // segments
}

// dL/dSH of a batch of views from its factors: dL/dSH[g][k] = sum_v basis_k(dir(g, v)) * gcol[v][g], with gcol the
// clamp-masked colour gradients that preprocess_bwd_k<false> leaves in dL_dcolor and dir the normalised direction
// camera v -> Gaussian g.  The basis expressions are those of the SH backward above, term by term, and the views are
// added in index order starting from view 0: the result is bit-identical to summing the per-view dL/dSH arrays in
// that order.  Used by the data-parallel gradient exchange (dist.py): 12 bytes per Gaussian and view travel instead
// of 192.  One thread per Gaussian, rows staged through LDS like preprocess_bwd_k.
__global__ __launch_bounds__(256) void sh_grad_from_views_k(int P, int D, int M, int V, const float* __restrict__ means3D,
                                                            const float* __restrict__ campos,  // [V,3]
                                                            const float* __restrict__ gcol,    // [V,P,3]
                                                            float* __restrict__ dL_dsh) {      // [P,M,3]
    extern __shared__ float s_dsh[];  // [256][3 M + 1]
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gtid < P ? gtid : P - 1;
    const V3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    V3 acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = V3{0, 0, 0};
    for (int v = 0; v < V; v++) {
        const V3 cp = {campos[3 * v], campos[3 * v + 1], campos[3 * v + 2]};
        const V3 dir_orig = mean - cp;
        const V3 dir = dir_orig / sqrtf(dot3(dir_orig, dir_orig));
        const float* gp = gcol + ((size_t)v * P + idx) * 3;
        const V3 g = {gp[0], gp[1], gp[2]};
        const float x = dir.x, y = dir.y, z = dir.z;
        V3 t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = V3{0, 0, 0};
        t[0] = kSH0 * g;
        if (D > 0) {
            t[1] = (-kSH1 * y) * g;
            t[2] = (kSH1 * z) * g;
            t[3] = (-kSH1 * x) * g;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                t[4] = (kSH2[0] * xy) * g;
                t[5] = (kSH2[1] * yz) * g;
                t[6] = (kSH2[2] * (2.f * zz - xx - yy)) * g;
                t[7] = (kSH2[3] * xz) * g;
                t[8] = (kSH2[4] * (xx - yy)) * g;
                if (D > 2) {
                    t[9] = (kSH3[0] * y * (3.f * xx - yy)) * g;
                    t[10] = (kSH3[1] * xy * z) * g;
                    t[11] = (kSH3[2] * y * (4.f * zz - xx - yy)) * g;
                    t[12] = (kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                    t[13] = (kSH3[4] * x * (4.f * zz - xx - yy)) * g;
                    t[14] = (kSH3[5] * z * (xx - yy)) * g;
                    t[15] = (kSH3[6] * x * (xx - 3.f * yy)) * g;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = v == 0 ? t[k] : acc[k] + t[k];
    }
    float* row = s_dsh + (size_t)threadIdx.x * (3 * M + 1);
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < M) {
            row[3 * k] = acc[k].x;
            row[3 * k + 1] = acc[k].y;
            row[3 * k + 2] = acc[k].z;
        }
    __syncthreads();
    const int w = 3 * M;
    const int rows = min(256, P - (int)(blockIdx.x * blockDim.x));
    float* out = dL_dsh + (size_t)blockIdx.x * blockDim.x * w;
    for (int i = threadIdx.x; i < rows * w; i += 256) {
        const int r = i / w, col = i - r * w;
        out[i] = s_dsh[r * (w + 1) + col];
    }
}

__global__ __launch_bounds__(256) void mark_visible_k(int P, const float* __restrict__ means3D, const float* view,
                                                      uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const V3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = view[i];
    present[idx] = xform_point_4x3(p, m).z > 0.2f ? 1 : 0;
}

}  // namespace

void launch_preprocess_fwd(const GoiRasterScene& sc, const GeomView& g, int* radii, uint2* ranges, int n_tiles,
                           hipStream_t s, const float* zcut, uint32_t* zlearn) {
    PreArgs a;
    a.zcut = g_options.cull_variant >= 2 ? zcut : nullptr;  // (the cut lives in the ellipse tile masks)
    a.zlearn = zlearn;
    a.P = sc.P; a.D = sc.D; a.M = sc.M; a.W = sc.W; a.H = sc.H;
    a.gx = (sc.W + TILE - 1) / TILE;
    a.gy = (sc.H + TILE - 1) / TILE;
    a.prefiltered = sc.prefiltered;
    a.cull = g_options.cull_variant;
    a.means3D = sc.means3D; a.shs = sc.shs; a.colors_precomp = sc.colors_precomp; a.opacities = sc.opacities;
    a.scales = sc.scales; a.rotations = sc.rotations; a.cov3D_precomp = sc.cov3D_precomp;
    a.scale_modifier = sc.scale_modifier; a.tan_fovx = sc.tan_fovx; a.tan_fovy = sc.tan_fovy;
    a.focal_y = sc.H / (2.0f * sc.tan_fovy);
    a.focal_x = sc.W / (2.0f * sc.tan_fovx);
    a.view_p = sc.viewmatrix; a.proj_p = sc.projmatrix; a.campos_p = sc.campos;
    static_assert(PRE_BLOCK == 256, "preprocess_fwd_k is written for 256-thread workgroups");
    // SH colours in the degree-3 layout: the wave moves its visible lanes' rows by LDS-DMA (pre_shdma 0: every lane fetches its own)
    if (sc.shs && !sc.colors_precomp && sc.M == 16 && g_options.pre_shdma)
        preprocess_fwd_k<true><<<dim3((sc.P + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, s>>>(
            a, g.rec, g.cov3D, g.tiles_touched, g.clamped, g.sort_keys[1], g.aux, g.blk_agg, g.blk_coarse, radii, g.counters, ranges, n_tiles);
    else
        preprocess_fwd_k<false><<<dim3((sc.P + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, s>>>(
            a, g.rec, g.cov3D, g.tiles_touched, g.clamped, g.sort_keys[1], g.aux, g.blk_agg, g.blk_coarse, radii, g.counters, ranges, n_tiles);
}

void launch_preprocess_bwd(const GoiRasterScene& sc, const GeomView& g, const int* radii, float* dL_dmean2D,
                           const float* dL_dconic, float* dL_dcolor, const float* dL_ddepth, float* dL_dmean3D,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, hipStream_t s,
                           const float* record_rows, float* dL_dopacity, float* dL_dsemantic, const int* prev_radii,
                           const uint8_t* row_flags, int N_cap, bool accumulate) {
    BwdArgs a;
    a.prev_radii = prev_radii;
    a.accumulate = accumulate ? 1 : 0;
    a.P = sc.P; a.D = sc.D; a.M = sc.M; a.W = sc.W; a.H = sc.H;
    a.means3D = sc.means3D; a.shs = sc.shs; a.scales = sc.scales; a.rotations = sc.rotations;
    a.cov3D = sc.cov3D_precomp ? sc.cov3D_precomp : g.cov3D;
    a.scale_modifier = sc.scale_modifier; a.tan_fovx = sc.tan_fovx; a.tan_fovy = sc.tan_fovy;
    a.focal_y = sc.H / (2.0f * sc.tan_fovy);
    a.focal_x = sc.W / (2.0f * sc.tan_fovx);
    a.view_p = sc.viewmatrix; a.proj_p = sc.projmatrix; a.campos_p = sc.campos;
    const bool with_sh = sc.shs && sc.M > 0 && dL_dsh;  // dL_dsh == NULL with SH colours: factored mode
    // record_rows: the blend gradients are the records reduce_rows_k<.., RECORD> left in the row scratch -- or, with row_flags
    // (bwd_records 2), the rows themselves: the kernel sums them (128-byte rows only: the caller checks)
    const bool fused = record_rows != nullptr && row_flags != nullptr;
    const int tile_rows = fused ? BWD_TILE_ROWS_FUSED : BWD_TILE_ROWS_DEFAULT;
    const size_t lds = with_sh ? (size_t)tile_rows * (3 * sc.M + 1) * sizeof(float) : 0;  // 43.9 KB at M = 16 (40.8 fused)
    RecArgs ra;
    ra.rows = record_rows; ra.aux = g.aux; ra.tiles_touched = g.tiles_touched;
    ra.row_floats = bwd_row_floats(sc.S); ra.S = sc.S; ra.nch = 4 * ((sc.S + 3) / 4) + 4;
    ra.dL_dopacity = dL_dopacity; ra.dL_dsemantic = dL_dsemantic;
    ra.flags = row_flags; ra.n_dev = g.counters + COUNTER_N; ra.N_cap = (uint32_t)std::max(N_cap, 0);
    // persistent workgroups: every CU gets as many as fit (registers and the staging tile: BWD_BLOCKS_PER_CU), each walks
    // segments of 256 ids
    static const int n_cu = []() {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const dim3 grid((unsigned)std::min((sc.P + 255) / 256, n_cu * BWD_BLOCKS_PER_CU));
#define GOI_PBWD(DSH, SRC, LDS)                                                                                             \
    preprocess_bwd_k<DSH, SRC><<<grid, dim3(256), LDS, s>>>(a, radii, g.counters, g.clamped, dL_dmean2D, dL_dconic, dL_dcolor, \
                                                            dL_ddepth, ra, dL_dmean3D, dL_dcov3D, DSH ? dL_dsh : nullptr,      \
                                                            dL_dscale, dL_drot)
    if (with_sh && fused) GOI_PBWD(true, 2, lds);
    else if (with_sh && record_rows) GOI_PBWD(true, 1, lds);
    else if (with_sh) GOI_PBWD(true, 0, lds);
    else if (fused) GOI_PBWD(false, 2, 0);
    else if (record_rows) GOI_PBWD(false, 1, 0);
    else GOI_PBWD(false, 0, 0);
#undef GOI_PBWD
}

void launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* gcol,
                               float* dL_dsh, hipStream_t s) {
    const size_t lds = (size_t)256 * (3 * M + 1) * sizeof(float);
    sh_grad_from_views_k<<<dim3((P + 255) / 256), dim3(256), lds, s>>>(P, D, M, V, means3D, campos, gcol, dL_dsh);
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    mark_visible_k<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, means3D, view, present);
}

}  // namespace goi
