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
#include <type_traits>

#include "common.h"
#include "gmath.h"

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

__global__ __launch_bounds__(256) void preprocess_fwd_k(const PreArgs args, GaussRec* __restrict__ rec,
                                                        float* __restrict__ cov3D_out,
                                                        uint32_t* __restrict__ tiles_touched,
                                                        uint8_t* __restrict__ clamped, uint32_t* __restrict__ raw_key,
                                                        uint4* __restrict__ aux,
                                                        uint2* __restrict__ blk_agg, int* __restrict__ radii,
                                                        uint32_t* __restrict__ counters, uint2* __restrict__ ranges,
                                                        int n_tiles) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    // the tile ranges start from zero (emit accumulates per-tile counts into them): cleared here for free
    for (int t = gtid; t < n_tiles; t += gridDim.x * blockDim.x) ranges[t] = make_uint2(0u, 0u);
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
        const float con_a = cz * det_inv, con_b = -cy * det_inv, con_c = cx * det_inv;
        const float mid = 0.5f * (cx + cz);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pix = ndc_to_pix(projx, a.W), piy = ndc_to_pix(projy, a.H);
        int x0, y0, x1, y1;
        tile_rect(pix, piy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;

        float cr, cg, cb;
        if (a.colors_precomp) {
            cr = a.colors_precomp[3 * idx];
            cg = a.colors_precomp[3 * idx + 1];
            cb = a.colors_precomp[3 * idx + 2];
        } else {
            const V3 campos = {a.campos[0], a.campos[1], a.campos[2]};
            V3 dir = p - campos;
            dir = dir / sqrtf(dot3(dir, dir));
            const V3* sh = reinterpret_cast<const V3*>(a.shs) + (size_t)idx * a.M;
            V3 res = kSH0 * sh[0];
            if (a.D > 0) {
                const float x = dir.x, y = dir.y, z = dir.z;
                res = res - kSH1 * y * sh[1] + kSH1 * z * sh[2] - kSH1 * x * sh[3];
                if (a.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    res = res + kSH2[0] * xy * sh[4] + kSH2[1] * yz * sh[5] + kSH2[2] * (2.0f * zz - xx - yy) * sh[6] +
                          kSH2[3] * xz * sh[7] + kSH2[4] * (xx - yy) * sh[8];
                    if (a.D > 2) {
                        res = res + kSH3[0] * y * (3.0f * xx - yy) * sh[9] + kSH3[1] * xy * z * sh[10] +
                              kSH3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                              kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                              kSH3[4] * x * (4.0f * zz - xx - yy) * sh[13] + kSH3[5] * z * (xx - yy) * sh[14] +
                              kSH3[6] * x * (xx - 3.0f * yy) * sh[15];
                    }
                }
            }
            res = res + V3{0.5f, 0.5f, 0.5f};
            clamp_bits = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
            cr = fmaxf(res.x, 0.0f);
            cg = fmaxf(res.y, 0.0f);
            cb = fmaxf(res.z, 0.0f);
        }
        // Box outside which alpha = min(0.99, o*exp(power)) < 1/255 is certain (see DESIGN.md,
        // "exact contribution box"): the ellipse power >= -tau of the COMPUTED conic, tau inflated.
        const float o = a.opacities[idx];
        float hx = -1.f, hy = -1.f;
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
        GaussRec r;
        r.q0 = make_float4(pix, piy, con_a, con_b);
        r.q1 = make_float4(con_c, o, hx, hy);       // what the hit test needs beside q0: fetched for every CANDIDATE
        r.q2 = make_float4(cr, cg, cb, p_view.z);   // what only a HIT needs (r, g, b, depth: the first staged feature quad as it is)
        rec[idx] = r;
        my_radius_i = (int)my_radius;
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
            tmask_v = mask;
            touched = (uint32_t)__popcll(mask);
        }
        key = __float_as_uint(p_view.z);
    } while (false);

    if (live) {
        radii[idx] = my_radius_i;
        tiles_touched[idx] = touched;
        clamped[idx] = clamp_bits;
        raw_key[idx] = key;  // depth bits by Gaussian id; compact_listed_k keeps the listed ones for the depth sort
        aux[idx] = make_uint4(0u, (uint32_t)my_radius_i, (uint32_t)tmask_v, (uint32_t)(tmask_v >> 32));  // (.x: emit)
    }
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
        blk_agg[blockIdx.x] = make_uint2(s_wvis[0] + s_wvis[1] + s_wvis[2] + s_wvis[3], bsum);
    }
}

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
constexpr int COMPACT_ROUNDS = 8;
__global__ __launch_bounds__(PRE_BLOCK) void compact_listed_k(int P, const uint32_t* __restrict__ tiles_touched,
                                                              const uint32_t* __restrict__ raw_key,
                                                              const uint2* __restrict__ blk_agg,
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
    // ---- base rank (blocks in front) and totals
    uint32_t before = 0, all_v = 0, all_t = 0;
    for (int i = tid; i < nblk; i += PRE_BLOCK) {
        const uint2 a = blk_agg[i];
        before += i < blk0 ? a.x : 0u;
        all_v += a.x;
        all_t += a.y;
    }
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
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t c = s_h[p][tid];
            if (c) atomicAdd(&ghist[p * 256 + tid], c);
        }
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
};
template <bool WITH_DSH, bool FROM_ROWS>
__global__ __launch_bounds__(256) void preprocess_bwd_k(const BwdArgs args, const int* __restrict__ radii,
                                                        const uint32_t* __restrict__ counters,
                                                        const uint8_t* __restrict__ clamped,
                                                        float* dL_dmean2D,
                                                        const float* __restrict__ dL_dconic,
                                                        float* dL_dcolor,
                                                        const float* __restrict__ dL_ddepth, const RecArgs ra,
                                                        float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
                                                        float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
                                                        float* __restrict__ dL_drot) {
    // dL/dSH is the widest output (192 B per Gaussian at degree 3).  Written per thread it is 48 stores with a
    // 192-byte lane stride; instead every thread fills its row of an LDS tile (odd row stride: no bank conflicts)
    // and the block streams the tile out as full lines.
    extern __shared__ float s_dsh[];  // [256][3 M + 1] when dL_dsh
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gtid < args.P;
    const int idx = live ? gtid : args.P - 1;  // lanes past P redo the last Gaussian and write nothing
    const Camera cam = load_camera(args.view_p, args.proj_p, args.campos_p);
    struct : BwdArgs {
        const float *view, *proj, *campos;
    } a;
    static_cast<BwdArgs&>(a) = args;
    a.view = cam.view;
    a.proj = cam.proj;
    a.campos = cam.campos;
    V3 gmean = {0, 0, 0};
    float gcov[6] = {0, 0, 0, 0, 0, 0};
    V3 gscale = {0, 0, 0};
    float4 grot = make_float4(0, 0, 0, 0);
    // (a truncated frame -- COUNTER_OVF -- is treated as if nothing were visible: all gradients zero)
    const bool visible = radii[idx] > 0 && counters[COUNTER_OVF] == 0;
    // ---- the blend gradients: from the per-id arrays, or from the Gaussian's record
    float in_conic[3] = {0.f, 0.f, 0.f}, in_m2d[2] = {0.f, 0.f}, in_depth = 0.f;
    V3 in_col = {0.f, 0.f, 0.f};
    if constexpr (FROM_ROWS) {
        const bool listed = visible && ra.tiles_touched[idx] != 0;
        const float* rec = ra.rows + (size_t)(listed ? ra.aux[idx].x : 0u) * 4 * ra.row_floats;
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
        if (live) {
            ra.dL_dopacity[idx] = opa;
            dL_dmean2D[3 * idx] = in_m2d[0];
            dL_dmean2D[3 * idx + 1] = in_m2d[1];
            dL_dmean2D[3 * idx + 2] = 0.f;
            dL_dcolor[3 * idx] = in_col.x;  // (factored SH mode overwrites it with the clamp-masked gradient below)
            dL_dcolor[3 * idx + 1] = in_col.y;
            dL_dcolor[3 * idx + 2] = in_col.z;
        }
        // dL/dsemantics, the widest of them (64 bytes per Gaussian at S = 16).  Per thread it is S / 4 loads and stores with a
        // 64-byte lane stride; with 4 lanes per Gaussian (one float4 each) an instruction moves 16 Gaussians: 16 x 64-byte
        // pieces of their records in, ONE contiguous kilobyte of the output out.
        if ((ra.S & 3) == 0 && ra.S <= 16) {
            const int lane = threadIdx.x & 63, sub = lane & 3, S4 = ra.S >> 2;
            const uint32_t my = listed ? ra.aux[idx].x : 0xFFFFFFFFu;
            const int wave_first = gtid - lane;  // the Gaussian of lane 0
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int src = (lane >> 2) + 16 * k;
                const uint32_t o = (uint32_t)__shfl((int)my, src, 64);
                const int id2 = wave_first + src;
                if (sub < S4 && id2 < args.P) {
                    const float4 v = o != 0xFFFFFFFFu ? *reinterpret_cast<const float4*>(ra.rows + (size_t)o * 4 * ra.row_floats + 4 * sub)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(ra.dL_dsemantic + (size_t)id2 * ra.S + 4 * sub) = v;
                }
            }
        } else if (live) {
            float* ds = ra.dL_dsemantic + (size_t)idx * ra.S;
            if ((ra.S & 3) == 0) {
                for (int ch = 0; ch < ra.S; ch += 4)
                    *reinterpret_cast<float4*>(ds + ch) = listed ? *reinterpret_cast<const float4*>(rec + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int ch = 0; ch < ra.S; ch++) ds[ch] = listed ? rec[ch] : 0.f;
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
    V3* dsh = WITH_DSH ? reinterpret_cast<V3*>(s_dsh + (size_t)threadIdx.x * (3 * a.M + 1)) : nullptr;
    auto put = [&](int k, const V3& v) {
        if constexpr (WITH_DSH) dsh[k] = v;
    };

    if (visible) {
        const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        // ---- conic -> cov2D -> cov3D and the covariance path of the mean gradient
        {
            float cov3D[6];
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = a.cov3D[(size_t)6 * idx + i];
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
            const V3* sh = reinterpret_cast<const V3*>(a.shs) + (size_t)idx * a.M;
            const uint8_t cl = clamped[idx];
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
            const int ncoef = (a.D + 1) * (a.D + 1);
            for (int k = ncoef; k < a.M; k++) put(k, V3{0, 0, 0});
            put(0, kSH0 * g);
            if (a.D > 0) {
                put(1, (-kSH1 * y) * g);
                put(2, (kSH1 * z) * g);
                put(3, (-kSH1 * x) * g);
                dRGBdx = -kSH1 * sh[3];
                dRGBdy = -kSH1 * sh[1];
                dRGBdz = kSH1 * sh[2];
                if (a.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    put(4, (kSH2[0] * xy) * g);
                    put(5, (kSH2[1] * yz) * g);
                    put(6, (kSH2[2] * (2.f * zz - xx - yy)) * g);
                    put(7, (kSH2[3] * xz) * g);
                    put(8, (kSH2[4] * (xx - yy)) * g);
                    dRGBdx = dRGBdx + (kSH2[0] * y * sh[4] + kSH2[2] * 2.f * -x * sh[6] + kSH2[3] * z * sh[7] +
                                       kSH2[4] * 2.f * x * sh[8]);
                    dRGBdy = dRGBdy + (kSH2[0] * x * sh[4] + kSH2[1] * z * sh[5] + kSH2[2] * 2.f * -y * sh[6] +
                                       kSH2[4] * 2.f * -y * sh[8]);
                    dRGBdz = dRGBdz + (kSH2[1] * y * sh[5] + kSH2[2] * 2.f * 2.f * z * sh[6] + kSH2[3] * x * sh[7]);
                    if (a.D > 2) {
                        put(9, (kSH3[0] * y * (3.f * xx - yy)) * g);
                        put(10, (kSH3[1] * xy * z) * g);
                        put(11, (kSH3[2] * y * (4.f * zz - xx - yy)) * g);
                        put(12, (kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g);
                        put(13, (kSH3[4] * x * (4.f * zz - xx - yy)) * g);
                        put(14, (kSH3[5] * z * (xx - yy)) * g);
                        put(15, (kSH3[6] * x * (xx - 3.f * yy)) * g);
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
        }
        // ---- cov3D -> scale / rotation
        if (a.scales) {
            const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const M3 R = rotation_from_quat(r, x, y, z);
            M3 S = make_m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
            const V3 s = a.scale_modifier * V3{a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
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
    } else if (WITH_DSH) {
        for (int k = 0; k < a.M; k++) put(k, V3{0, 0, 0});
    }
    if (live) {
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
    if constexpr (WITH_DSH) {
        __syncthreads();
        const int w = 3 * a.M;
        const int rows = min(256, a.P - (int)(blockIdx.x * blockDim.x));
        float* out = dL_dsh + (size_t)blockIdx.x * blockDim.x * w;
        for (int i = threadIdx.x; i < rows * w; i += 256) {
            const int row = i / w, col = i - row * w;
            out[i] = s_dsh[row * (w + 1) + col];
        }
    }
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
template <int K, int GPQ, bool RECORD>  // K = row_floats / 16; GPQ = Gaussians per quarter wave
__global__ __launch_bounds__(256) void reduce_rows_k(int P, int S, int nch, uint32_t N_cap, const uint32_t* __restrict__ n_dev,
                                                     const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ tiles_touched,
                                                     float* rows, const uint8_t* __restrict__ flags,
                                                     float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                                     float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
                                                     float* __restrict__ dL_dsemantic, float* __restrict__ dL_ddepth) {
    constexpr int RF = 16 * K;
#ifndef GOI_REDUCE_INFLIGHT
#define GOI_REDUCE_INFLIGHT 32
#endif
    constexpr int INFLIGHT = GOI_REDUCE_INFLIGHT;
    // N_cap: the slot capacity the scratch was laid out for; n_dev: the forward's instance count on the device (the
    // exact forward passes N_cap = num_rendered; the speculative one a capacity, and an overflowed frame stored only
    // the first N_cap instances)
    // a TRUNCATED frame (COUNTER_OVF, set by emit) has no valid rows: every Gaussian gets zeros
    const bool truncated = n_dev[COUNTER_OVF - COUNTER_N] != 0;
    const uint32_t N = truncated ? 0u : min(N_cap, *n_dev);
    const int V = truncated ? 0 : (int)n_dev[COUNTER_V - COUNTER_N];  // listed Gaussians: the only ones that own rows
    const int lane = threadIdx.x & 63, quarter = lane >> 4, e = lane & 15;
    const uint32_t* flags32 = reinterpret_cast<const uint32_t*>(flags);
    const int nsem = nch - 4;
    // ---- phase 0 (the first ceil(P/256) workgroups): zeros for the Gaussians that are NOT listed (culled, or a culled
    // rectangle without tiles; all of them for a truncated frame) -- one Gaussian per lane.  The listed ones are written
    // by phase 1 below, so every element of the six arrays is written exactly once.
    if constexpr (!RECORD) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < P && (truncated || tiles_touched[i] == 0)) {
            if ((S & 3) == 0) {
                float4* d4 = reinterpret_cast<float4*>(dL_dsemantic + (size_t)i * S);
                for (int ch = 0; ch < S / 4; ch++) d4[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int ch = 0; ch < S; ch++) dL_dsemantic[(size_t)i * S + ch] = 0.f;
            }
            if (dL_dopacity) {  // (NULL in the feature-gradient-only reduction)
                dL_dopacity[i] = 0.f;
                dL_ddepth[i] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) dL_dcolor[(size_t)i * 3 + k] = dL_dmean2D[(size_t)i * 3 + k] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) dL_dconic[(size_t)i * 4 + k] = 0.f;
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
    // the 4 quadrant bytes of instance off0 + c + e (first chunk of a Gaussian: c = 0)
    auto load_flags = [&](const Meta& m, uint32_t c) { return (c + e < m.off1 - m.off0) ? flags32[m.off0 + c + e] : 0u; };

    Meta cur = load_meta(0), nxt = load_meta(1);
    uint32_t w_cur = load_flags(cur, 0);
#pragma unroll 1
    for (int k = 0; k < GPQ; k++) {
        if (i0 + 16 * k - (int)(threadIdx.x >> 4) >= V) break;  // (block-uniform: nothing left for any quarter wave)
        const Meta nn = load_meta(k + 2);        // two Gaussians ahead: slot range
        const uint32_t w_nxt = load_flags(nxt, 0);  // one ahead: validity bytes of its first 16 instances
        const bool live = i0 + 16 * k < V;
        const uint32_t cnt = cur.off1 - cur.off0;
        const size_t inst0 = cur.off0;
        float sum[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) sum[kk] = 0.f;
        // every lane of the wave must reach the ballots: loop to the wave's largest count
        uint32_t cmax = cnt;
#pragma unroll
        for (int d = 32; d >= 16; d >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d, 64));
        uint32_t w_chunk = w_cur;
        for (uint32_t c = 0; c < cmax; c += 16) {
            const uint32_t w = w_chunk;                          // 4 quadrant bytes of instance c+e
            if (c + 16 < cmax) w_chunk = load_flags(cur, c + 16);  // (the next chunk's, under this chunk's rows)
            unsigned long long m = 0;  // bit 16q + i: quadrant q of instance c+i is valid
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long bal = __ballot(((w >> (8 * q)) & 0xFFu) != 0);
                m |= ((bal >> (16 * quarter)) & 0xFFFFull) << (16 * q);
            }
            const float* chunk = rows + (inst0 + c) * 4 * RF;
            // NF rows requested back to back, then added in slot order (absent slots add +0: the sums do not depend on
            // NF).  Most Gaussians own a handful of rows -- 6 on average, half of them at most 4 -- and the 16-slot trip
            // costs ~160 vector instructions whatever it finds (the kernel issued 60 M of them: 44 % VALU-busy on top
            // of its memory waits): when no quarter of the wave has more than 4 rows left, a 4-slot trip does.
            auto trip = [&](auto nf_c) {
                constexpr int NF = decltype(nf_c)::value;
                float v[NF][K];
#pragma unroll
                for (int i = 0; i < NF; i++) {
                    const bool have = m != 0;
                    const int bit = have ? __builtin_ctzll(m) : 0;
                    if (have) m &= m - 1;
                    const float* r = chunk + (size_t)((bit & 15) * 4 + (bit >> 4)) * RF;
                    if (K == 2) {  // one 8-byte load per lane: the quarter wave reads the 128-byte row in one request
                        const float2 t = have ? reinterpret_cast<const float2*>(r)[e] : make_float2(0.f, 0.f);
                        v[i][0] = t.x;
                        v[i][K - 1] = t.y;
                    } else {
#pragma unroll
                        for (int kk = 0; kk < K; kk++) v[i][kk] = have ? r[e + 16 * kk] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < NF; i++)
#pragma unroll
                    for (int kk = 0; kk < K; kk++) sum[kk] += v[i][kk];
            };
            int left = __popcll(m);  // rows this quarter still has to fetch; the wave's largest decides the trip
#pragma unroll
            for (int d = 32; d >= 16; d >>= 1) left = max(left, __shfl_xor(left, d, 64));
            left = __builtin_amdgcn_readfirstlane(left);
            while (left > 0) {
                if (left <= 4) {
                    trip(std::integral_constant<int, 4>{});
                    left -= 4;
                } else if (left <= 12) {
                    trip(std::integral_constant<int, 12>{});
                    left -= 12;
                } else {
                    trip(std::integral_constant<int, INFLIGHT>{});
                    left -= INFLIGHT;
                }
            }
        }
        if constexpr (RECORD) {
            if (live && cnt > 0) {  // (the row elements this lane summed, back where it read them)
                float* dst = rows + inst0 * 4 * RF;
                if (K == 2) {
                    reinterpret_cast<float2*>(dst)[e] = make_float2(sum[0], sum[K - 1]);
                } else {
#pragma unroll
                    for (int kk = 0; kk < K; kk++) dst[e + 16 * kk] = sum[kk];
                }
            }
        } else if (live) {
            const uint32_t g = cur.g;
#pragma unroll
            for (int kk = 0; kk < K; kk++) {
                const float v = sum[kk];
                const int el = K == 2 ? 2 * e + kk : e + 16 * kk;  // element of the row this lane summed
                if (el < nsem) {
                    if (el < S) dL_dsemantic[(size_t)g * S + el] = v;
                } else if (el < nsem + 3) {
                    dL_dcolor[(size_t)g * 3 + (el - nsem)] = v;
                } else if (el == nsem + 3) {
                    dL_ddepth[g] = v;
                } else if (el < nch + 2) {
                    dL_dmean2D[(size_t)g * 3 + (el - nch)] = v;
                    if (el == nch + 1) dL_dmean2D[(size_t)g * 3 + 2] = 0.f;
                } else if (el < nch + 5) {
                    const int c = el - nch - 2;  // a, b, c -> x, y, w of the [P,2,2] conic gradient
                    dL_dconic[(size_t)g * 4 + (c == 2 ? 3 : c)] = v;
                    if (c == 2) dL_dconic[(size_t)g * 4 + 2] = 0.f;
                } else if (el == nch + 5) {
                    dL_dopacity[g] = v;
                }
            }
        }
        cur = nxt;
        nxt = nn;
        w_cur = w_nxt;
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
constexpr int EMIT_ROUNDS = 4;

template <bool COUNT>
__global__ __launch_bounds__(256) void emit_k(int P, int gx, int gy, const GaussRec* __restrict__ rec,
                                              const int* __restrict__ radii, const uint32_t* __restrict__ order,
                                              const uint32_t* __restrict__ offsets, uint4* __restrict__ aux,
                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                              uint32_t* __restrict__ tile_count, uint32_t* __restrict__ counters,
                                              uint32_t* __restrict__ clear, uint32_t clear_words, uint32_t cap) {
    // cap: number of instances keys[] / vals[] can hold.  The exact forward sizes them for num_rendered, so the
    // guard below never fires; the speculative forward sizes them from a guess, and a frame that overflows must
    // stay memory-safe and self-consistent (the tile counts only count what was stored) until the host notices.
    const bool cull = counters[COUNTER_CULL] != 0;
    P = min(P, (int)counters[COUNTER_V]);  // order[] / offsets[] hold the LISTED Gaussians only (front of the depth order)
    // the frame's "truncated" flag (COUNTER_OVF): emit is the first kernel that knows both the count and the capacity
    // (... or was depth-sorted wrongly because a look-back of the sort timed out: COUNTER_SORTERR)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counters[COUNTER_OVF] = (counters[COUNTER_N] > cap ? 1u : 0u) | (counters[COUNTER_SORTERR] ? 2u : 0u);
    // COUNT: the blocks also zero the control words of the tile sort that follows (its own memset launch otherwise)
    if (COUNT)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < clear_words; i += gridDim.x * 256u) clear[i] = 0u;
    constexpr int ROUNDS_PER_BLOCK = COUNT ? EMIT_ROUNDS : 1;
    if ((int)blockIdx.x * ROUNDS_PER_BLOCK * 256 >= P) return;  // (block-uniform) nothing listed left for this block
    extern __shared__ uint32_t s_cnt[];  // [gx * gy] when COUNT
    __shared__ unsigned long long s_mask[4][64];  // the rectangles' tile masks (cull_variant 2)
    __shared__ unsigned long long s_mark[4];  // per wave and trip: bit p = some rectangle's last instance is at position p
    __shared__ uint4 s_info[4][64];  // (x0 | y0 << 16, exclusive count, offsets[] - exclusive count, Gaussian id)
    __shared__ int s_w[4][64];       // rectangle width in tiles
    const int T = gx * gy;
    if (COUNT) {
        for (int t = threadIdx.x; t < T; t += 256) s_cnt[t] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the counting variant amortises zeroing and flushing its tile histogram over EMIT_ROUNDS x 256 Gaussians
    constexpr int ROUNDS = COUNT ? EMIT_ROUNDS : 1;
    // a round's Gaussian: id -> radius, position and box are dependent gathers (two DRAM round trips); the next round's
    // are requested before this round's instances are written, or every round would start with both exposed (emit is a
    // small kernel: two waves per SIMD have nothing to hide them behind)
    struct Fetched {
        uint32_t g, off;
        int r;
        float4 q0, q2;
        unsigned long long mask;
    };
    auto fetch = [&](int rnd) {
        Fetched f{0u, 0u, 0, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, -1.f, -1.f), TMASK_FULL};
        const int i = (blockIdx.x * ROUNDS + rnd) * 256 + threadIdx.x;
        if (rnd < ROUNDS && i < P) {
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
    Fetched nxt = fetch(0);
#pragma unroll 1
    for (int rnd = 0; rnd < ROUNDS; rnd++) {
        const int i = (blockIdx.x * ROUNDS + rnd) * 256 + threadIdx.x;
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
    if (COUNT) {
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += 256) {
            const uint32_t c = s_cnt[t];
            if (c) atomicAdd(&tile_count[2 * t], c);
        }
    }
}

// One workgroup: per-tile counts (in ranges[t].y) -> ranges[t] = [start, end) ((0,0) for an empty tile, as
// the reference leaves it) and the global digit histograms of the tile sort's passes.  A thread owns IT = ceil(T / 1024)
// consecutive tiles (IT <= 12: emit only counts grids of at most 12288 tiles), so the prefix sum is ONE block scan
// (chunks of 1024 tiles with three barriers each took 14 us at 6600 tiles: pure latency).
constexpr int TRH_MAX_IT = 12;
__global__ __launch_bounds__(1024) void tile_ranges_hist_k(int T, uint2* __restrict__ ranges, int passes, int shift0,
                                                           int nbits0, int shift1, int nbits1,
                                                           uint32_t* __restrict__ ghist) {
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
        c[k] = (k < IT && t0 + k < T) ? ranges[t0 + k].y : 0u;
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

void launch_preprocess_fwd(const GoiRasterScene& sc, const GeomView& g, int* radii, uint2* ranges, int n_tiles,
                           hipStream_t s) {
    PreArgs a;
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
    preprocess_fwd_k<<<dim3((sc.P + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, s>>>(
        a, g.rec, g.cov3D, g.tiles_touched, g.clamped, g.sort_keys[1], g.aux, g.blk_agg, radii, g.counters, ranges, n_tiles);
}

void launch_compact_listed(int P, const GeomView& g, uint32_t* ghist, bool pad, hipStream_t s) {
    const int nblk = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    compact_listed_k<<<dim3((nblk + COMPACT_ROUNDS - 1) / COMPACT_ROUNDS), dim3(PRE_BLOCK), 0, s>>>(
        P, g.tiles_touched, g.sort_keys[1], g.blk_agg, g.counters, g.sort_keys[0], g.sort_vals[0], ghist, pad ? 1 : 0);
}

void launch_preprocess_bwd(const GoiRasterScene& sc, const GeomView& g, const int* radii, float* dL_dmean2D,
                           const float* dL_dconic, float* dL_dcolor, const float* dL_ddepth, float* dL_dmean3D,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, hipStream_t s,
                           const float* record_rows, float* dL_dopacity, float* dL_dsemantic) {
    BwdArgs a;
    a.P = sc.P; a.D = sc.D; a.M = sc.M; a.W = sc.W; a.H = sc.H;
    a.means3D = sc.means3D; a.shs = sc.shs; a.scales = sc.scales; a.rotations = sc.rotations;
    a.cov3D = sc.cov3D_precomp ? sc.cov3D_precomp : g.cov3D;
    a.scale_modifier = sc.scale_modifier; a.tan_fovx = sc.tan_fovx; a.tan_fovy = sc.tan_fovy;
    a.focal_y = sc.H / (2.0f * sc.tan_fovy);
    a.focal_x = sc.W / (2.0f * sc.tan_fovx);
    a.view_p = sc.viewmatrix; a.proj_p = sc.projmatrix; a.campos_p = sc.campos;
    const bool with_sh = sc.shs && sc.M > 0 && dL_dsh;  // dL_dsh == NULL with SH colours: factored mode
    const size_t lds = with_sh ? (size_t)256 * (3 * sc.M + 1) * sizeof(float) : 0;  // 50 KB at M = 16
    // record_rows: the blend gradients are the records reduce_rows_k<.., RECORD> left in the row scratch
    RecArgs ra;
    ra.rows = record_rows; ra.aux = g.aux; ra.tiles_touched = g.tiles_touched;
    ra.row_floats = bwd_row_floats(sc.S); ra.S = sc.S; ra.nch = 4 * ((sc.S + 3) / 4) + 4;
    ra.dL_dopacity = dL_dopacity; ra.dL_dsemantic = dL_dsemantic;
    const dim3 grid((sc.P + 255) / 256);
    if (with_sh && record_rows)
        preprocess_bwd_k<true, true><<<grid, dim3(256), lds, s>>>(
            a, radii, g.counters, g.clamped, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, ra, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    else if (with_sh)
        preprocess_bwd_k<true, false><<<grid, dim3(256), lds, s>>>(
            a, radii, g.counters, g.clamped, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, ra, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    else if (record_rows)
        preprocess_bwd_k<false, true><<<grid, dim3(256), 0, s>>>(
            a, radii, g.counters, g.clamped, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, ra, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot);
    else
        preprocess_bwd_k<false, false><<<grid, dim3(256), 0, s>>>(
            a, radii, g.counters, g.clamped, dL_dmean2D, dL_dconic, dL_dcolor, dL_ddepth, ra, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot);
}

void launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* gcol,
                               float* dL_dsh, hipStream_t s) {
    const size_t lds = (size_t)256 * (3 * M + 1) * sizeof(float);
    sh_grad_from_views_k<<<dim3((P + 255) / 256), dim3(256), lds, s>>>(P, D, M, V, means3D, campos, gcol, dL_dsh);
}

#ifndef GOI_REDUCE_GPQ
#define GOI_REDUCE_GPQ 2
#endif
constexpr int REDUCE_GPQ = GOI_REDUCE_GPQ;  // Gaussians per quarter wave of reduce_rows_k

// records: the sums stay in the row scratch as per-Gaussian records (see reduce_rows_k); the six arrays are not written
void launch_reduce_rows(const GoiRasterScene& sc, const GeomView& g, int N, const BwdScratchView& scr, float* dL_dmean2D,
                        float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic, float* dL_ddepth,
                                hipStream_t s, bool records) {
    const int rf = bwd_row_floats(sc.S), nch = 4 * ((sc.S + 3) / 4) + 4;
    const dim3 grid((sc.P + 16 * REDUCE_GPQ - 1) / (16 * REDUCE_GPQ));
    const uint32_t* order = g.sort_vals[depth_sort_result_index()];
    if (records) {
        if (rf == 32)
            reduce_rows_k<2, REDUCE_GPQ, true><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        else if (rf == 16)
            reduce_rows_k<1, REDUCE_GPQ, true><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        else
            reduce_rows_k<3, REDUCE_GPQ, true><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        return;
    }
    if (rf == 32)
        reduce_rows_k<2, REDUCE_GPQ, false><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                    dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepth);
    else if (rf == 16)
        reduce_rows_k<1, REDUCE_GPQ, false><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                    dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepth);
    else
        reduce_rows_k<3, REDUCE_GPQ, false><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, scr.rows, scr.flags,
                                                    dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dsemantic, dL_ddepth);
}

void launch_reduce_sem_rows(const GoiRasterScene& sc, const GeomView& g, int N, const float* rows, const uint8_t* flags,
                            int row_floats, float* dL_dsemantic, hipStream_t s) {
    // rows hold semantic channels only: with nch = row_floats + 4 every element index is a semantic one
    const int nch = row_floats + 4;
    const dim3 grid((sc.P + 16 * REDUCE_GPQ - 1) / (16 * REDUCE_GPQ));
    const uint32_t* order = g.sort_vals[depth_sort_result_index()];
    if (row_floats == 16)
        reduce_rows_k<1, REDUCE_GPQ, false><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, const_cast<float*>(rows), flags, nullptr,
                                                    nullptr, nullptr, nullptr, dL_dsemantic, nullptr);
    else
        reduce_rows_k<2, REDUCE_GPQ, false><<<grid, dim3(256), 0, s>>>(sc.P, sc.S, nch, (uint32_t)N, g.counters + COUNTER_N, order, g.offsets, g.tiles_touched, const_cast<float*>(rows), flags, nullptr,
                                                    nullptr, nullptr, nullptr, dL_dsemantic, nullptr);
}

void launch_emit(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                 uint32_t* vals, uint32_t cap, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    emit_k<false><<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, gx, gy, g.rec, radii, order, g.offsets, g.aux, keys, vals,
                                                              nullptr, g.counters, nullptr, 0u, cap);
}

bool emit_can_count_tiles(int W, int H) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    // (tile_ranges_hist_k: at most TRH_MAX_IT x 1024 tiles; the per-tile LDS counters of emit: 48 KB)
    return (size_t)gx * gy <= (size_t)TRH_MAX_IT * 1024 && (size_t)gx * gy * sizeof(uint32_t) <= 48 * 1024 &&
           tile_key_bits((uint32_t)(gx * gy)) <= 16;
}

// emit + per-tile counts; ranges must be zeroed by the caller's stream order (done here)
void launch_emit_counting(int P, int W, int H, const GeomView& g, const uint32_t* order, const int* radii, uint32_t* keys,
                          uint32_t* vals, uint2* ranges, uint32_t* clear, size_t clear_words, uint32_t cap, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    // `ranges` was zeroed by preprocess_fwd_k
    emit_k<true><<<dim3((P + 256 * EMIT_ROUNDS - 1) / (256 * EMIT_ROUNDS)), dim3(256), (size_t)gx * gy * sizeof(uint32_t), s>>>(
        P, gx, gy, g.rec, radii, order, g.offsets, g.aux, keys, vals, reinterpret_cast<uint32_t*>(ranges) + 1, g.counters,
        clear, (uint32_t)clear_words, cap);
}

// per-tile counts -> ranges and the two digit histograms (written to ghist[0..511]) of a sort on
// key bits [0, bits) split as radix_sort_pairs splits them
void launch_tile_ranges_hist(int W, int H, uint2* ranges, uint32_t* ghist, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int bits = tile_key_bits((uint32_t)(gx * gy));
    const int passes = (bits + 7) / 8;
    const int n0 = (bits + passes - 1) / passes, n1 = bits - n0;
    tile_ranges_hist_k<<<dim3(1), dim3(1024), 0, s>>>(gx * gy, ranges, passes, 0, n0, n0, n1 > 0 ? n1 : 1, ghist);
}

void launch_ranges(int N, const uint32_t* n_dev, const uint32_t* sorted_keys, uint2* ranges, int T, hipStream_t s) {
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, s);
    if (N > 0) ranges_k<<<dim3((N + 255) / 256), dim3(256), 0, s>>>(N, n_dev, sorted_keys, ranges);
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    mark_visible_k<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, means3D, view, present);
}

}  // namespace goi
