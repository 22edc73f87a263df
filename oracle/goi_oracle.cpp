/*
 * oracle/goi_oracle.cpp -- CPU oracle for the Gaussian rasterizer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is a restatement in plain C++17 (no CUDA, no glm) of the
 * arithmetic of the reference's rasterizer; every function cites the reference lines it follows
 * (paths relative to /root/reference/submodules/diff-gaussian-rasterization/, "CR/" =
 * cuda_rasterizer/).  Nothing here is shipped: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg load liboracle.so.
 *
 * PARITY STATUS -- "parity partially pinned":
 *   The reference is CUDA-only (nvcc + cub + cooperative_groups) and has no tests, golden
 *   vectors or CPU path, so it cannot be built or run in this image without writing stand-ins
 *   for the CUDA headers/libraries, which the project rules forbid.  What IS pinned against the
 *   reference's own (importable, Python) code by tests/golden/make_golden.py:
 *     - SH -> RGB evaluation            vs utils/sh_utils.py:eval_sh (+0.5, clamp_min 0)
 *     - scale/rotation -> cov3D         vs utils/general_utils.py:build_scaling_rotation
 *     - view / projection matrices      vs utils/graphics_utils.py
 *   What is NOT pinned by reference outputs (restated from the CUDA source, cross-checked only
 *   by an independent dense PyTorch restatement, autograd and finite differences in tests/):
 *     - EWA cov2D, radius/tile rectangle, key sort order, alpha blending, the whole backward.
 *
 * Differences from the reference that are deliberate and documented:
 *   - Quantities the reference accumulates with float atomicAdd in nondeterministic order
 *     (CR/backward.cu:565-621) are accumulated here in double and rounded once; this is the
 *     value every float summation order approximates.
 *   - The per-tile 256-thread batch structure (CR/forward.cu:312-372) is replaced by a plain
 *     per-pixel walk over the tile's sorted list; per-pixel results are identical because the
 *     block-wide early exit only fires when every pixel is already done.
 *   - trace (CR/forward.cu:521-526) is racy in the reference; here it is a deterministic sum
 *     (gau_sem) and count (num_gsem += S per hit, as the reference's inner loop does).
 *
 * Where the reference's OWN fp32 arithmetic is the noise (round 4).  The pair exponent
 *     power = -0.5f * (a dx dx + c dy dy) - b dx dy                      (CR/forward.cu:341, CR/backward.cu:527)
 *   is a difference of terms that, for a needle-shaped Gaussian hundreds of pixels long, are 1e3..1e6 times the
 *   result: its fp32 value depends on the compiler's contraction choices by up to several per cent of alpha (this
 *   file's plain and FMA-contracted builds -- two legal compilations of the reference's statement; nvcc contracts by
 *   default -- differ by 0.04 in colour on 3 % of the pixels of scene.make_clustered_scene).  Two instruments:
 *     - the forward's per-pixel flag byte gets a second bit: bit 0 = a guard within fragile_eps of flipping (as
 *       before), bit 1 = ILL-CONDITIONED: the accumulated first-order effect of the exponents' fp32 rounding bounds
 *       (2^-22 x the sum of the terms' magnitudes per pair, weighted by the pair's alpha T) exceeds 2.5e-5, a quarter
 *       of the forward tolerance.  `flag == 0` keeps meaning "two correct fp32 implementations agree here".
 *     - the build -DGOI_ORACLE_POWER_F64 (liboracle_f64p.so, variant "f64power") evaluates that one statement in
 *       double from the same fp32 operands and rounds once: the exact value of the reference's formula on the
 *       reference's inputs, which is what an ill-conditioned pixel is compared with instead.
 */
#include "goi_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
// the pair exponent of CR/forward.cu:341 / CR/backward.cu:527 / CR/forward.cu:508 (see the header: GOI_ORACLE_POWER_F64)
static inline float pair_power(const float* co, float dx, float dy) {
#ifdef GOI_ORACLE_POWER_F64
    return (float)(-0.5 * ((double)co[0] * dx * dx + (double)co[2] * dy * dy) - (double)co[1] * dx * dy);
#else
    return -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
#endif
}
// bound on the fp32 rounding error of that statement: a handful of roundings of its largest terms
static inline float pair_power_noise(const float* co, float dx, float dy) {
    return 2.4e-7f * (0.5f * std::fabs(co[0]) * dx * dx + 0.5f * std::fabs(co[2]) * dy * dy + std::fabs(co[1] * dx * dy));
}

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int BLOCK_X = 16;  // CR/config.h:16
constexpr int BLOCK_Y = 16;  // CR/config.h:17

// CR/auxiliary.h:21-39
constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                           -1.0925484305920792f, 0.5462742152960396f};
constexpr float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                           -0.5900435899266435f};

struct V3 {
    float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm::dot for vec3 sums the component products left to right.
inline float dot(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}

// Column-major 3x3 with glm::mat3 semantics: m.c[col][row]; the 9-scalar constructor fills
// columns (third_party/glm/glm/detail/type_mat3x3.inl).
struct M3 {
    float c[3][3];
};
inline M3 m3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
    M3 m;
    m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
    m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
    m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
    return m;
}
inline M3 transpose(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.c[i][j] = a.c[j][i];
    return r;
}
// glm operator*(mat3, mat3): Result[col][row] = A[0][row]*B[col][0] + A[1][row]*B[col][1] + A[2][row]*B[col][2]
inline M3 mul(const M3& a, const M3& b) {
    M3 r;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++)
            r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
    return r;
}
inline M3 scale(float s, const M3& a) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.c[i][j] = a.c[i][j] * s;
    return r;
}
inline V3 col(const M3& a, int i) { return {a.c[i][0], a.c[i][1], a.c[i][2]}; }

// CR/auxiliary.h:41-44 -- evaluated in double because of the double literals.
inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// CR/auxiliary.h:46-56
inline void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t& minx, uint32_t& miny,
                    uint32_t& maxx, uint32_t& maxy) {
    minx = (uint32_t)std::min(gx, std::max(0, (int)((px - max_radius) / BLOCK_X)));
    miny = (uint32_t)std::min(gy, std::max(0, (int)((py - max_radius) / BLOCK_Y)));
    maxx = (uint32_t)std::min(gx, std::max(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    maxy = (uint32_t)std::min(gy, std::max(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// CR/auxiliary.h:58-77
inline V3 transformPoint4x3(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
struct V4 {
    float x, y, z, w;
};
inline V4 transformPoint4x4(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
// CR/auxiliary.h:89-97
inline V3 transformVec4x3Transpose(V3 p, const float* m) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// CR/auxiliary.h:107-117
inline V3 dnormvdv(V3 v, V3 dv) {
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / std::sqrt(sum2 * sum2 * sum2);
    V3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

// CR/auxiliary.h:139-164 (the prefiltered trap is reported through the return code instead)
inline bool in_frustum(int idx, const float* pts, const float* view, V3& p_view) {
    V3 p = {pts[3 * idx], pts[3 * idx + 1], pts[3 * idx + 2]};
    p_view = transformPoint4x3(p, view);
    return !(p_view.z <= 0.2f);
}

// CR/forward.cu:118-152
inline void computeCov3D(V3 scale_, float mod, const float* rot, float* cov3D) {
    M3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale_.x;
    S.c[1][1] = mod * scale_.y;
    S.c[2][2] = mod * scale_.z;
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];  // not normalised (CR/forward.cu:127)
    M3 R = m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
              2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
              2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    M3 M = mul(S, R);
    M3 Sigma = mul(transpose(M), M);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}

struct Cov2DCtx {
    M3 T, Vrk, W;
    V3 t;           // clamped view-space mean
    float txtz, tytz, limx, limy;
};
// CR/forward.cu:74-113 and CR/backward.cu:166-199 (same forward recomputation)
inline void cov2D_common(V3 mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* cov3D,
                         const float* view, Cov2DCtx& c, M3& cov) {
    V3 t = transformPoint4x3(mean, view);
    c.limx = 1.3f * tan_fovx;
    c.limy = 1.3f * tan_fovy;
    c.txtz = t.x / t.z;
    c.tytz = t.y / t.z;
    t.x = std::min(c.limx, std::max(-c.limx, c.txtz)) * t.z;
    t.y = std::min(c.limy, std::max(-c.limy, c.tytz)) * t.z;
    c.t = t;
    M3 J = m3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    c.W = m3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c.T = mul(c.W, J);
    c.Vrk = m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    cov = mul(mul(transpose(c.T), transpose(c.Vrk)), c.T);
}

// CR/forward.cu:20-71
inline V3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, V3 campos, const float* shs,
                             uint8_t* clamped) {
    V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    V3 dir = pos - campos;
    dir = dir / std::sqrt(dot(dir, dir));
    const V3* sh = reinterpret_cast<const V3*>(shs) + (size_t)idx * max_coeffs;
    V3 result = SH_C0 * sh[0];
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] +
                     SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] + SH_C2[3] * xz * sh[7] + SH_C2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    result = result + V3{0.5f, 0.5f, 0.5f};
    clamped[3 * idx + 0] = (result.x < 0);
    clamped[3 * idx + 1] = (result.y < 0);
    clamped[3 * idx + 2] = (result.z < 0);
    return {std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f)};
}

// CR/rasterizer_impl.cu:35-50
inline uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb)
            msb += step;
        else
            msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

inline uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

}  // namespace

struct GoiOracleState {
    int P = 0, N = 0, W = 0, H = 0, gx = 0, gy = 0;
    // GeometryState (CR/rasterizer_impl.h:29-45)
    std::vector<float> depths, means2D, cov3D, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    // BinningState (CR/rasterizer_impl.h:56-65)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> point_list;
    // ImageState (CR/rasterizer_impl.h:47-54)
    std::vector<uint32_t> ranges;  // [T][2]
    std::vector<uint32_t> n_contrib;
};

extern "C" {

GoiOracleState* goi_oracle_state_new(void) { return new GoiOracleState(); }
void goi_oracle_state_free(GoiOracleState* s) { delete s; }

static int threads_or_default(int n) {
#ifdef _OPENMP
    return n > 0 ? n : omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

// CR/forward.cu:155-256 (preprocessCUDA), CR/rasterizer_impl.cu:281-322 (scan, duplicate, sort, ranges)
static int oracle_geometry_and_binning(const GoiOracleScene* sc, GoiOracleState* st, int* radii_out, int nt) {
    const int P = sc->P, W = sc->W, H = sc->H;
    st->P = P;
    st->W = W;
    st->H = H;
    st->gx = (W + BLOCK_X - 1) / BLOCK_X;
    st->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int gx = st->gx, gy = st->gy;
    st->depths.assign(P, 0.f);
    st->means2D.assign((size_t)2 * P, 0.f);
    st->cov3D.assign((size_t)6 * P, 0.f);
    st->conic_opacity.assign((size_t)4 * P, 0.f);
    st->rgb.assign((size_t)3 * P, 0.f);
    st->clamped.assign((size_t)3 * P, 0);
    st->radii.assign(P, 0);
    st->tiles_touched.assign(P, 0);
    st->point_offsets.assign(P, 0);

    // CR/rasterizer_impl.cu:226-227
    const float focal_y = H / (2.0f * sc->tan_fovy);
    const float focal_x = W / (2.0f * sc->tan_fovx);
    const V3 campos = {sc->campos[0], sc->campos[1], sc->campos[2]};
    int bad_prefilter = 0;

#pragma omp parallel for num_threads(nt) schedule(static)
    for (int idx = 0; idx < P; idx++) {
        V3 p_view;
        if (!in_frustum(idx, sc->means3D, sc->viewmatrix, p_view)) {
            if (sc->prefiltered) bad_prefilter = 1;
            continue;
        }
        V3 p_orig = {sc->means3D[3 * idx], sc->means3D[3 * idx + 1], sc->means3D[3 * idx + 2]};
        V4 p_hom = transformPoint4x4(p_orig, sc->projmatrix);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        V3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

        const float* cov3D;
        if (sc->cov3D_precomp != nullptr) {
            cov3D = sc->cov3D_precomp + (size_t)idx * 6;
        } else {
            V3 s = {sc->scales[3 * idx], sc->scales[3 * idx + 1], sc->scales[3 * idx + 2]};
            computeCov3D(s, sc->scale_modifier, sc->rotations + (size_t)4 * idx, st->cov3D.data() + (size_t)idx * 6);
            cov3D = st->cov3D.data() + (size_t)idx * 6;
        }
        Cov2DCtx ctx;
        M3 covm;
        cov2D_common(p_orig, focal_x, focal_y, sc->tan_fovx, sc->tan_fovy, cov3D, sc->viewmatrix, ctx, covm);
        covm.c[0][0] += 0.3f;
        covm.c[1][1] += 0.3f;
        V3 cov = {covm.c[0][0], covm.c[0][1], covm.c[1][1]};

        float det = (cov.x * cov.z - cov.y * cov.y);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        V3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};

        float mid = 0.5f * (cov.x + cov.z);
        float lambda1 = mid + std::sqrt(std::max(0.1f, mid * mid - det));
        float lambda2 = mid - std::sqrt(std::max(0.1f, mid * mid - det));
        float my_radius = std::ceil(3.f * std::sqrt(std::max(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
        uint32_t minx, miny, maxx, maxy;
        getRect(pix, piy, (int)my_radius, gx, gy, minx, miny, maxx, maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;

        if (sc->colors_precomp == nullptr) {
            V3 c = computeColorFromSH(idx, sc->D, sc->M, sc->means3D, campos, sc->shs, st->clamped.data());
            st->rgb[3 * idx + 0] = c.x;
            st->rgb[3 * idx + 1] = c.y;
            st->rgb[3 * idx + 2] = c.z;
        }
        st->depths[idx] = p_view.z;
        st->radii[idx] = (int)my_radius;
        st->means2D[2 * idx] = pix;
        st->means2D[2 * idx + 1] = piy;
        st->conic_opacity[4 * idx + 0] = conic.x;
        st->conic_opacity[4 * idx + 1] = conic.y;
        st->conic_opacity[4 * idx + 2] = conic.z;
        st->conic_opacity[4 * idx + 3] = sc->opacities[idx];
        st->tiles_touched[idx] = (maxy - miny) * (maxx - minx);
    }
    if (bad_prefilter) return -2;
    if (radii_out) std::memcpy(radii_out, st->radii.data(), sizeof(int) * P);

    // CR/rasterizer_impl.cu:281 inclusive scan; :285 num_rendered
    uint32_t run = 0;
    for (int i = 0; i < P; i++) {
        run += st->tiles_touched[i];
        st->point_offsets[i] = run;
    }
    const int N = P > 0 ? (int)run : 0;
    st->N = N;

    // CR/rasterizer_impl.cu:70-111 duplicateWithKeys
    std::vector<uint64_t> keys_unsorted(N);
    std::vector<uint32_t> vals_unsorted(N);
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1024)
    for (int idx = 0; idx < P; idx++) {
        if (st->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : st->point_offsets[idx - 1];
            uint32_t minx, miny, maxx, maxy;
            getRect(st->means2D[2 * idx], st->means2D[2 * idx + 1], st->radii[idx], gx, gy, minx, miny, maxx, maxy);
            for (uint32_t y = miny; y < maxy; y++)
                for (uint32_t x = minx; x < maxx; x++) {
                    uint64_t key = (uint64_t)y * gx + x;
                    key <<= 32;
                    key |= fbits(st->depths[idx]);
                    keys_unsorted[off] = key;
                    vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    // CR/rasterizer_impl.cu:304-312: stable radix sort on the low (32 + bit) key bits.
    const int bit = (int)getHigherMsb((uint32_t)(gx * gy));
    const uint64_t mask = (32 + bit) >= 64 ? ~0ull : ((1ull << (32 + bit)) - 1ull);
    std::vector<uint32_t> perm(N);
    std::iota(perm.begin(), perm.end(), 0u);
    std::stable_sort(perm.begin(), perm.end(),
                     [&](uint32_t a, uint32_t b) { return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask); });
    st->keys.resize(N);
    st->point_list.resize(N);
    for (int i = 0; i < N; i++) {
        st->keys[i] = keys_unsorted[perm[i]];
        st->point_list[i] = vals_unsorted[perm[i]];
    }
    // CR/rasterizer_impl.cu:314-321, 116-138 identifyTileRanges
    st->ranges.assign((size_t)2 * gx * gy, 0);
    for (int i = 0; i < N; i++) {
        uint32_t cur = (uint32_t)(st->keys[i] >> 32);
        if (i == 0)
            st->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(st->keys[i - 1] >> 32);
            if (cur != prev) {
                st->ranges[2 * prev + 1] = i;
                st->ranges[2 * cur] = i;
            }
        }
        if (i == N - 1) st->ranges[2 * cur + 1] = N;
    }
    st->n_contrib.assign((size_t)W * H, 0);
    return N;
}

int goi_oracle_forward(const GoiOracleScene* sc, GoiOracleState* st, float* out_color, float* out_semantic,
                       float* out_depth, float* out_alpha, int* radii, uint8_t* fragile, float fragile_eps,
                       int num_threads) {
    const int nt = threads_or_default(num_threads);
    const int W = sc->W, H = sc->H, S = sc->S;
    const size_t HW = (size_t)W * H;
    // DGR/rasterize_points.cu:69-73: outputs start at 0 (also the P == 0 answer, :84-85)
    std::fill(out_color, out_color + 3 * HW, 0.f);
    std::fill(out_semantic, out_semantic + (size_t)S * HW, 0.f);
    std::fill(out_depth, out_depth + HW, 0.f);
    std::fill(out_alpha, out_alpha + HW, 0.f);
    if (fragile) std::fill(fragile, fragile + HW, (uint8_t)0);
    if (sc->P == 0) {
        st->P = 0;
        st->N = 0;
        return 0;
    }
    int N = oracle_geometry_and_binning(sc, st, radii, nt);
    if (N < 0) return N;
    const float* features = sc->colors_precomp ? sc->colors_precomp : st->rgb.data();  // CR/rasterizer_impl.cu:325
    const int gx = st->gx, gy = st->gy;

    // CR/forward.cu:261-386 renderCUDA, one pixel at a time.
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        std::vector<float> Cs(S);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pixfx = (float)px, pixfy = (float)py;
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float C[3] = {0, 0, 0};
                std::fill(Cs.begin(), Cs.end(), 0.f);
                float D = 0;
                bool frag = false;
                float noise = 0.f;  // accumulated first-order effect of the exponents' fp32 rounding on this pixel
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t g = st->point_list[k];
                    const float dx = st->means2D[2 * g] - pixfx, dy = st->means2D[2 * g + 1] - pixfy;
                    const float* co = &st->conic_opacity[4 * g];
                    const float power = pair_power(co, dx, dy);
                    const float pn = pair_power_noise(co, dx, dy);
                    if (std::fabs(power) < std::max(1e-6f, pn)) frag = true;
                    if (power > 0.0f) continue;
                    const float ea = co[3] * std::exp(power);
                    const float alpha = std::min(0.99f, ea);
                    if (std::fabs(alpha - 1.0f / 255.0f) < fragile_eps * (1.0f / 255.0f)) frag = true;
                    // (... or the exponent is within its own rounding bound of the one at which alpha crosses 1/255)
                    if (pn > 1e-5f && std::fabs(power - std::log(1.0f / (255.0f * co[3]))) < pn) frag = true;
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    // (the stopping test moves with alpha: d test_T = T alpha pn)
                    if (std::fabs(test_T - 0.0001f) < fragile_eps * 0.0001f + T * alpha * pn) frag = true;
                    if (test_T < 0.0001f) break;  // done = true (CR/forward.cu:353-357)
                    if (ea < 0.99f) noise += alpha * T * pn;
                    for (int ch = 0; ch < 3; ch++) C[ch] += features[(size_t)g * 3 + ch] * alpha * T;
                    for (int ch = 0; ch < S; ch++) Cs[ch] += sc->semantics[(size_t)g * S + ch] * alpha * T;
                    D += st->depths[g] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                st->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * sc->bg[ch];
                for (int ch = 0; ch < S; ch++) out_semantic[ch * HW + pix_id] = Cs[ch];
                out_alpha[pix_id] = 1 - T;
                out_depth[pix_id] = D;
                if (fragile) fragile[pix_id] = (uint8_t)((frag ? 1 : 0) | (noise > 2.5e-5f ? 2 : 0));
            }
    }
    return N;
}

// CR/backward.cu:20-139
static void sh_backward(int idx, int deg, int max_coeffs, const float* means, V3 campos, const float* shs,
                        const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans, float* dL_dshs) {
    V3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    V3 dir_orig = pos - campos;
    V3 dir = dir_orig / std::sqrt(dot(dir_orig, dir_orig));
    const V3* sh = reinterpret_cast<const V3*>(shs) + (size_t)idx * max_coeffs;
    V3 dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
    V3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    V3* dL_dsh = reinterpret_cast<V3*>(dL_dshs) + (size_t)idx * max_coeffs;
    dL_dsh[0] = SH_C0 * dL_dRGB;
    if (deg > 0) {
        dL_dsh[1] = (-SH_C1 * y) * dL_dRGB;
        dL_dsh[2] = (SH_C1 * z) * dL_dRGB;
        dL_dsh[3] = (-SH_C1 * x) * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3];
        dRGBdy = -SH_C1 * sh[1];
        dRGBdz = SH_C1 * sh[2];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            dL_dsh[4] = (SH_C2[0] * xy) * dL_dRGB;
            dL_dsh[5] = (SH_C2[1] * yz) * dL_dRGB;
            dL_dsh[6] = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB;
            dL_dsh[7] = (SH_C2[3] * xz) * dL_dRGB;
            dL_dsh[8] = (SH_C2[4] * (xx - yy)) * dL_dRGB;
            dRGBdx = dRGBdx + (SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] +
                               SH_C2[4] * 2.f * x * sh[8]);
            dRGBdy = dRGBdy + (SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] +
                               SH_C2[4] * 2.f * -y * sh[8]);
            dRGBdz = dRGBdz + (SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7]);
            if (deg > 2) {
                dL_dsh[9] = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB;
                dL_dsh[10] = (SH_C3[1] * xy * z) * dL_dRGB;
                dL_dsh[11] = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[12] = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                dL_dsh[13] = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[14] = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
                dL_dsh[15] = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                dRGBdx = dRGBdx + (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz +
                                   SH_C3[2] * sh[11] * -2.f * xy + SH_C3[3] * sh[12] * -3.f * 2.f * xz +
                                   SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * sh[14] * 2.f * xz +
                                   SH_C3[6] * sh[15] * 3.f * (xx - yy));
                dRGBdy = dRGBdy + (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                                   SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                                   SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                                   SH_C3[6] * sh[15] * -3.f * 2.f * xy);
                dRGBdz = dRGBdz + (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                                   SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                                   SH_C3[5] * sh[14] * (xx - yy));
            }
        }
    }
    V3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
    V3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

// CR/backward.cu:278-341
static void cov3D_backward(int idx, V3 scale_, float mod, const float* rot, const float* dL_dcov3Ds, float* dL_dscales,
                           float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    M3 R = m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
              1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
              1.f - 2.f * (x * x + y * y));
    M3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    V3 s = mod * scale_;
    S.c[0][0] = s.x;
    S.c[1][1] = s.y;
    S.c[2][2] = s.z;
    M3 M = mul(S, R);
    const float* d = dL_dcov3Ds + (size_t)6 * idx;
    M3 dL_dSigma = m3(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    M3 dL_dM = mul(scale(2.0f, M), dL_dSigma);
    M3 Rt = transpose(R);
    M3 dL_dMt = transpose(dL_dM);
    dL_dscales[3 * idx + 0] = dot(col(Rt, 0), col(dL_dMt, 0));
    dL_dscales[3 * idx + 1] = dot(col(Rt, 1), col(dL_dMt, 1));
    dL_dscales[3 * idx + 2] = dot(col(Rt, 2), col(dL_dMt, 2));
    for (int j = 0; j < 3; j++) {
        dL_dMt.c[0][j] *= s.x;
        dL_dMt.c[1][j] *= s.y;
        dL_dMt.c[2][j] *= s.z;
    }
    const auto& A = dL_dMt.c;
    float qx = 2 * z * (A[0][1] - A[1][0]) + 2 * y * (A[2][0] - A[0][2]) + 2 * x * (A[1][2] - A[2][1]);
    float qy = 2 * y * (A[1][0] + A[0][1]) + 2 * z * (A[2][0] + A[0][2]) + 2 * r * (A[1][2] - A[2][1]) -
               4 * x * (A[2][2] + A[1][1]);
    float qz = 2 * x * (A[1][0] + A[0][1]) + 2 * r * (A[2][0] - A[0][2]) + 2 * z * (A[1][2] + A[2][1]) -
               4 * y * (A[2][2] + A[0][0]);
    float qw = 2 * r * (A[0][1] - A[1][0]) + 2 * x * (A[2][0] + A[0][2]) + 2 * y * (A[1][2] + A[2][1]) -
               4 * z * (A[1][1] + A[0][0]);
    dL_drots[4 * idx + 0] = qx;
    dL_drots[4 * idx + 1] = qy;
    dL_drots[4 * idx + 2] = qz;
    dL_drots[4 * idx + 3] = qw;
}

int goi_oracle_backward(const GoiOracleScene* sc, const GoiOracleState* st, const float* out_alpha,
                        const float* dL_dpix, const float* dL_dpixsem, const float* dL_dpix_depth,
                        const float* dL_dalphas, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                        float* dL_dcolor, float* dL_dsemantic, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                        float* dL_dsh, float* dL_dscale, float* dL_drot, int num_threads) {
    const int nt = threads_or_default(num_threads);
    const int P = sc->P, W = sc->W, H = sc->H, S = sc->S, M = sc->M;
    const size_t HW = (size_t)W * H;
    // DGR/rasterize_points.cu:252-262 zero-filled gradient tensors
    std::fill(dL_dmean2D, dL_dmean2D + (size_t)3 * P, 0.f);
    std::fill(dL_dconic, dL_dconic + (size_t)4 * P, 0.f);
    std::fill(dL_dopacity, dL_dopacity + P, 0.f);
    std::fill(dL_dcolor, dL_dcolor + (size_t)3 * P, 0.f);
    std::fill(dL_dsemantic, dL_dsemantic + (size_t)S * P, 0.f);
    std::fill(dL_ddepth, dL_ddepth + P, 0.f);
    std::fill(dL_dmean3D, dL_dmean3D + (size_t)3 * P, 0.f);
    std::fill(dL_dcov3D, dL_dcov3D + (size_t)6 * P, 0.f);
    if (dL_dsh && M > 0) std::fill(dL_dsh, dL_dsh + (size_t)3 * M * P, 0.f);
    std::fill(dL_dscale, dL_dscale + (size_t)3 * P, 0.f);
    std::fill(dL_drot, dL_drot + (size_t)4 * P, 0.f);
    if (P == 0) return 0;

    const int gx = st->gx, gy = st->gy;
    const float* colors = sc->colors_precomp ? sc->colors_precomp : st->rgb.data();  // CR/rasterizer_impl.cu:549
    // double accumulators for what the reference sums with float atomics (see file header)
    const int NQ = 3 + S + 1 + 2 + 3 + 1;  // colour, semantics, depth, mean2D.xy, conic.xyw, opacity
    std::vector<double> acc((size_t)NQ * P, 0.0);
    const float ddelx_dx = (float)(0.5 * W);  // CR/backward.cu:498-499
    const float ddely_dy = (float)(0.5 * H);

    // CR/backward.cu:415-625 renderCUDA, one pixel at a time, back to front.
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        if (r1 == r0) continue;
        std::vector<float> accum_recsem(S), last_sem(S), dL_dps(S);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pixfx = (float)px, pixfy = (float)py;
                const float T_final = 1 - out_alpha[pix_id];  // CR/backward.cu:466
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = st->n_contrib[pix_id];
                float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, dL_dpixel[3];
                std::fill(accum_recsem.begin(), accum_recsem.end(), 0.f);
                std::fill(last_sem.begin(), last_sem.end(), 0.f);
                float accum_depth_rec = 0, accum_alpha_rec = 0, last_alpha = 0, last_depth = 0;
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[i * HW + pix_id];
                for (int i = 0; i < S; i++) dL_dps[i] = dL_dpixsem[i * HW + pix_id];
                const float dL_dpixel_depth = dL_dpix_depth[pix_id];
                const float dL_dalpha = dL_dalphas[pix_id];
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t g = st->point_list[k];
                    const float dx = st->means2D[2 * g] - pixfx, dy = st->means2D[2 * g + 1] - pixfy;
                    const float* co = &st->conic_opacity[4 * g];
                    const float power = pair_power(co, dx, dy);
                    if (power > 0.0f) continue;
                    const float G = std::exp(power);
                    const float alpha = std::min(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    double* a = &acc[(size_t)g * NQ];
                    float dL_dopa = 0.0f;
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[(size_t)g * 3 + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dopa += (c - accum_rec[ch]) * dL_dchannel;
                        const float v = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                        a[ch] += (double)v;
                    }
                    for (int sch = 0; sch < S; sch++) {
                        const float sl = sc->semantics[(size_t)g * S + sch];
                        accum_recsem[sch] = last_alpha * last_sem[sch] + (1.f - last_alpha) * accum_recsem[sch];
                        last_sem[sch] = sl;
                        const float dL_dchannel = dL_dps[sch];
                        dL_dopa += (sl - accum_recsem[sch]) * dL_dchannel;
                        const float v = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                        a[3 + sch] += (double)v;
                    }
                    const float c_d = st->depths[g];
                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dopa += (c_d - accum_depth_rec) * dL_dpixel_depth;
                    {
                        const float v = dchannel_dcolor * dL_dpixel_depth;
#pragma omp atomic
                        a[3 + S] += (double)v;
                    }
                    accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                    dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
                    dL_dopa *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += sc->bg[i] * dL_dpixel[i];
                    dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                    const float dL_dG = co[3] * dL_dopa;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    const float v0 = dL_dG * dG_ddelx * ddelx_dx, v1 = dL_dG * dG_ddely * ddely_dy;
                    const float v2 = -0.5f * gdx * dx * dL_dG, v3 = -0.5f * gdx * dy * dL_dG,
                                v4 = -0.5f * gdy * dy * dL_dG;
                    const float v5 = G * dL_dopa;
#pragma omp atomic
                    a[4 + S] += (double)v0;
#pragma omp atomic
                    a[5 + S] += (double)v1;
#pragma omp atomic
                    a[6 + S] += (double)v2;
#pragma omp atomic
                    a[7 + S] += (double)v3;
#pragma omp atomic
                    a[8 + S] += (double)v4;
#pragma omp atomic
                    a[9 + S] += (double)v5;
                }
            }
    }
    for (int g = 0; g < P; g++) {
        const double* a = &acc[(size_t)g * NQ];
        for (int ch = 0; ch < 3; ch++) dL_dcolor[(size_t)g * 3 + ch] = (float)a[ch];
        for (int ch = 0; ch < S; ch++) dL_dsemantic[(size_t)g * S + ch] = (float)a[3 + ch];
        dL_ddepth[g] = (float)a[3 + S];
        dL_dmean2D[(size_t)g * 3 + 0] = (float)a[4 + S];
        dL_dmean2D[(size_t)g * 3 + 1] = (float)a[5 + S];
        dL_dconic[(size_t)g * 4 + 0] = (float)a[6 + S];
        dL_dconic[(size_t)g * 4 + 1] = (float)a[7 + S];
        dL_dconic[(size_t)g * 4 + 3] = (float)a[8 + S];
        dL_dopacity[g] = (float)a[9 + S];
    }

    const float focal_y = H / (2.0f * sc->tan_fovy);
    const float focal_x = W / (2.0f * sc->tan_fovx);
    const float* cov3Ds = sc->cov3D_precomp ? sc->cov3D_precomp : st->cov3D.data();  // CR/rasterizer_impl.cu:579
    const V3 campos = {sc->campos[0], sc->campos[1], sc->campos[2]};
    const float* view = sc->viewmatrix;
    const float* proj = sc->projmatrix;

#pragma omp parallel for num_threads(nt) schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(st->radii[idx] > 0)) continue;
        // ---- CR/backward.cu:144-274 computeCov2DCUDA
        {
            const float* cov3D = cov3Ds + (size_t)6 * idx;
            V3 mean = {sc->means3D[3 * idx], sc->means3D[3 * idx + 1], sc->means3D[3 * idx + 2]};
            V3 dL_dcon = {dL_dconic[4 * idx], dL_dconic[4 * idx + 1], dL_dconic[4 * idx + 3]};
            Cov2DCtx c;
            M3 cov2D;
            cov2D_common(mean, focal_x, focal_y, sc->tan_fovx, sc->tan_fovy, cov3D, view, c, cov2D);
            const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
            const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
            const auto& T = c.T.c;
            const auto& Vrk = c.Vrk.c;
            const auto& Wm = c.W.c;
            float a = cov2D.c[0][0] += 0.3f;
            float b = cov2D.c[0][1];
            float cc = cov2D.c[1][1] += 0.3f;
            float denom = a * cc - b * b;
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            float* dcov = dL_dcov3D + (size_t)6 * idx;
            if (denom2inv != 0) {
                dL_da = denom2inv * (-cc * cc * dL_dcon.x + 2 * b * cc * dL_dcon.y + (denom - a * cc) * dL_dcon.z);
                dL_dc = denom2inv * (-a * a * dL_dcon.z + 2 * a * b * dL_dcon.y + (denom - a * cc) * dL_dcon.x);
                dL_db = denom2inv * 2 * (b * cc * dL_dcon.x - (denom + 2 * b * b) * dL_dcon.y + a * b * dL_dcon.z);
                dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
                dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
                dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
                dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
                          2 * T[1][0] * T[1][1] * dL_dc;
                dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
                          2 * T[1][0] * T[1][2] * dL_dc;
                dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
                          2 * T[1][1] * T[1][2] * dL_dc;
            } else {
                for (int i = 0; i < 6; i++) dcov[i] = 0;
            }
            float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                            (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
            float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                            (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
            float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                            (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
            float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                            (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
            float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                            (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
            float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                            (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
            float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
            float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
            float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
            float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
            float tz = 1.f / c.t.z;
            float tz2 = tz * tz;
            float tz3 = tz2 * tz;
            float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
            float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
            float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * c.t.x) * tz3 * dL_dJ02 +
                           (2 * focal_y * c.t.y) * tz3 * dL_dJ12;
            V3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, view);
            dL_dmean3D[3 * idx + 0] = dL_dmean.x;  // assigns (CR/backward.cu:273)
            dL_dmean3D[3 * idx + 1] = dL_dmean.y;
            dL_dmean3D[3 * idx + 2] = dL_dmean.z;
        }
        // ---- CR/backward.cu:346-412 preprocessCUDA
        {
            V3 m = {sc->means3D[3 * idx], sc->means3D[3 * idx + 1], sc->means3D[3 * idx + 2]};
            V4 m_hom = transformPoint4x4(m, proj);
            float m_w = 1.0f / (m_hom.w + 0.0000001f);
            float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
            float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
            const float d2x = dL_dmean2D[3 * idx], d2y = dL_dmean2D[3 * idx + 1];
            V3 dL_dmean;
            dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
            dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
            dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
            dL_dmean3D[3 * idx + 0] += dL_dmean.x;
            dL_dmean3D[3 * idx + 1] += dL_dmean.y;
            dL_dmean3D[3 * idx + 2] += dL_dmean.z;
            float mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
            V3 dL_dmean2;
            dL_dmean2.x = (view[2] - view[3] * mul3) * dL_ddepth[idx];
            dL_dmean2.y = (view[6] - view[7] * mul3) * dL_ddepth[idx];
            dL_dmean2.z = (view[10] - view[11] * mul3) * dL_ddepth[idx];
            dL_dmean3D[3 * idx + 0] += dL_dmean2.x;
            dL_dmean3D[3 * idx + 1] += dL_dmean2.y;
            dL_dmean3D[3 * idx + 2] += dL_dmean2.z;
            if (sc->shs)
                sh_backward(idx, sc->D, M, sc->means3D, campos, sc->shs, st->clamped.data(), dL_dcolor, dL_dmean3D,
                            dL_dsh);
            if (sc->scales) {
                V3 s = {sc->scales[3 * idx], sc->scales[3 * idx + 1], sc->scales[3 * idx + 2]};
                cov3D_backward(idx, s, sc->scale_modifier, sc->rotations + (size_t)4 * idx, dL_dcov3D, dL_dscale,
                               dL_drot);
            }
        }
    }
    return 0;
}

// CR/forward.cu:422-551 traceCUDA (deterministic restatement, see file header)
int goi_oracle_trace(const GoiOracleScene* sc, GoiOracleState* st, const float* img_sem, float* out_color,
                     float* gau_sem, int* num_gsem, int* radii, int num_threads) {
    const int nt = threads_or_default(num_threads);
    const int W = sc->W, H = sc->H, S = sc->S, P = sc->P;
    const size_t HW = (size_t)W * H;
    std::fill(out_color, out_color + 3 * HW, 0.f);
    std::fill(gau_sem, gau_sem + (size_t)P * S, 0.f);
    std::fill(num_gsem, num_gsem + P, 0);
    if (P == 0) return 0;
    int N = oracle_geometry_and_binning(sc, st, radii, nt);
    if (N < 0) return N;
    const float* features = sc->colors_precomp ? sc->colors_precomp : st->rgb.data();
    const int gx = st->gx, gy = st->gy;
    std::vector<double> acc((size_t)P * S, 0.0);
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                float T = 1.0f;
                float C[3] = {0, 0, 0};
                uint32_t contributor = 0, last_contributor = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t g = st->point_list[k];
                    const float dx = st->means2D[2 * g] - (float)px, dy = st->means2D[2 * g + 1] - (float)py;
                    const float* co = &st->conic_opacity[4 * g];
                    const float power = pair_power(co, dx, dy);
                    if (power > 0.0f) continue;
                    const float alpha = std::min(0.99f, co[3] * std::exp(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break;
                    for (int ch = 0; ch < 3; ch++) C[ch] += features[(size_t)g * 3 + ch] * alpha * T;
                    if ((double)alpha > 0.005) {  // CR/forward.cu:521 (double literal)
                        for (int ch = 0; ch < S; ch++) {
                            acc[(size_t)g * S + ch] += (double)img_sem[ch * HW + pix_id];
                            num_gsem[g] += 1;  // inside the channel loop: +S per hit (CR/forward.cu:524)
                        }
                    }
                    T = test_T;
                    last_contributor = contributor;
                }
                st->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * sc->bg[ch];
            }
    }
    for (size_t i = 0; i < (size_t)P * S; i++) gau_sem[i] = (float)acc[i];
    return N;
}

// CR/rasterizer_impl.cu:54-66, 141-153
void goi_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                             uint8_t* present) {
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        V3 pv;
        present[i] = in_frustum(i, means3D, viewmatrix, pv) ? 1 : 0;
    }
}

int goi_oracle_state_counts(const GoiOracleState* st, int* P, int* N, int* T) {
    if (P) *P = st->P;
    if (N) *N = st->N;
    if (T) *T = st->gx * st->gy;
    return 0;
}

void goi_oracle_state_get(const GoiOracleState* st, float* depths, float* means2D, float* conic_opacity, float* rgb,
                          float* cov3D, uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_list,
                          uint64_t* point_list_keys, uint32_t* ranges, uint32_t* n_contrib) {
    auto cp = [](auto* dst, const auto& v) {
        if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
    };
    cp(depths, st->depths);
    cp(means2D, st->means2D);
    cp(conic_opacity, st->conic_opacity);
    cp(rgb, st->rgb);
    cp(cov3D, st->cov3D);
    cp(clamped, st->clamped);
    cp(tiles_touched, st->tiles_touched);
    cp(point_list, st->point_list);
    cp(point_list_keys, st->keys);
    cp(ranges, st->ranges);
    cp(n_contrib, st->n_contrib);
}

}  // extern "C"
