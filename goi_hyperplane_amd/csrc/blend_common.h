// Helpers shared by the tile blend kernels (forward, trace, backward).
#pragma once
#include "common.h"

namespace goi {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 0.0001f;
constexpr float kAlphaMax = 0.99f;
constexpr float kPowerTol = 1e-4f;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// alpha evaluation shared by forward, backward and trace so that all three agree on which
// (pixel, Gaussian) pairs contribute (guards of CR/forward.cu:341-351, CR/backward.cu:535-542).
struct PairEval {
    float E;      // opacity * exp(power), before the 0.99 clamp (CR/forward.cu:350: alpha = min(0.99, E))
    float alpha;  // min(0.99, E)
    bool below, seen;  // the two guards: power <= tolerance, alpha >= 1/255
    bool hit;          // both
};
// "does any lane satisfy all of these?" from the comparisons themselves: each ballot of a comparison IS the mask the
// v_cmp wrote, the AND is scalar -- a ballot of the combined bool makes the compiler materialise it per lane first
__device__ __forceinline__ bool any_all(bool a, bool b, bool c) {
    return (__builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b) & __builtin_amdgcn_ballot_w64(c)) != 0;
}

// alpha of a Gaussian over one 8x8 quadrant.  With (u, v) in [-3.5, 3.5] the QUADRANT-CENTRED pixel coordinates
// and dx = Dx - u, dy = Dy - v (Dx, Dy: Gaussian centre relative to the quadrant centre)
//     log2(opacity * exp(-0.5 (a dx^2 + c dy^2) - b dx dy))  =  A0 + u (A1 + A3 u + A4 v) + v (A2 + A5 v)
// (CR/forward.cu:341-350 with log2(e) and log2(opacity) folded into the coefficients).  The coefficients are
// formed once per (quadrant, Gaussian) by the lane that stages the Gaussian; a pixel then spends one packed FMA,
// three FMAs and one v_exp_f32 instead of the 11 instructions + multiply + exp + multiply of the direct form.
// Centring keeps every term small (|u|, |v| <= 3.5): the rounding error of the expanded exponent is <= ~5e-5
// for the sharpest admissible conic (a = c = 1/0.3) and ~1e-6 typically.  Forward, trace and every backward kernel
// evaluate a pair ONLY through these two functions, spelled with explicit FMAs, so that all of them take
// bit-identical contribution decisions.
//
// The reference skips a pair whose exponent is > 0 (CR/forward.cu:346).  For a positive-definite conic that can
// only be a rounding artefact next to the Gaussian's centre, where the exact exponent is ~0; the expanded form
// rounds differently there, so the guard is applied with a tolerance: power <= kPowerTol contributes (with
// exp(power) <= 1.0001).  In the folded form the guard reads  A0 + ... <= lim = log2(opacity) + kPowerTol log2(e).
//
// BUILD SWITCH  GOI_EXTRA_FLAGS=-DGOI_ALPHA_DIRECT  (python -m goi_hyperplane_amd.build --force): every kernel
// evaluates alpha in the REFERENCE'S OWN FORM instead -- power = -1/2 (a dx^2 + c dy^2) - b dx dy, skip power > 0 exactly,
// alpha = min(0.99, o exp(power)), CR/forward.cu:341-350 -- through the same two functions (the five staged words then
// hold a, c, Dx, Dy, o, b).  Slower; it exists to A/B a configuration on which the folded polynomial and the oracle
// disagree (guard flips, needles): tests/test_gpu_fuzz.py passes with either build.
struct PolyCoef {
    f32x2 A35, A12;  // (A3, A5), (A1, A2): operands of the packed FMA
    float A0, A4, lim;
};
constexpr float kLog2e = 1.4426950408889634f;
#ifdef GOI_ALPHA_DIRECT
__device__ __forceinline__ PolyCoef poly_coefs(float gx_, float gy_, float ca, float cb, float cc, float o, float qcx,
                                               float qcy) {
    PolyCoef p;
    p.A35 = f32x2{ca, cc};
    p.A12 = f32x2{gx_ - qcx, gy_ - qcy};  // dx = Dx - u, dy = Dy - v
    p.A0 = o;
    p.A4 = cb;
    p.lim = 0.f;
    return p;
}
__device__ __forceinline__ PairEval eval_poly(f32x2 A35, f32x2 A12, float A0, float A4, float lim, f32x2 uv) {
    (void)lim;
    PairEval e;
    const float dx = A12.x - uv.x, dy = A12.y - uv.y;
    const float power = -0.5f * (A35.x * dx * dx + A35.y * dy * dy) - A4 * dx * dy;
    e.E = A0 * __expf(power);
    e.alpha = fminf(kAlphaMax, e.E);
    e.below = !(power > 0.0f);
    e.seen = e.alpha >= kAlphaMin;
    e.hit = e.below && e.seen;
    return e;
}
// what the backward's flush reads back from the staged words: the conic and 1 / opacity
__device__ __forceinline__ void coef_decode(f32x4 g, f32x4 g2, float& ca, float& cb, float& cc, float& inv_o) {
    ca = g.x;
    cc = g.y;
    cb = g2.y;
    inv_o = __builtin_amdgcn_rcpf(g2.x);
}
#else
__device__ __forceinline__ void coef_decode(f32x4 g, f32x4 g2, float& ca, float& cb, float& cc, float& inv_o) {
    // conic back from A3, A4, A5 (= -log2e/2 a, -log2e b, -log2e/2 c) and 1/opacity from lim
    constexpr float kLn2 = 0.6931471805599453f;
    ca = (-2.f * kLn2) * g.x;
    cb = -kLn2 * g2.y;
    cc = (-2.f * kLn2) * g.y;
    inv_o = __builtin_amdgcn_exp2f(1e-4f * 1.4426950408889634f - g2.z);  // lim = log2(o) + kPowerTol log2(e)
}
__device__ __forceinline__ PolyCoef poly_coefs(float gx_, float gy_, float ca, float cb, float cc, float o, float qcx,
                                               float qcy) {
    // A0 and A1, A2 are differences of terms of size a Dx^2 and a Dx, which for a long thin Gaussian (thin across,
    // hundreds of pixels along) are 1e3..1e6 times the result -- in fp32 the exponent would be off by up to ~0.3
    // there (the reference's own per-pixel evaluation has that same error).  Whenever the terms are large (S >= 16:
    // fp32 would lose more than ~3e-6) the coefficients are formed in DOUBLE precision and then rounded; with exact
    // coefficients the error is bounded by the 5e-5 of the quadrant-centred polynomial whatever the Gaussian's size.
    // One lane does this once per (quadrant, Gaussian); the fp64 path is skipped when no lane of the wave needs it.
    const float Dxf = gx_ - qcx, Dyf = gy_ - qcy;
    const float S = fabsf(ca) * Dxf * Dxf + 2.f * fabsf(cb * Dxf * Dyf) + fabsf(cc) * Dyf * Dyf;
    // (which path a Gaussian takes depends on ITS OWN S only -- forward and backward stage it in different company
    // and must get the same bits; the wave-uniform test merely skips the fp64 instructions when nobody needs them)
    const bool wide = S >= 16.f;
    const float lo = __builtin_amdgcn_logf(o);  // v_log_f32 = log2; opacity 0 gives -inf: never contributes
    const float f1 = fmaf(ca, Dxf, cb * Dyf), f2 = fmaf(cc, Dyf, cb * Dxf);
    float A0 = fmaf(kLog2e, -0.5f * fmaf(Dxf, f1, Dyf * f2), lo), A1 = kLog2e * f1, A2 = kLog2e * f2;
    if (__builtin_amdgcn_ballot_w64(wide) != 0) {
        constexpr double L = 1.4426950408889634;
        const double Dx = (double)gx_ - (double)qcx, Dy = (double)gy_ - (double)qcy;
        const double a = ca, b = cb, c = cc;
        const double d1 = a * Dx + b * Dy;
        const double d2 = c * Dy + b * Dx;
        const double d0 = -0.5 * (Dx * d1 + Dy * d2);  // -0.5 (a Dx^2 + 2 b Dx Dy + c Dy^2)
        const float w0 = (float)(L * d0 + (double)lo), w1 = (float)(L * d1), w2 = (float)(L * d2);
        A0 = wide ? w0 : A0;
        A1 = wide ? w1 : A1;
        A2 = wide ? w2 : A2;
    }
    PolyCoef p;
    p.A0 = A0;
    p.A12 = f32x2{A1, A2};
    p.A35 = f32x2{(-0.5f * kLog2e) * ca, (-0.5f * kLog2e) * cc};
    p.A4 = -kLog2e * cb;
    p.lim = lo + kPowerTol * kLog2e;
    return p;
}
// uv = this lane's (u, v)
__device__ __forceinline__ PairEval eval_poly(f32x2 A35, f32x2 A12, float A0, float A4, float lim, f32x2 uv) {
    PairEval e;
    const f32x2 t = __builtin_elementwise_fma(A35, uv, A12);  // (A3 u + A1, A5 v + A2)
    const float t1 = fmaf(A4, uv.y, t.x);
    const float P = fmaf(uv.y, t.y, fmaf(uv.x, t1, A0));
    e.E = __builtin_amdgcn_exp2f(P);
    e.alpha = fminf(kAlphaMax, e.E);
    e.below = P <= lim;
    e.seen = e.alpha >= kAlphaMin;
    e.hit = e.below && e.seen;
    return e;
}
#endif  // GOI_ALPHA_DIRECT

// ---- split operands of the backward kernels' MFMA reductions (render_bwd.hip, "Two flushes"; the bf16 helpers also serve
// the loss kernels, codebook_loss.hip)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// [member row][pixel] transposition buffer of the split-f16 flush (render_bwd.hip): rows 0..7 = w, rows 8..15 = h of the
// group's members (render_bwd_sem.hip: 16 w rows); lane (kq, mm) reads member row 2 (mm >> 2) + (mm & 1) of either set at
// pixels 32 c + 8 kq, c = (mm >> 1) & 1, with ds_read_b128, which the LDS serves in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ...: a stride of 72 floats plus 4 for odd rows keeps the 16 lanes of every group on 16 different
// bank quads (searched exhaustively over strides and per-bit offsets, then timed on the device: tools/probes/lds_probe.hip).
__device__ __forceinline__ constexpr int f16_row(int r) { return (r & 7) * 72 + 4 * (r & 1) + (r >> 3) * 572; }
constexpr int F16_FLOATS = 2 * 572;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// eight consecutive K values -> f16 hi = rne(x) and lo = rne(x - hi): |x - hi - lo| <= 2^-22 |x| while lo stays a normal
// f16 number (|x| >= 2^-2), 2^-25 absolute below -- the caller scales x towards 2^15
__device__ __forceinline__ void split16_pack8(const float (&y)[8], f16x8& h, f16x8& l) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f16x2 hh = __builtin_convertvector(f32x2{y[2 * i], y[2 * i + 1]}, f16x2);  // v_cvt_pk_f16_f32
        // x - hi as ONE mixed-precision FMA per value (the compiler expands the f16 halves with two conversions first)
        const uint32_t hb = __builtin_bit_cast(uint32_t, hh);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(y[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(y[2 * i + 1]));
        const f16x2 ll = __builtin_convertvector(f32x2{r0, r1}, f16x2);
        hw[i] = __builtin_bit_cast(uint32_t, hh);
        lw[i] = __builtin_bit_cast(uint32_t, ll);
    }
    h = __builtin_bit_cast(f16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    l = __builtin_bit_cast(f16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// LDS-DMA: 16 bytes per active lane straight from global memory to LDS, lane l's at lds_base + 16 l (lds_base wave-uniform; the
// instruction takes it in M0), tracked by vmcnt like any load (tools/probes/lds_dma_probe.hip checks the addressing).  The
// builtin exists in the device pass only: hipcc's host pass, which parses kernel bodies too, drops a kernel that names it.
__device__ __forceinline__ void lds_dma16(const void* global_src, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(global_src, lds_base, 16, 0, 0);
#endif
}

// Largest value of a wave's 64 lanes, as a scalar.  (__shfl_xor compiles to ds_bpermute: six dependent LDS round trips.  A
// DPP butterfly inside the 16-lane rows + four v_readlane, and v_permlane16/32_swap for the column maximum below, avoid
// the LDS altogether and were measured 4-5 us SLOWER on the backward blend, same box, two runs each: 0.532 vs 0.527 ms.)
__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d, 64));
    return __builtin_amdgcn_readfirstlane(x);
}
// max over the four lanes mm + 16 kq (kq = 0..3) of every column mm, in all of them
__device__ __forceinline__ float column_max_f32(float x) {
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
}

// B operand of the split-f16 flush for ONE 16-column block: y[chunk][i] = dL[pixel 32 chunk + 8 kq + i][column mm] of lane
// (kq, mm).  The column is scaled by 2^k = 2^(14 - exponent(largest |y| of the column over the quadrant's 64 pixels)) (k
// clamped to <= 99: upstream gradients below 2^-85 keep fewer bits; largest = Inf: k = -114 and Inf stays Inf; a NaN is
// not seen by fmaxf and comes out of the products as NaN) and split into planes; `unscale` = 2^-15 / 2^k undoes it and the
// 2^15 of the weights (f16_a_operands).  render_bwd_rows_k and render_bwd_sem_k share this function, the A operands and
// the instruction order of the products: their dL/dsemantics are bit-identical.
__device__ __forceinline__ void f16_b_operand(float (&y)[2][8], f16x8 (&hi)[2], f16x8 (&lo)[2], float& unscale) {
    float big = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
        for (int i = 0; i < 8; i++) big = fmaxf(big, fabsf(y[c2][i]));
    big = column_max_f32(big);  // the column's other pixels: lanes mm + 16 kq
    const int e = (int)((__float_as_uint(big) >> 23) & 0xFFu);
    const int fs = min(268 - e, 226);
    const float scale = __uint_as_float((uint32_t)fs << 23);
    unscale = __uint_as_float((uint32_t)(239 - fs) << 23);
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
#pragma unroll
        for (int i = 0; i < 8; i++) y[c2][i] *= scale;
        split16_pack8(y[c2], hi[c2], lo[c2]);
    }
}

// A operands of the split-f16 flush for eight members (rows r0 .. r0 + 7 of the transposition buffer at f16_row()).
// A[row mm] = plane (mm >> 1) & 1 (f16 hi / lo) of the weights of member 2 (mm >> 2) + (mm & 1), so that lane (kq, mm) of
// D = A B holds, for column mm, the hi products of members 2 kq and 2 kq + 1 in elements 0, 1 and their lo products in
// elements 2, 3.  A lane reads and splits 8 pixels of ONE row -- chunk = its plane: f16_lane_offset() -- and takes the
// plane it needs of the other chunk from its partner lane mm ^ 2:
//   A0 (pixels 0..31)  = plane-0 lanes: own hi,          plane-1 lanes: the partner's lo;
//   A1 (pixels 32..63) = plane-0 lanes: the partner's hi, plane-1 lanes: own lo.
// The partner exchange is quad_perm [2, 3, 0, 1] fused into the select (v_cndmask_b32_dpp: D = vcc ? src1 : dpp(src0));
// the leading s_nop covers the two wait states a DPP read needs after a VALU write of its source.
// The weights, in (0, 1], are split as w 2^15: two f16 planes then carry 22 bits of every weight above 2^-17 and the
// smallest ones (alpha T ~ 2^-21) to 18.
__device__ __forceinline__ int f16_lane_offset(int mm, int kq) {
    return f16_row(2 * (mm >> 2) + (mm & 1)) + (((mm >> 1) & 1) ? 32 : 0) + 8 * kq;
}
__device__ __forceinline__ void f16_a_operands(const f32x4* src, f16x8& A0, f16x8& A1) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const f32x4 b0 = src[0] * 32768.f, b1 = src[1] * 32768.f;
    const float y[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    f16x8 wh, wl;
    split16_pack8(y, wh, wl);
    const u32x4 whu = __builtin_bit_cast(u32x4, wh), wlu = __builtin_bit_cast(u32x4, wl);
    constexpr unsigned long long plane0_mask = 0x3333333333333333ull, plane1_mask = ~plane0_mask;  // lanes with bit 1 clear / set
    u32x4 P0, P1;
    asm volatile(
        "s_nop 1\n"
        "s_mov_b64 vcc, %16\n"
        "v_cndmask_b32_dpp %0, %12, %8, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %1, %13, %9, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %2, %14, %10, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %3, %15, %11, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_mov_b64 vcc, %17\n"
        "v_cndmask_b32_dpp %4, %8, %12, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %5, %9, %13, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %6, %10, %14, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_cndmask_b32_dpp %7, %11, %15, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        : "=&v"(P0[0]), "=&v"(P0[1]), "=&v"(P0[2]), "=&v"(P0[3]), "=&v"(P1[0]), "=&v"(P1[1]), "=&v"(P1[2]), "=&v"(P1[3])
        : "v"(whu[0]), "v"(whu[1]), "v"(whu[2]), "v"(whu[3]), "v"(wlu[0]), "v"(wlu[1]), "v"(wlu[2]), "v"(wlu[3]),
          "s"(plane0_mask), "s"(plane1_mask)
        : "vcc");
    A0 = __builtin_bit_cast(f16x8, P0);
    A1 = __builtin_bit_cast(f16x8, P1);
}

// (a, b) -> packed bf16 pairs: hi = rne(a), rne(b) (a in the low half), lo = rne(a - hi_a), rne(b - hi_b)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));  // v_cvt_pk_bf16_f32
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a - ah, b - bh}, bf16x2));
}
// eight consecutive K values of an MFMA operand -> hi, lo and a THIRD plane t = rne(x - hi - lo): hi + lo + t carries x
// to ~2^-25 (against a B operand that is exact in bf16 -- the moment basis -- the product is then fp32-grade)
__device__ __forceinline__ void split3_pack8(const float (&y)[8], bf16x8& h, bf16x8& l, bf16x8& t) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t hw[4], lw[4], tw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float a = y[2 * i], b = y[2 * i + 1];
        hw[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
        const float a1 = a - __uint_as_float(hw[i] << 16), b1 = b - __uint_as_float(hw[i] & 0xFFFF0000u);
        lw[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a1, b1}, bf16x2));
        const float a2 = a1 - __uint_as_float(lw[i] << 16), b2 = b1 - __uint_as_float(lw[i] & 0xFFFF0000u);
        tw[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a2, b2}, bf16x2));
    }
    h = __builtin_bit_cast(bf16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    l = __builtin_bit_cast(bf16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
    t = __builtin_bit_cast(bf16x8, u32x4{tw[0], tw[1], tw[2], tw[3]});
}
// eight consecutive K values of an MFMA operand -> hi and lo planes
__device__ __forceinline__ void split_pack8(const float (&y)[8], bf16x8& h, bf16x8& l) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) split_pair(y[2 * i], y[2 * i + 1], hw[i], lw[i]);
    h = __builtin_bit_cast(bf16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    l = __builtin_bit_cast(bf16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// Does the Gaussian's exact contribution box (GaussRec hx/hy) touch the 8x8 quadrant whose first
// pixel is (X0, Y0)?  hx < 0: the Gaussian can never reach alpha >= 1/255.
__device__ __forceinline__ bool box_hits_quadrant(float x, float y, float hx, float hy, float X0, float Y0) {
    return (hx >= 0.f) && (x - hx <= X0 + 7.f) && (x + hx >= X0) && (y - hy <= Y0 + 7.f) && (y + hy >= Y0);
}

// Can the Gaussian reach alpha >= 1/255 at some point of the 8x8 quadrant whose first pixel is (X0, Y0)?  Exact test
// against the ELLIPSE  1/2 (a dx^2 + c dy^2) + b dx dy <= tau  (tau = the inflated ln(255 opacity) the contribution
// box was built from, preprocess_fwd_k): the minimum of the convex quadratic over the rectangle is 0 when the
// centre lies inside and otherwise sits on one of the four edges, where it is a clamped 1-D minimisation.  The
// bounding box of the ellipse (box_hits_quadrant) keeps a quadrant whenever the BOX touches it: for round
// Gaussians 1 - pi/4 of the box is empty, for elongated diagonal ones most of it.  Every candidate the test
// removes saves the staging of the Gaussian and a 64-lane evaluation that could not contribute.  Conservative:
// the continuous minimum over the rectangle bounds the minimum over its pixel centres, tau carries the 1 % + 0.01
// margin of the box, and an unknown bound (hx = +inf) keeps the candidate.  hx < 0: can never contribute (also
// marks absent lanes).
__device__ __forceinline__ bool ellipse_hits_quadrant(float x, float y, float ca, float cb, float cc, float o, float hx,
                                                      float hy, float X0, float Y0) {
    if (!box_hits_quadrant(x, y, hx, hy, X0, Y0)) return false;
    if (!(hx < 3.0e38f)) return true;  // no bound known
    const float ux0 = X0 - x, ux1 = ux0 + 7.f, uy0 = Y0 - y, uy1 = uy0 + 7.f;  // rectangle relative to the centre
    if (ux0 <= 0.f && ux1 >= 0.f && uy0 <= 0.f && uy1 >= 0.f) return true;     // centre inside
    const float tau = 1.01f * 0.6931471805599453f * __builtin_amdgcn_logf(255.f * o) + 0.0101f;
    const float rb_c = -cb * __builtin_amdgcn_rcpf(cc), rb_a = -cb * __builtin_amdgcn_rcpf(ca);
    // The minimiser along an edge may be slightly off (fp32): being at a minimum that costs nothing.  The VALUE is a
    // difference of terms that, for a long thin Gaussian, are 1e3..1e6 times the result: every edge value is
    // therefore lowered by a bound on its fp32 error (1e-6 x the sum of the terms' magnitudes; the true bound is
    // ~6e-7) before it is compared -- the test must never remove a candidate that contributes, while keeping one
    // too many only costs an evaluation.
    auto edge = [&](float dx, float dy) {
        const float t1 = 0.5f * ca * dx * dx, t2 = 0.5f * cc * dy * dy, t3 = cb * dx * dy;
        return (t1 + t2 + t3) - 1e-6f * (fabsf(t1) + fabsf(t2) + fabsf(t3));
    };
    const float qmin = fminf(fminf(edge(ux0, fminf(fmaxf(rb_c * ux0, uy0), uy1)), edge(ux1, fminf(fmaxf(rb_c * ux1, uy0), uy1))),
                             fminf(edge(fminf(fmaxf(rb_a * uy0, ux0), ux1), uy0), edge(fminf(fmaxf(rb_a * uy1, ux0), ux1), uy1)));
    return !(qmin > tau * 1.0001f + 1e-4f);  // (NaN keeps the candidate)
}

// One workgroup = one 16x16 tile; wave w owns quadrant (w&1, w>>1); lane l owns pixel
// (l&7, l>>3) of the quadrant.
struct TileGeom {
    int tile, tx, ty, w, lane, px, py;
    bool inside;
    float pxf, pyf;
};
__device__ __forceinline__ TileGeom tile_geom(int W, int H, int gx) {
    TileGeom t;
    t.tile = blockIdx.x;
    t.tx = t.tile % gx;
    t.ty = t.tile / gx;
    t.w = threadIdx.x >> 6;
    t.lane = threadIdx.x & 63;
    t.px = t.tx * TILE + (t.w & 1) * 8 + (t.lane & 7);
    t.py = t.ty * TILE + (t.w >> 1) * 8 + (t.lane >> 3);
    t.inside = t.px < W && t.py < H;
    t.pxf = (float)t.px;
    t.pyf = (float)t.py;
    return t;
}

// Wave-per-quadrant geometry: one 64-thread workgroup per 8x8 quadrant.  Hardware deals
// workgroup b to XCD b % 8, so quadrant index tq = (b % 8) * per + b / 8 gives every XCD a
// contiguous band of tiles (all four quadrants of a tile and its neighbours share one L2).
struct QuadGeom {
    int tile, q, tx, ty, lane, px, py;
    bool inside;
    float pxf, pyf, QX0, QY0;
};
// launch slot of this workgroup in band-major order: slot = (XCD the hardware deals it to) * per + position in the band
__device__ __forceinline__ int quad_slot() {
    const int per = (int)(gridDim.x >> 3);
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}
__device__ __forceinline__ QuadGeom quad_geom_of(int tq, int W, int H, int gx, int n_quads) {
    QuadGeom t;
    t.lane = threadIdx.x & 63;
    if (tq < 0 || tq >= n_quads) {
        t.tile = -1;
        return t;
    }
    t.tile = tq >> 2;
    t.q = tq & 3;
    t.tx = t.tile % gx;
    t.ty = t.tile / gx;
    const int qx0 = t.tx * TILE + (t.q & 1) * 8, qy0 = t.ty * TILE + (t.q >> 1) * 8;
    t.px = qx0 + (t.lane & 7);
    t.py = qy0 + (t.lane >> 3);
    t.inside = t.px < W && t.py < H;
    t.pxf = (float)t.px;
    t.pyf = (float)t.py;
    t.QX0 = (float)qx0;
    t.QY0 = (float)qy0;
    return t;
}
__device__ __forceinline__ QuadGeom quad_geom(int W, int H, int gx, int n_quads) {
    return quad_geom_of(quad_slot(), W, H, gx, n_quads);
}
inline int quad_grid(int n_quads) { return 8 * ((n_quads + 7) / 8); }

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long m) {
    return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
           (unsigned int)__builtin_amdgcn_readfirstlane((int)(m & 0xFFFFFFFFull));
}

#define GOI_DISPATCH_S4(S, CALL)                     \
    switch (((S) + 3) / 4) {                         \
        case 1: CALL(1); break;                      \
        case 2: CALL(2); break;                      \
        case 3: CALL(3); break;                      \
        case 4: CALL(4); break;                      \
        case 5: CALL(5); break;                      \
        case 6: CALL(6); break;                      \
        case 7: CALL(7); break;                      \
        default: CALL(8); break;                     \
    }

}  // namespace goi
