// Fused semantic-head decode for gfx950 (the inference half of SURVEY.md row a23).
//
// Reference behaviour (gui/main.py:364-386 with scene/semantic_model.py:13-50 used as ONE
// Linear(S -> n_codes, bias) -- train.py:64 -- and a code book LUT[n_codes, 256]):
//     dec   = MLP(f)                       f = rendered semantic feature of a pixel, [S]
//     idx   = argmax softmax(10 * dec)     (= argmax dec)
//     feat  = LUT[idx] / |LUT[idx]|
//     sim   = sigmoid(LinearSVM(feat))  or  vlm similarity(feat)
//     sim[sim < thresh] = 0
// Everything after the argmax depends on the code index only, so the host folds it into a table
// code_score[n_codes]; the per-pixel work is the dense contraction F[HW,S] x W^T[S,n_codes] + b and an
// argmax.  That contraction runs on the matrix cores -- for S <= 16 as three bf16 MFMAs on exact 3-way
// splits of the fp32 operands (semantic_decode3n_k, below), otherwise in fp32 (v_mfma_f32_16x16x4_f32:
// M = 16 pixels, N = 16 codes, K = 4 channels per instruction) -- and never materialises the [HW, n_codes]
// logits, the [HW, 256] gathered features or the permuted [HW, S] copy of the rasterizer output:
// the kernel reads the rasterizer's channel-major [S, H, W] tensor directly (A operand: 16
// consecutive pixels of one channel = one 64-byte segment) and writes 4-8 bytes per pixel.
#include "common.h"

namespace goi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));


// (value, index) max with lowest-index tie break across the 16 lanes of a DPP row
__device__ __forceinline__ void row_argmax(float& v, int& i) {
#define GOI_STEP(ctrl)                                                                                  \
    {                                                                                                   \
        const float ov = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xF, 0xF, false)); \
        const int oi = __builtin_amdgcn_update_dpp(i, i, ctrl, 0xF, 0xF, false);                        \
        const bool take = (ov > v) | ((ov == v) & (oi < i)); /* bitwise: no exec-mask branches */        \
        v = take ? ov : v;                                                                              \
        i = take ? oi : i;                                                                              \
    }
    GOI_STEP(0xB1)   // quad_perm [1,0,3,2]
    GOI_STEP(0x4E)   // quad_perm [2,3,0,1]
    GOI_STEP(0x141)  // row_half_mirror
    GOI_STEP(0x140)  // row_mirror
#undef GOI_STEP
}

// NBLK_T > 0: the number of 16-code blocks is a compile-time constant (19 for the reference's 300 codes): every LDS
// operand then sits at a constant offset from one base address and the block loop unrolls completely.
template <int K4, int NBLK_T>
__global__ __launch_bounds__(256, 2) void semantic_decode_k(const float* __restrict__ sem, int S, long long HW,
                                                         const float* __restrict__ Wm, const float* __restrict__ bias,
                                                         int n_codes, const float* __restrict__ code_score, float thresh,
                                                         float* __restrict__ sim_out, int* __restrict__ idx_out,
                                                         uint8_t* __restrict__ bg_mask_out) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // W^T padded: [4*K4][ncp], then bias[ncp]
    const int nblk = NBLK_T > 0 ? NBLK_T : (n_codes + 15) / 16, ncp = nblk * 16;
    float* s_b = s_w + (size_t)4 * K4 * ncp;
    for (int i = threadIdx.x; i < 4 * K4 * ncp; i += 256) {
        const int ch = i / ncp, code = i - ch * ncp;
        s_w[i] = (ch < S && code < n_codes) ? Wm[(size_t)code * S + ch] : 0.f;
    }
    for (int i = threadIdx.x; i < ncp; i += 256) s_b[i] = i < n_codes ? bias[i] : -__builtin_inff();
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, mm = lane & 15;
    const long long n_groups = (HW + 63) / 64;
    // A fragments of pixel block `it` (16 pixels): pixel 16 it + mm, channels 4 s + kq.  The blocks a wave handles
    // are 16 it for it = 4 grp + mb; the next block's loads are issued before the current block's MFMAs.
    auto load_a = [&](long long it, float (&dst)[K4]) {
        const long long pa = it * 16 + mm;
#pragma unroll
        for (int s = 0; s < K4; s++) {
            const int ch = 4 * s + kq;
            dst[s] = (ch < S && pa < HW && it >= 0) ? sem[(size_t)ch * HW + pa] : 0.f;
        }
    };
    const long long n_it = n_groups * 4, it_stride = (long long)gridDim.x * 16;
    auto next_it = [&](long long it) { return ((it & 3) != 3) ? it + 1 : it - 3 + it_stride; };
    long long it = ((long long)blockIdx.x * 4 + wave) * 4;
    float a[K4], a_next[K4];
    if (it < n_it) load_a(it, a);
    for (; it < n_it; it = next_it(it)) {
        const long long pix0 = (it >> 2) * 64;
        const int mb = (int)(it & 3);
        {
            const long long nx = next_it(it);
            load_a(nx < n_it ? nx : -1, a_next);
            float bv[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            int bi[4] = {0, 0, 0, 0};
            // B fragments (decoder weights) and the bias of code block nb + 1 are fetched from LDS while block nb is
            // on the matrix cores: left to itself the compiler issues read -> wait -> MFMA five times per block
            const float* wp = s_w + (size_t)kq * ncp + mm;
            float wv[K4], wn[K4], b0 = s_b[mm], bn = 0.f;
#pragma unroll
            for (int s = 0; s < K4; s++) wv[s] = wp[(size_t)4 * s * ncp];
#pragma unroll
            for (int nb = 0; nb < nblk; nb++) {
                if (nb + 1 < nblk) {
                    bn = s_b[(nb + 1) * 16 + mm];
#pragma unroll
                    for (int s = 0; s < K4; s++) wn[s] = wp[(size_t)4 * s * ncp + (nb + 1) * 16];
                }
                f32x4 acc = {b0, b0, b0, b0};
#pragma unroll
                for (int s = 0; s < K4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wv[s], acc, 0, 0, 0);
                const int code = nb * 16 + mm;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (acc[r] > bv[r]) {  // strict: earlier (lower) codes win ties
                        bv[r] = acc[r];
                        bi[r] = code;
                    }
                b0 = bn;
#pragma unroll
                for (int s = 0; s < K4; s++) wv[s] = wn[s];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) row_argmax(bv[r], bi[r]);
            if (mm == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const long long p = pix0 + 16 * mb + 4 * kq + r;  // D row = 4*kq + r
                    if (p < HW) {
                        const int code = bi[r];
                        float sc = code_score ? code_score[code] : 0.f;
                        const bool bg = sc < thresh;
                        if (bg) sc = 0.f;
                        if (sim_out) sim_out[p] = sc;
                        if (idx_out) idx_out[p] = code;
                        if (bg_mask_out) bg_mask_out[p] = bg ? 1 : 0;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < K4; s++) a[s] = a_next[s];
        }
    }
}

// ---- S <= 16: the same contraction at the bf16 matrix rate, to fp32 accuracy ---------------------------------
// fp32 MFMA runs at the vector rate (32 clocks per 16x16x4 instruction per SIMD) and its time adds to the VALU time
// of the co-resident waves (tools/mfma_mix_probe.hip); the four instructions per (16 pixels, 16 codes) were ~40 % of
// this kernel.  Here every fp32 operand is carried EXACTLY as three bf16 numbers, x = h + m + l (8 + 8 + 8 mantissa
// bits), and the products of weight >= 2^-24 are formed by three v_mfma_f32_16x16x32_bf16 (K = 32 = two 16-channel
// halves) accumulating in fp32:
//     [f_h | f_m] x [W_h | W_h]  =  (f_h + f_m) W_h
//     [f_h | f_m] x [W_m | W_m]  =  (f_h + f_m) W_m
//     [f_l | f_h] x [W_h | W_l]  =  f_l W_h + f_h W_l
// (dropped: f_m W_l, f_l W_m, f_l W_l <= 2^-24 of the product -- the rounding level of the fp32 chain itself).
// 3 x ~20 clocks instead of 4 x 32.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// (a, b) -> three packed bf16 pairs (a in the low half): h = rne(x), m = rne(x - h), l = rne(x - h - m)
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
    const float a1 = a - __uint_as_float(h << 16), b1 = b - __uint_as_float(h & 0xFFFF0000u);
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a1, b1}, bf16x2_t));
    const float a2 = a1 - __uint_as_float(m << 16), b2 = b1 - __uint_as_float(m & 0xFFFF0000u);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a2, b2}, bf16x2_t));
}

// The kernel.  A first version (one pixel block per operand fetch, AGPR accumulators) issued ~40 vector instructions
// around the 3 MFMAs of a (16 pixels, 16 codes) pair (profiles/README.md): accumulator traffic through AGPRs, register
// moves that rotate the prefetched B operands, four address increments, and a 44-instruction cross-lane argmax per
// output row; it was VALU-bound (SQ_ACTIVE_INST_VALU ~90 % of the kernel's cycles, LDS < 10 %).  Here
//   * NPB pixel blocks share every code block's B operands (one set of LDS reads and increments per 3 NPB MFMAs,
//     NPB independent accumulator chains);
//   * the launch bounds keep the kernel under 256 VGPRs, so the MFMAs take VGPR accumulators (no v_accvgpr moves),
//     and the bias enters as the C operand, read from LDS already replicated over the four accumulator rows;
//   * the code-block loop is unrolled by two over two operand sets (no rotation moves);
//   * the row argmax is a max reduction of the value followed by a min reduction of the index among the lanes that
//     hold the maximum (10 DPP instructions per row; same result: the lowest code among the largest logits);
//   * lanes mm = 0..3 of each 16-lane row write rows 4 kq + mm: one 64-byte store per output and pixel block.
// Per (pixel, code) the arithmetic and its order do not depend on NPB: the results are bit-identical for NPB = 1, 2, 4.
#define GOI_DPP_I(x, ctrl) __builtin_amdgcn_update_dpp((x), (x), (ctrl), 0xF, 0xF, false)
#define GOI_DPP_F(x, ctrl) __int_as_float(GOI_DPP_I(__float_as_int(x), ctrl))

__device__ __forceinline__ int row_argmax_index(float v, int i) {
    float m = v;
    m = fmaxf(m, GOI_DPP_F(m, 0xB1));   // quad_perm [1,0,3,2]
    m = fmaxf(m, GOI_DPP_F(m, 0x4E));   // quad_perm [2,3,0,1]
    m = fmaxf(m, GOI_DPP_F(m, 0x141));  // row_half_mirror
    m = fmaxf(m, GOI_DPP_F(m, 0x140));  // row_mirror
    int c = (v == m) ? i : 0x7fffffff;
    c = min(c, GOI_DPP_I(c, 0xB1));
    c = min(c, GOI_DPP_I(c, 0x4E));
    c = min(c, GOI_DPP_I(c, 0x141));
    c = min(c, GOI_DPP_I(c, 0x140));
    return c;
}

template <int NPB>
__global__ __launch_bounds__(256, 2) void semantic_decode3n_k(const float* __restrict__ sem, int S, long long HW,
                                                              const float* __restrict__ Wm, const float* __restrict__ bias,
                                                              int n_codes, const float* __restrict__ code_score, float thresh,
                                                              float* __restrict__ sim_out, int* __restrict__ idx_out,
                                                              uint8_t* __restrict__ bg_mask_out) {
    // bias replicated x4 [ncp] (16-byte aligned, first), then three bf16 planes of W, [code][16 channels]
    extern __shared__ __attribute__((aligned(16))) char s_raw3[];
    const int nblk = (n_codes + 15) / 16, ncp = nblk * 16;
    f32x4* s_b4 = reinterpret_cast<f32x4*>(s_raw3);
    uint16_t* s_h = reinterpret_cast<uint16_t*>(s_b4 + ncp);
    uint16_t* s_m = s_h + (size_t)ncp * 16;
    uint16_t* s_l = s_m + (size_t)ncp * 16;
#pragma unroll 4
    for (int i = threadIdx.x; i < ncp * 8; i += 256) {  // two channels per thread
        const int code = i >> 3, ch = (i & 7) * 2;
        const float w0 = (ch < S && code < n_codes) ? Wm[(size_t)code * S + ch] : 0.f;
        const float w1 = (ch + 1 < S && code < n_codes) ? Wm[(size_t)code * S + ch + 1] : 0.f;
        uint32_t h, m, l;
        split3_pair(w0, w1, h, m, l);
        reinterpret_cast<uint32_t*>(s_h)[i] = h;
        reinterpret_cast<uint32_t*>(s_m)[i] = m;
        reinterpret_cast<uint32_t*>(s_l)[i] = l;
    }
    for (int i = threadIdx.x; i < ncp; i += 256) {
        const float b = i < n_codes ? bias[i] : -__builtin_inff();
        s_b4[i] = f32x4{b, b, b, b};
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, mm = lane & 15;
    const int ch0 = 8 * (kq & 1);  // this lane's 8 channels of either K half
    const bool lowk = kq < 2;
    constexpr int UPIX = 16 * NPB;  // pixels of one unit of work
    const long long n_units = (HW + UPIX - 1) / UPIX, u_stride = (long long)gridDim.x * 4;
    auto load_a = [&](long long u, float (&dst)[NPB][8]) {
#pragma unroll
        for (int mb = 0; mb < NPB; mb++) {
            const long long pa = u * UPIX + 16 * mb + mm;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int ch = ch0 + i;
                dst[mb][i] = (ch < S && pa < HW && u < n_units) ? sem[(size_t)ch * HW + pa] : 0.f;
            }
        }
    };
    const uint16_t* p1 = s_h + (size_t)mm * 16 + ch0;
    const uint16_t* p2 = s_m + (size_t)mm * 16 + ch0;
    const uint16_t* p3 = (lowk ? s_h : s_l) + (size_t)mm * 16 + ch0;
    const f32x4* pb = s_b4 + mm;
    long long u = (long long)blockIdx.x * 4 + wave;
    float a[NPB][8];
    load_a(u, a);
    for (; u < n_units; u += u_stride) {
        // A operands of the unit's pixel blocks: [f_h | f_m] and [f_l | f_h]
        bf16x8_t A1[NPB], A3[NPB];
#pragma unroll
        for (int mb = 0; mb < NPB; mb++) {
            uint32_t hw[4], mw[4], lw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) split3_pair(a[mb][2 * i], a[mb][2 * i + 1], hw[i], mw[i], lw[i]);
            A1[mb] = __builtin_bit_cast(bf16x8_t, lowk ? u32x4_t{hw[0], hw[1], hw[2], hw[3]} : u32x4_t{mw[0], mw[1], mw[2], mw[3]});
            A3[mb] = __builtin_bit_cast(bf16x8_t, lowk ? u32x4_t{lw[0], lw[1], lw[2], lw[3]} : u32x4_t{hw[0], hw[1], hw[2], hw[3]});
        }
        load_a(u + u_stride, a);  // the next unit's features arrive while this one is on the matrix cores
        float bv[NPB][4];
        int bi[NPB][4];
#pragma unroll
        for (int mb = 0; mb < NPB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                bv[mb][r] = -__builtin_inff();
                bi[mb][r] = 0;
            }
        struct Ops {
            bf16x8_t b1, b2, b3;
            f32x4 bb;
        };
        auto fetch = [&](int nb, Ops& o) {
            o.b1 = *reinterpret_cast<const bf16x8_t*>(p1 + (size_t)nb * 256);
            o.b2 = *reinterpret_cast<const bf16x8_t*>(p2 + (size_t)nb * 256);
            o.b3 = *reinterpret_cast<const bf16x8_t*>(p3 + (size_t)nb * 256);
            o.bb = pb[nb * 16];
        };
        auto block = [&](int nb, const Ops& o) {
            f32x4 acc[NPB];
#pragma unroll
            for (int mb = 0; mb < NPB; mb++) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3[mb], o.b3, o.bb, 0, 0, 0);  // smallest terms first
#pragma unroll
            for (int mb = 0; mb < NPB; mb++) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[mb], o.b2, acc[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < NPB; mb++) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[mb], o.b1, acc[mb], 0, 0, 0);
            const int code = nb * 16 + mm;
#pragma unroll
            for (int mb = 0; mb < NPB; mb++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (acc[mb][r] > bv[mb][r]) {  // strict: earlier (lower) codes win ties
                        bv[mb][r] = acc[mb][r];
                        bi[mb][r] = code;
                    }
        };
        Ops even, odd;
        fetch(0, even);
        int nb = 0;
        for (; nb + 1 < nblk; nb += 2) {  // operands of block nb + 1 / nb + 2 are fetched while nb / nb + 1 computes
            fetch(nb + 1, odd);
            block(nb, even);
            if (nb + 2 < nblk) fetch(nb + 2, even);
            block(nb + 1, odd);
        }
        if (nb < nblk) block(nb, even);
#pragma unroll
        for (int mb = 0; mb < NPB; mb++) {
            int code = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int c = row_argmax_index(bv[mb][r], bi[mb][r]);
                code = (mm == r) ? c : code;
            }
            const long long p = u * UPIX + 16 * mb + 4 * kq + mm;  // D row = 4*kq + r, written by lane mm = r
            if (mm < 4 && p < HW) {
                float sc = code_score ? code_score[code] : 0.f;
                const bool bg = sc < thresh;
                if (bg) sc = 0.f;
                if (sim_out) sim_out[p] = sc;
                if (idx_out) idx_out[p] = code;
                if (bg_mask_out) bg_mask_out[p] = bg ? 1 : 0;
            }
        }
    }
}
#undef GOI_DPP_I
#undef GOI_DPP_F

}  // namespace

int launch_semantic_decode(const float* sem, int S, long long HW, const float* W, const float* bias, int n_codes,
                           const float* code_score, float thresh, float* sim_out, int* idx_out, uint8_t* bg_mask_out,
                           hipStream_t s) {
    const int K4 = (S + 3) / 4;
    const int ncp = ((n_codes + 15) / 16) * 16;
    const size_t lds = ((size_t)4 * K4 * ncp + ncp) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    const long long n_groups = (HW + 63) / 64;
    long long blocks = (n_groups + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride: W^T is staged into LDS once per workgroup
    if (blocks < 1) blocks = 1;
#define GOI_LAUNCH(KERNEL)                                                                                     \
    do {                                                                                                       \
        if (lds > 64 * 1024)                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        KERNEL<<<dim3((unsigned)blocks), dim3(256), lds, s>>>(sem, S, HW, W, bias, n_codes, code_score, thresh, \
                                                              sim_out, idx_out, bg_mask_out);                  \
    } while (0)
#define GOI_CASE(N)                          \
    case N:                                  \
        GOI_LAUNCH((semantic_decode_k<N, 0>)); \
        break;
    if (S <= 16 && g_options.decode_variant >= 1) {  // split-bf16 contraction (fp32 accuracy at the bf16 matrix rate)
        const size_t lds3 = (size_t)ncp * 16 * 2 * 3 + (size_t)ncp * 4 * sizeof(float);
        if (lds3 > 64 * 1024) return -1;
        // pixel blocks per operand fetch: 2 (default; 0.152 ms at 1600x1056, 300 codes), 4 (0.158) or 1 (0.165)
        const int npb = g_options.decode_variant == 1 ? 2 : g_options.decode_variant == 2 ? 4 : 1;
        // persistent grid: as many workgroups as are resident at once (W is split and staged once per workgroup)
        long long nb3 = ((HW + 16 * npb - 1) / (16 * npb) + 3) / 4;
        // (queried once per instantiation and LDS size: the GPUs of a node are identical)
        auto resident = [&](const void* k, long long (&cache)[2]) {
            if (cache[0] != (long long)lds3) {
                int dev = 0, cus = 256, per_cu = 0;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, lds3) != hipSuccess || per_cu < 1) per_cu = 2;
                cache[1] = (long long)cus * per_cu;
                cache[0] = (long long)lds3;
            }
            return cache[1];
        };
#define GOI_L3(N)                                                                                                  \
    do {                                                                                                           \
        static long long cache[2] = {-1, 0};                                                                       \
        const long long cap = resident(reinterpret_cast<const void*>(semantic_decode3n_k<N>), cache);              \
        if (nb3 > cap) nb3 = cap;                                                                                  \
        if (nb3 < 1) nb3 = 1;                                                                                      \
        semantic_decode3n_k<N><<<dim3((unsigned)nb3), dim3(256), lds3, s>>>(sem, S, HW, W, bias, n_codes, code_score, \
                                                                           thresh, sim_out, idx_out, bg_mask_out); \
    } while (0)
        if (npb == 4) GOI_L3(4);
        else if (npb == 2) GOI_L3(2);
        else GOI_L3(1);
#undef GOI_L3
        return 0;
    }
    if (K4 == 4 && ncp == 19 * 16) {  // the reference's configuration: S = 16 (or 13..16), 300 codes
        GOI_LAUNCH((semantic_decode_k<4, 19>));
        return 0;
    }
    switch (K4) {
        GOI_CASE(1) GOI_CASE(2) GOI_CASE(3) GOI_CASE(4) GOI_CASE(5) GOI_CASE(6) GOI_CASE(7) GOI_CASE(8)
        default: return -1;  // S > 32
    }
#undef GOI_CASE
#undef GOI_LAUNCH
    return 0;
}

}  // namespace goi
