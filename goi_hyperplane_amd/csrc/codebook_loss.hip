// Fused row pass of the semantic training losses for gfx950 (training half of SURVEY.md row a23 /
// 8(f) rank 2).
//
// Reference (train.py:142-163), per pixel p with rendered feature f [S], decoder (W [C,S], b [C]),
// code book LUT [C,D] and ground-truth feature g [D]:
//     P     = softmax(W f + b)                          sem_label
//     sim_c = <g/|g|, LUT_c/|LUT_c|>                    sim;  m = max_c sim_c;  label_c = (sim_c == m)
//     lab   = 50 * mean_{p,c} (P_c - label_c)^2
//     sl    = 1 - mean_p m
//     recc  = 1 - mean_p cos(LUT[argmax_c P_c], g)      (= sim at the decoder's argmax)
//     sl1   = mean_p H(softmax(t * sim)),  t = 1 (iteration < 1000) or 2
//     loss  = lab + sl + 0.3 sl1 + recc
// PyTorch runs this as two library GEMMs plus ~40 elementwise / reduction / gather / scatter kernels
// over [HW, C] tensors and keeps ~20 GB of autograd state at 1600x1056 (109 ms per iteration on
// MI355X, tools/loss_time.py).  Here the two dense contractions stay library GEMMs on the matrix
// cores (sim_raw = g^T L1^T going in, dL1 = dsim^T g coming out; hipBLASLt through torch.matmul, on
// TRANSPOSED VIEWS of the [D,H,W] map so that no permuted copy is made) and everything between them is
// this ONE kernel: it reads a row of sim once, and writes the row of dL/dsim, the pixel's dL/df and
// -- accumulated in registers across all rows a wave visits -- dL/dW, dL/db and the four loss terms.
//
// Mapping: one wave per pixel row; lane l owns codes l, l+64, ... (CPL per lane).  The decoder
// weights of a lane's codes live in its registers for the whole kernel (CPL*S VGPRs) and so do its
// dW accumulators: the [C,S] weight gradient needs no cross-lane traffic until the final write.
// The pixel's feature is wave-uniform (SGPRs, read with v_readlane from a 64-pixel register tile).
// Row statistics are DPP wave reductions whose results are wave-uniform.  All sums have a fixed
// order: the losses and gradients are bit-reproducible.
#include <float.h>

#include "common.h"

namespace goi {

namespace {

constexpr int CBL_THREADS = 256;

#define GOI_DPP(v, ctrl, rmask, oldv) \
    __builtin_amdgcn_update_dpp((oldv), (v), (ctrl), (rmask), 0xF, false)

__device__ __forceinline__ float wave_sum_u(float v) {  // total in every ... lane 63; returned uniform
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0xB1, 0xF, 0));
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0x4E, 0xF, 0));
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0x141, 0xF, 0));
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0x140, 0xF, 0));
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0x142, 0xA, 0));  // row_bcast:15 into rows 1, 3
    v += __int_as_float(GOI_DPP(__float_as_int(v), 0x143, 0xC, 0));  // row_bcast:31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_u(float v) {
    const int ninf = __float_as_int(-__builtin_inff());
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0xB1, 0xF, ninf)));
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0x4E, 0xF, ninf)));
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0x141, 0xF, ninf)));
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0x140, 0xF, ninf)));
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0x142, 0xA, ninf)));
    v = fmaxf(v, __int_as_float(GOI_DPP(__float_as_int(v), 0x143, 0xC, ninf)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_u(int v) {
    v = min(v, GOI_DPP(v, 0xB1, 0xF, 0x7FFFFFFF));
    v = min(v, GOI_DPP(v, 0x4E, 0xF, 0x7FFFFFFF));
    v = min(v, GOI_DPP(v, 0x141, 0xF, 0x7FFFFFFF));
    v = min(v, GOI_DPP(v, 0x140, 0xF, 0x7FFFFFFF));
    v = min(v, GOI_DPP(v, 0x142, 0xA, 0x7FFFFFFF));
    v = min(v, GOI_DPP(v, 0x143, 0xC, 0x7FFFFFFF));
    return __builtin_amdgcn_readlane(v, 63);
}

struct CblArgs {
    const float* sim;        // [HW][C]   raw <g, L1_c> (g not normalised)
    const float* inv_gnorm;  // [HW]      1 / |g_p|
    const float* sem;        // [S][HW]   rendered feature, channel-major (the rasterizer's output)
    const float* W;          // [C][S]
    const float* bias;       // [C] or NULL
    float* dsim;             // [HW][C]   dL/dsim_raw (already divided by |g_p|)
    float* dsem;             // [S][HW]
    float* partials;         // [n_waves][C*(S+1) + 4]: dW rows (S values + db), then lab, m, H, sim_a sums
    long long HW;
    int C, S;
    float t;        // anneal factor (1 or 2)
    float kappa;    // 2 * 50 / (HW * C)
    float inv_hw;   // 1 / HW
    float w_sl1;    // 0.3
};

template <int CPL, int SP>
__global__ __launch_bounds__(CBL_THREADS, (CPL <= 5 ? 2 : 1)) void codebook_rows_k(const CblArgs a) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (CBL_THREADS / 64) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (CBL_THREADS / 64);
    const int C = a.C, S = a.S;
    const long long HW = a.HW;

    // lane-stationary decoder rows and their gradient accumulators.  Only the LAST code slot of a lane can
    // fall beyond C (cpl = ceil(C / 64)); it is handled with selects, not branches: a padding code has
    // z = sim = -inf, hence P = q = 0 and no gradient.
    constexpr int KL = CPL - 1;
    const bool vlast = lane + 64 * KL < C;
    const int c_last = min(lane + 64 * KL, C - 1);  // clamped: always a readable address
    float Wr[CPL][SP], dWr[CPL][SP], br[CPL], dbr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        const int c = k < KL ? lane + 64 * k : c_last;
        const bool ok = k < KL || vlast;
        br[k] = (ok && a.bias) ? a.bias[c] : 0.f;
        dbr[k] = 0.f;
#pragma unroll
        for (int s = 0; s < SP; s++) {
            Wr[k][s] = (ok && s < S) ? a.W[(size_t)c * S + s] : 0.f;
            dWr[k][s] = 0.f;
        }
    }
    float acc_lab = 0.f, acc_m = 0.f, acc_H = 0.f, acc_sa = 0.f;  // wave-uniform running loss sums
    const float NEG_INF = -__builtin_inff();
    const float g_ent = a.w_sl1 * a.t * a.inv_hw;

    const long long n_chunks = (HW + 63) / 64;
    for (long long ch = wave; ch < n_chunks; ch += n_waves) {
        const long long p0 = ch * 64;
        // 64-pixel register tile of the feature map and of 1/|g|: lane l holds pixel p0 + l
        const long long pl = min(p0 + lane, HW - 1);
        float fv[SP];
#pragma unroll
        for (int s = 0; s < SP; s++) fv[s] = a.sem[(size_t)min(s, S - 1) * HW + pl];
        const float invv = a.inv_gnorm[pl];
        const int rows = (int)min((long long)64, HW - p0);
        float x[CPL];
        const float* srow = a.sim + (size_t)p0 * C;  // wave-uniform row pointers
        float* drow = a.dsim + (size_t)p0 * C;
#pragma unroll
        for (int k = 0; k < CPL; k++) x[k] = srow[k < KL ? lane + 64 * k : c_last];
        for (int i = 0; i < rows; i++, drow += C) {
            const long long p = p0 + i;
            const float inv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invv), i));
            float xs[CPL];
#pragma unroll
            for (int k = 0; k < CPL; k++) xs[k] = x[k] * inv;
            xs[KL] = vlast ? xs[KL] : NEG_INF;
            {  // prefetch the next row while this one is processed (the last row re-reads itself)
                if (p + 1 < HW) srow += C;
#pragma unroll
                for (int k = 0; k < CPL; k++) x[k] = srow[k < KL ? lane + 64 * k : c_last];
            }
            // ---- decoder logits of this lane's codes (the feature is wave-uniform)
            float fs[SP];
#pragma unroll
            for (int s = 0; s < SP; s++) fs[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fv[s]), i));
            float z[CPL];
            float zmax = NEG_INF, smax = NEG_INF;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                float acc = br[k];
#pragma unroll
                for (int s = 0; s < SP; s++) acc = fmaf(Wr[k][s], fs[s], acc);
                z[k] = acc;
            }
            z[KL] = vlast ? z[KL] : NEG_INF;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                zmax = fmaxf(zmax, z[k]);
                smax = fmaxf(smax, xs[k]);
            }
            const float mz = wave_max_u(zmax);
            const float ms = wave_max_u(smax);
            // ---- softmax of the logits, first arg-maxima, softmax of t * sim
            float P[CPL], q[CPL], lx[CPL];
            float sZp = 0.f, sZq = 0.f, sAq = 0.f, sNl = 0.f;
            int ia = 0x7FFFFFFF, is = 0x7FFFFFFF;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int c = lane + 64 * k;
                P[k] = __expf(z[k] - mz);  // exp(-inf) = 0 for padding
                sZp += P[k];
                lx[k] = a.t * (xs[k] - ms);  // <= 0
                q[k] = __expf(lx[k]);
                sZq += q[k];
                lx[k] = fmaxf(lx[k], -FLT_MAX);  // keep 0 * lx finite for padding
                sAq += q[k] * lx[k];
                ia = min(ia, z[k] == mz ? c : 0x7FFFFFFF);
                is = min(is, xs[k] == ms ? c : 0x7FFFFFFF);
                sNl += xs[k] == ms ? 1.f : 0.f;
            }
            const float Zp = wave_sum_u(sZp);
            const float Zq = wave_sum_u(sZq);
            const float Aq = wave_sum_u(sAq);
            const float nl = wave_sum_u(sNl);
            const int arg_a = wave_min_u(ia);  // argmax_c P_c   (first maximum)
            const int arg_s = wave_min_u(is);  // argmax_c sim_c (first maximum)
            const float rZp = 1.f / Zp, rZq = 1.f / Zq;
            const float logZq = __logf(Zq);
            const float Hq = logZq - Aq * rZq;  // entropy of softmax(t sim)
            float sP2 = 0.f, sPl = 0.f, sSa = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int c = lane + 64 * k;
                P[k] *= rZp;
                sP2 = fmaf(P[k], P[k], sP2);
                sPl += xs[k] == ms ? P[k] : 0.f;
                sSa += c == arg_a ? xs[k] : 0.f;
            }
            const float P2 = wave_sum_u(sP2);
            const float Pl = wave_sum_u(sPl);
            const float sim_a = wave_sum_u(sSa);
            acc_lab += (P2 - 2.f * Pl) + nl;
            acc_m += ms;
            acc_H += Hq;
            acc_sa += sim_a;
            // ---- gradients of this row
            const float Dsum = a.kappa * (P2 - Pl);
            // the feature again (a second v_readlane per channel is cheaper than 16 SGPRs kept live
            // across the reductions: they spill)
            int i2 = i;
            asm volatile("" : "+s"(i2));
            float fs2[SP];
#pragma unroll
            for (int s = 0; s < SP; s++) fs2[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fv[s]), i2));
            float dz[CPL];
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int c = lane + 64 * k;
                const float lab = xs[k] == ms ? 1.f : 0.f;
                dz[k] = P[k] * (a.kappa * (P[k] - lab) - Dsum);
                const float qk = q[k] * rZq;
                const float logq = lx[k] - logZq;
                float d = -g_ent * qk * (logq + Hq);   // d(0.3 mean H)/dsim
                d -= c == arg_s ? a.inv_hw : 0.f;      // d(1 - mean m)/dsim
                d -= c == arg_a ? a.inv_hw : 0.f;      // d(1 - mean sim_a)/dsim
                if (k < KL || vlast) drow[c] = d * inv;
                dbr[k] += dz[k];
#pragma unroll
                for (int s = 0; s < SP; s++) dWr[k][s] = fmaf(dz[k], fs2[s], dWr[k][s]);
            }
            // dL/df_s = sum_c dz_c W[c][s]: 16 lane partials, then a REDUCE-SCATTER over the wave (each exchange
            // halves the values a lane carries: 8 + 4 + 2 + 1 adds instead of 16 full reductions); lane l ends up
            // with the wave total of channel l & 15
            float part[SP];
#pragma unroll
            for (int s = 0; s < SP; s++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < CPL; k++) acc = fmaf(dz[k], Wr[k][s], acc);
                part[s] = acc;
            }
            float dfo;
            {
                static_assert(SP == 16, "the reduce-scatter below is written for 16 channels");
                const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
                float w8[8], w4[4], w2[2];
#pragma unroll
                for (int j = 0; j < 8; j++) {  // partner lane ^ 1: keep channel 2j + b0
                    const float keep = b0 ? part[2 * j + 1] : part[2 * j], send = b0 ? part[2 * j] : part[2 * j + 1];
                    w8[j] = keep + __int_as_float(GOI_DPP(__float_as_int(send), 0xB1, 0xF, 0));
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {  // partner lane ^ 2: keep channel 4j + 2 b1 + b0
                    const float keep = b1 ? w8[2 * j + 1] : w8[2 * j], send = b1 ? w8[2 * j] : w8[2 * j + 1];
                    w4[j] = keep + __int_as_float(GOI_DPP(__float_as_int(send), 0x4E, 0xF, 0));
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {  // partner lane ^ 4 (ds_swizzle, xor mode)
                    const float keep = b2 ? w4[2 * j + 1] : w4[2 * j], send = b2 ? w4[2 * j] : w4[2 * j + 1];
                    w2[j] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x101F));
                }
                {  // partner lane ^ 8
                    const float keep = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1];
                    dfo = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x201F));
                }
                dfo += __shfl_xor(dfo, 16, 64);  // the four rows of 16 lanes
                dfo += __shfl_xor(dfo, 32, 64);
            }
            if (lane < S) a.dsem[(size_t)lane * HW + p] = dfo;
        }
    }
    // ---- this wave's partial sums
    float* out = a.partials + (size_t)wave * ((size_t)C * (S + 1) + 4);
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        const int c = lane + 64 * k;
        if (k < KL || vlast) {
#pragma unroll
            for (int s = 0; s < SP; s++)
                if (s < S) out[(size_t)c * (S + 1) + s] = dWr[k][s];
            out[(size_t)c * (S + 1) + S] = dbr[k];
        }
    }
    if (lane == 0) {
        float* lo = out + (size_t)C * (S + 1);
        lo[0] = acc_lab;
        lo[1] = acc_m;
        lo[2] = acc_H;
        lo[3] = acc_sa;
    }
}


// ---- dL/dL1 = dsim^T [C x HW] * g^T [HW x 256]: split-K fp32 MFMA GEMM --------------------------
// The reduction runs over the PIXEL axis (K = HW = 1.7 M) and the output is only C x 256, so the
// classic output tiling leaves the chip idle (hipBLASLt: 5.0 ms = 52 TFLOP/s).  Here every CU owns a
// contiguous pixel range and keeps a FULL 304 x 256 partial result in the accumulators of its 8
// waves (wave w: all 19 code blocks x one feature block = 76 VGPRs; two workgroups split the 256 columns): no atomics, no output traffic
// until the final 311 KB per CU, summed afterwards in a fixed order.
//   A[m = code][k = pixel]: the dsim tile of 32 pixels x 304 codes is shared by the 8 waves through
//       LDS (row stride 308 floats: the four k lanes of a fragment read land 16 banks apart);
//   B[k = pixel][n = feature]: each lane loads one float4 = 4 consecutive pixels of ITS feature row
//       straight from the channel-major map (64 B contiguous per row across the 4 k lanes) and uses
//       register r as the B operand of k-step r: the k <-> pixel assignment (16 jj + 4 kq + r) is a
//       permutation of the tile, applied to A's addressing as well.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DL_KP = 32;       // pixels per LDS stage
constexpr int DL_THREADS = 512;  // 8 waves: a workgroup owns 128 of the 256 feature columns, wave w one block of 16

template <int NCB>
__global__ __launch_bounds__(DL_THREADS) void codebook_dlut_k(const float* __restrict__ dsim, const float* __restrict__ g,
                                                              long long HW, int C, float* __restrict__ partial) {
    constexpr int NC = NCB * 16, LDW = NC + 4;
    constexpr int PER_T = (DL_KP * NC + DL_THREADS - 1) / DL_THREADS;
    extern __shared__ __attribute__((aligned(16))) float s_tile[];  // [2][DL_KP][LDW]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kq = lane >> 4, mm = lane & 15;
    const int dh = blockIdx.x & 1, nrange = gridDim.x >> 1, range = blockIdx.x >> 1;
    const long long per = ((HW + nrange - 1) / nrange + DL_KP - 1) / DL_KP * DL_KP;
    const long long pb = (long long)range * per;
    const long long pe = min(HW, pb + per);
    const int nst = pe > pb ? (int)((pe - pb + DL_KP - 1) / DL_KP) : 0;
    const int dcol = 128 * dh + 16 * w + mm;
    const float* grow = g + (size_t)dcol * HW;

    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    float stage[PER_T];
    auto fetch_tile = [&](int st) {  // global -> registers (coalesced: the tile is one contiguous block of rows)
        const long long p0 = pb + (long long)st * DL_KP;
#pragma unroll
        for (int i = 0; i < PER_T; i++) {
            const int idx = tid + i * DL_THREADS;
            const int row = idx / NC, col = idx - row * NC;
            const long long p = p0 + row;
            stage[i] = (idx < DL_KP * NC && p < pe && col < C) ? dsim[(size_t)p * C + col] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER_T; i++) {
            const int idx = tid + i * DL_THREADS;
            const int row = idx / NC, col = idx - row * NC;
            if (idx < DL_KP * NC) s_tile[(buf * DL_KP + row) * LDW + col] = stage[i];
        }
    };
    float4 bq[2], bn[2];
    auto fetch_b = [&](int st, float4* dst) {
        const long long p0 = pb + (long long)st * DL_KP;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const long long p = p0 + 16 * jj + 4 * kq;  // pe and p are multiples of 4 (HW % 4 == 0, ranges of 32)
            dst[jj] = p < pe ? *reinterpret_cast<const float4*>(grow + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    if (nst > 0) {
        fetch_tile(0);
        fetch_b(0, bq);
        store_tile(0);
    }
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) {  // next stage's loads fly while this stage's MFMAs run
            fetch_tile(st + 1);
            fetch_b(st + 1, bn);
        }
        const float* tile = s_tile + buf * DL_KP * LDW;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const float bv[4] = {bq[jj].x, bq[jj].y, bq[jj].z, bq[jj].w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float* arow = tile + (16 * jj + 4 * kq + r) * LDW + mm;
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[16 * cb], bv[r], acc[cb], 0, 0, 0);
            }
        }
        if (st + 1 < nst) {
            store_tile(buf ^ 1);
            bq[0] = bn[0];
            bq[1] = bn[1];
        }
        __syncthreads();
    }
    // D[row = code 16 cb + 4 kq + r][col = feature 16 w + mm]
    float* out = partial + (size_t)range * NC * 256;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[(size_t)(16 * cb + 4 * kq + r) * 256 + dcol] = acc[cb][r];
}

}  // namespace

int codebook_loss_waves() { return 256 * 8; }  // persistent: 8 waves per CU (2 per SIMD at ~230 VGPRs)

int launch_codebook_rows(const float* sim, const float* inv_gnorm, const float* sem, const float* W, const float* bias,
                         long long HW, int C, int S, float t, float* dsim, float* dsem, float* partials, hipStream_t s) {
    if (S < 1 || S > 16 || C < 1 || C > 512) return -1;
    CblArgs a;
    a.sim = sim; a.inv_gnorm = inv_gnorm; a.sem = sem; a.W = W; a.bias = bias; a.dsim = dsim; a.dsem = dsem;
    a.partials = partials; a.HW = HW; a.C = C; a.S = S; a.t = t;
    a.kappa = (float)(2.0 * 50.0 / ((double)HW * (double)C));
    a.inv_hw = (float)(1.0 / (double)HW);
    a.w_sl1 = 0.3f;
    const int blocks = codebook_loss_waves() / (CBL_THREADS / 64);
    const int cpl = (C + 63) / 64;
#define GOI_CASE(N)                                                                       \
    case N:                                                                               \
        codebook_rows_k<N, 16><<<dim3(blocks), dim3(CBL_THREADS), 0, s>>>(a);              \
        break;
    switch (cpl) {
        GOI_CASE(1) GOI_CASE(2) GOI_CASE(3) GOI_CASE(4) GOI_CASE(5) GOI_CASE(6) GOI_CASE(7) GOI_CASE(8)
        default: return -1;
    }
#undef GOI_CASE
    return 0;
}

}  // namespace goi

namespace goi {
int codebook_dlut_blocks() { return 256; }  // pixel ranges; two workgroups (column halves) per range
// partial: [codebook_dlut_blocks()][304][256]; returns -1 when the shape is not the kernel's (C in 289..304,
// D = 256, HW % 4 = 0): the caller then uses a library GEMM
int launch_codebook_dlut(const float* dsim, const float* g, long long HW, int C, int D, float* partial, hipStream_t s) {
    constexpr int NCB = 19;
    if (D != 256 || C > NCB * 16 || C <= (NCB - 1) * 16 || (HW & 3) != 0) return -1;
    const size_t lds = (size_t)2 * DL_KP * (NCB * 16 + 4) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(codebook_dlut_k<NCB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    codebook_dlut_k<NCB><<<dim3(2 * codebook_dlut_blocks()), dim3(DL_THREADS), lds, s>>>(dsim, g, HW, C, partial);
    return 0;
}
}  // namespace goi
