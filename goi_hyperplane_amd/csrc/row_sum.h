// Row summation of the atomic-free backward, shared by reduce_rows.hip (reduce_rows_k, reduce_big_k) and preprocess.hip (the
// per-Gaussian backward that sums its Gaussians' rows itself: bwd_records 2).  Pure fp32 additions in slot order.
#pragma once
#include "common.h"

namespace goi {

#ifndef GOI_REDUCE_INFLIGHT
#define GOI_REDUCE_INFLIGHT 32
#endif

// Sums the rows of `cnt` consecutive instances starting at instance `inst0` (their validity words at flags32[inst0 ..]) into
// sum[] -- the lane's elements of the row -- in slot order.  Wave-synchronous: the four quarter waves of a wave call it
// together, each for its own (inst0, cnt); w_first = the validity word of instance inst0 + e (prefetched by the caller).
// COMPENSATED (reduce_big_k): Kahan summation -- the lost low bits of every addition are carried in comp[] and fed back.  A big
// Gaussian's sum runs over ten thousand rows of both signs: the plain fp32 sum's error grows with their number and depends on
// how the rows are split over quarter waves (one component of one needle's dL/dscale moved by 2e-3 of the tensor's scale
// between a 16- and a 64-part split: clustered workload, tools/diag_blown.py); the compensated sum is good to an ulp or two of
// the result whatever the split.  Four additions instead of one, on the few hundred Gaussians that take this path.
template <int K, bool COMPENSATED = false, int INFLIGHT = GOI_REDUCE_INFLIGHT>
__device__ __forceinline__ void sum_instances(const float* __restrict__ rows, const uint32_t* __restrict__ flags32,
                                              size_t inst0, uint32_t cnt, uint32_t w_first, int quarter, int e,
                                              float (&sum)[K], float (&comp)[K]) {
    constexpr int RF = 16 * K;
    auto load_flags = [&](uint32_t c) { return (c + e < cnt) ? flags32[inst0 + c + e] : 0u; };
    // every lane of the wave must reach the ballots: loop to the wave's largest count
    uint32_t cmax = cnt;
#pragma unroll
    for (int d = 32; d >= 16; d >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d, 64));
    // The instances are looked at 64 at a time: four validity words per lane, requested together (and the next 64 under this
    // block's rows), and a 16-instance chunk in which no quarter of the wave has a row is skipped on one ballot.
    uint32_t wq[4] = {w_first, 0u, 0u, 0u};
    if (cmax > 16) {
#pragma unroll
        for (int sblk = 1; sblk < 4; sblk++) wq[sblk] = load_flags(16u * sblk);
    }
    for (uint32_t c = 0; c < cmax; c += 64) {
        uint32_t wn[4] = {0u, 0u, 0u, 0u};
        if (c + 64 < cmax) {
#pragma unroll
            for (int sblk = 0; sblk < 4; sblk++) wn[sblk] = load_flags(c + 64 + 16u * sblk);
        }
#pragma unroll 1
        for (int sblk = 0; sblk < 4; sblk++) {  // (not unrolled: the queue is rotated instead of indexed)
            const uint32_t cc = c + 16u * sblk;
            if (cc >= cmax) break;                  // (wave-uniform)
            const uint32_t w = wq[0];               // 4 quadrant bytes of instance cc+e
            wq[0] = wq[1];
            wq[1] = wq[2];
            wq[2] = wq[3];
            if (__ballot(w != 0u) == 0) continue;   // (wave-uniform) no quarter has a row in this chunk
            unsigned long long m = 0;  // bit 16q + i: quadrant q of instance cc+i is valid
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long bal = __ballot(((w >> (8 * q)) & 0xFFu) != 0);
                m |= ((bal >> (16 * quarter)) & 0xFFFFull) << (16 * q);
            }
            const float* chunk = rows + (inst0 + cc) * 4 * RF;
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
                    for (int kk = 0; kk < K; kk++) {
                        if constexpr (COMPENSATED) {
                            const float y = v[i][kk] - comp[kk];
                            const float t = sum[kk] + y;
                            comp[kk] = (t - sum[kk]) - y;
                            sum[kk] = t;
                        } else {
                            sum[kk] += v[i][kk];
                        }
                    }
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
#pragma unroll
        for (int sblk = 0; sblk < 4; sblk++) wq[sblk] = wn[sblk];
    }
}

}  // namespace goi
