// distCUDA2 for gfx950: mean squared distance of every point to its 3 nearest neighbours
// (reference: submodules/simple-knn/simple_knn.cu:170-221 behind simple_knn._C.distCUDA2,
// spatial.cu:15-26; called once per scene by scene/gaussian_model.py:147 to size the initial
// Gaussians).
//
// The reference prunes an exhaustive search with 1024-point Morton boxes, one thread per point
// walking every box.  The result does not depend on the space partition (a box is skipped only
// when it cannot hold a nearer neighbour), so this build keeps the definition -- the 3 smallest
// fp32 squared distances to all OTHER sorted positions, (b0 + b1 + b2) / 3 -- and maps the search
// onto the CU instead:
//   * everything stays on the device: bounds by block reduction + 6 ordered-integer atomics per
//     block (no host read-back of min / max), Morton sort with the onesweep radix sort of
//     scan_sort.hip (30 key bits = 4 passes);
//   * a box is 256 consecutive sorted points = one workgroup of 4 waves.  Queries live in
//     registers, candidates are staged one box at a time through LDS (4 KB, read as same-address
//     broadcasts: no bank conflicts, one ds_read_b128 per 64 distance evaluations);
//   * pruning is two-level: 256 candidate boxes are tested per trip against the QUERY BOX inflated
//     by the workgroup's worst third-best distance (one box-box test per thread + a ballot), so
//     whole boxes are rejected without touching their points and the surviving ones are visited
//     uniformly by all 4 waves; inside, a lane whose own point is farther from the box than its
//     third best sits the box out.
// Distances are evaluated as fma(dz,dz, fma(dx,dx, dy*dy)) (this TU is compiled with
// -ffp-contract=off so that the spelling is the arithmetic), the form oracle/knn_oracle.cpp mirrors.
#include <float.h>

#include "common.h"

namespace goi {

namespace {

constexpr int KNN_BOX = 256;
// box bounds and point distances round differently; shrink the bound so rounding can never prune a
// box that holds a (by <= 1 ulp) nearer neighbour
constexpr float SLACK = 0.999999f;

__device__ __forceinline__ uint32_t ord_enc(float f) {  // order-preserving float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// enc[0..2] = min, enc[3..5] = max (ordered-integer encoding; initialised to ~0 / 0 by the launcher)
__global__ __launch_bounds__(256) void knn_bounds_k(int P, const float* __restrict__ pts, uint32_t* __restrict__ enc) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * (size_t)i + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
    __shared__ float s[4][6];
    for (int c = 0; c < 3; c++) {
        lo[c] = wave_min(lo[c]);
        hi[c] = wave_max(hi[c]);
    }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; c++) {
            s[threadIdx.x >> 6][c] = lo[c];
            s[threadIdx.x >> 6][3 + c] = hi[c];
        }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicMin(&enc[threadIdx.x], ord_enc(fminf(fminf(s[0][threadIdx.x], s[1][threadIdx.x]),
                                                   fminf(s[2][threadIdx.x], s[3][threadIdx.x]))));
    else if (threadIdx.x < 6)
        atomicMax(&enc[threadIdx.x], ord_enc(fmaxf(fmaxf(s[0][threadIdx.x], s[1][threadIdx.x]),
                                                   fmaxf(s[2][threadIdx.x], s[3][threadIdx.x]))));
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {  // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_k(int P, const float* __restrict__ pts, const uint32_t* __restrict__ enc,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t q[3];
    for (int c = 0; c < 3; c++) {
        const float lo = ord_dec(enc[c]), hi = ord_dec(enc[3 + c]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * (size_t)i + c] - lo) / ext * 1023.f : 0.f;
        t = fminf(fmaxf(t, 0.f), 1023.f);  // also maps NaN to 0
        q[c] = (uint32_t)t;
    }
    keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    vals[i] = (uint32_t)i;
}

// sorted[i] = (x, y, z, bits(original index)) in Morton order, padded to whole boxes with +inf points;
// box b = (min corner, max corner) of sorted[256 b .. 256 b + 255].
__global__ __launch_bounds__(KNN_BOX) void knn_boxes_k(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                       float4* __restrict__ sorted, float4* __restrict__ boxes) {
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    float4 p = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __uint_as_float(0xFFFFFFFFu));
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) {
        const uint32_t id = order[i];
        p = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
        lo[0] = hi[0] = p.x;
        lo[1] = hi[1] = p.y;
        lo[2] = hi[2] = p.z;
    }
    sorted[i] = p;
    __shared__ float s[4][6];
    for (int c = 0; c < 3; c++) {
        lo[c] = wave_min(lo[c]);
        hi[c] = wave_max(hi[c]);
    }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; c++) {
            s[threadIdx.x >> 6][c] = lo[c];
            s[threadIdx.x >> 6][3 + c] = hi[c];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 a, b;
        a.x = fminf(fminf(s[0][0], s[1][0]), fminf(s[2][0], s[3][0]));
        a.y = fminf(fminf(s[0][1], s[1][1]), fminf(s[2][1], s[3][1]));
        a.z = fminf(fminf(s[0][2], s[1][2]), fminf(s[2][2], s[3][2]));
        b.x = fmaxf(fmaxf(s[0][3], s[1][3]), fmaxf(s[2][3], s[3][3]));
        b.y = fmaxf(fmaxf(s[0][4], s[1][4]), fmaxf(s[2][4], s[3][4]));
        b.z = fmaxf(fmaxf(s[0][5], s[1][5]), fmaxf(s[2][5], s[3][5]));
        a.w = b.w = 0.f;
        boxes[2 * blockIdx.x] = a;
        boxes[2 * blockIdx.x + 1] = b;
    }
}

__device__ __forceinline__ void k_best(float d, float& b0, float& b1, float& b2) {  // simple_knn.cu:135-151
    const float n0 = fminf(b0, d);
    d = fmaxf(b0, d);
    const float n1 = fminf(b1, d);
    d = fmaxf(b1, d);
    b2 = fminf(b2, d);
    b0 = n0;
    b1 = n1;
}

// squared distance between a point and a box (simple_knn.cu:122-132)
__device__ __forceinline__ float box_point_d2(const float4& lo, const float4& hi, const float4& p) {
    const float dx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.f);
    const float dy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.f);
    const float dz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.f);
    return dx * dx + dy * dy + dz * dz;
}
// lower bound of the squared distance between any point of box A and any point of box B
__device__ __forceinline__ float box_box_d2(const float4& alo, const float4& ahi, const float4& blo, const float4& bhi) {
    const float dx = fmaxf(fmaxf(blo.x - ahi.x, alo.x - bhi.x), 0.f);
    const float dy = fmaxf(fmaxf(blo.y - ahi.y, alo.y - bhi.y), 0.f);
    const float dz = fmaxf(fmaxf(blo.z - ahi.z, alo.z - bhi.z), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(KNN_BOX) void knn_search_k(int P, int nbox, const float4* __restrict__ sorted,
                                                        const float4* __restrict__ boxes, float* __restrict__ out) {
    __shared__ float4 s_cand[KNN_BOX];
    __shared__ float s_red[4];
    __shared__ unsigned long long s_mask[4];
    const int qb = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int self = qb * KNN_BOX + t;
    const float4 q = sorted[self];
    const bool live = self < P;
    const float4 qlo = boxes[2 * qb], qhi = boxes[2 * qb + 1];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;

    auto scan_box = [&](int cb, bool want) {
        // all 256 threads stage the candidate box; lanes that cannot improve sit the scan out
        __syncthreads();
        s_cand[t] = sorted[(size_t)cb * KNN_BOX + t];
        __syncthreads();
        if (!want) return;
        const int base = cb * KNN_BOX;
        const int n = min(KNN_BOX, P - base);
        for (int j = 0; j < n; j++) {
            const float4 c = s_cand[j];
            const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
            const float d = fmaf(dz, dz, fmaf(dx, dx, dy * dy));
            if (base + j != self) k_best(d, b0, b1, b2);
        }
    };

    scan_box(qb, live);  // the own box first: it gives every live lane a tight third best

    for (int c0 = 0; c0 < nbox; c0 += KNN_BOX) {
        // workgroup's worst third-best bounds what any of its points can still accept
        float worst = live ? b2 : 0.f;
        worst = wave_max(worst);
        __syncthreads();
        if (lane == 0) s_red[wave] = worst;
        __syncthreads();
        worst = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        const int cb = c0 + t;
        bool near = false;
        if (cb < nbox && cb != qb) near = box_box_d2(qlo, qhi, boxes[2 * cb], boxes[2 * cb + 1]) * SLACK <= worst;
        const unsigned long long m = __ballot(near);
        if (lane == 0) s_mask[wave] = m;
        __syncthreads();
        for (int w = 0; w < 4; w++) {
            unsigned long long mw = s_mask[w];  // uniform across the workgroup
            while (mw) {
                const int bit = __builtin_ctzll(mw);
                mw &= mw - 1;
                const int cand = c0 + 64 * w + bit;
                const bool want = live && box_point_d2(boxes[2 * cand], boxes[2 * cand + 1], q) * SLACK <= b2;
                scan_box(cand, want);
            }
        }
    }
    if (live) out[__float_as_uint(q.w)] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

size_t knn_workspace_layout(int P, char* base, uint32_t** enc, uint32_t* keys[2], uint32_t* vals[2], float4** sorted,
                            float4** boxes, uint32_t** sort_scratch) {
    const size_t nbox = ((size_t)P + KNN_BOX - 1) / KNN_BOX;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    *enc = reinterpret_cast<uint32_t*>(carve(8 * sizeof(uint32_t)));
    for (int i = 0; i < 2; i++) keys[i] = reinterpret_cast<uint32_t*>(carve(sizeof(uint32_t) * (size_t)P));
    for (int i = 0; i < 2; i++) vals[i] = reinterpret_cast<uint32_t*>(carve(sizeof(uint32_t) * (size_t)P));
    *sorted = reinterpret_cast<float4*>(carve(sizeof(float4) * nbox * KNN_BOX));
    *boxes = reinterpret_cast<float4*>(carve(sizeof(float4) * 2 * nbox));
    *sort_scratch = reinterpret_cast<uint32_t*>(carve(sizeof(uint32_t) * sort_scratch_words((size_t)P)));
    return off;
}

size_t knn_workspace_bytes(int P) {
    uint32_t *enc, *keys[2], *vals[2], *sort_scratch;
    float4 *sorted, *boxes;
    return knn_workspace_layout(P, nullptr, &enc, keys, vals, &sorted, &boxes, &sort_scratch);
}

int launch_knn(int P, const float* points, float* mean_dist2, void* workspace, hipStream_t s) {
    uint32_t *enc, *keys[2], *vals[2], *sort_scratch;
    float4 *sorted, *boxes;
    knn_workspace_layout(P, static_cast<char*>(workspace), &enc, keys, vals, &sorted, &boxes, &sort_scratch);
    const int nbox = (P + KNN_BOX - 1) / KNN_BOX;
    (void)hipMemsetAsync(enc, 0xFF, 3 * sizeof(uint32_t), s);
    (void)hipMemsetAsync(enc + 3, 0x00, 3 * sizeof(uint32_t), s);
    const int rb = min(nbox, 1024);
    knn_bounds_k<<<dim3(rb), dim3(256), 0, s>>>(P, points, enc);
    knn_morton_k<<<dim3(nbox), dim3(256), 0, s>>>(P, points, enc, keys[0], vals[0]);
    const int fin = radix_sort_pairs(keys, vals, (size_t)P, 0, 30, sort_scratch, s);
    knn_boxes_k<<<dim3(nbox), dim3(KNN_BOX), 0, s>>>(P, points, vals[fin], sorted, boxes);
    knn_search_k<<<dim3(nbox), dim3(KNN_BOX), 0, s>>>(P, nbox, sorted, boxes, mean_dist2);
    return 0;
}

}  // namespace goi
