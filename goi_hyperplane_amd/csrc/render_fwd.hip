// Forward tile blend and trace for gfx950: front-to-back compositing of RGB + S-dim semantic feature
// + depth + alpha.  One 256-thread workgroup (4 waves) per 16x16 tile, one pixel per lane; each
// WAVE owns an 8x8 pixel quadrant.
//
// Behaviour follows the reference's renderCUDA (cuda_rasterizer/forward.cu:261-386) and traceCUDA
// (forward.cu:422-551): same per-pixel front-to-back order, same guards (power > 0, alpha < 1/255,
// T(1-alpha) < 1e-4, 0.99 clamp), same outputs.  The execution design is this library's own:
//   * a batch of Gaussians is staged ONCE per workgroup into LDS including its feature row
//     (rgb, depth, semantics), so the inner loop only issues broadcast LDS reads (the reference
//     gathers features from global memory per contributing pair, forward.cu:361-364);
//   * at staging time every Gaussian is tested against each wave's 8x8 quadrant with its exact
//     contribution box (GaussRec hx/hy); the four 64-bit hit masks go to LDS and each wave walks
//     only the set bits of its own masks (scalar bit scan), so Gaussians that provably cannot
//     reach alpha >= 1/255 in a quadrant cost that wave nothing;
//   * early termination is per wave (64-bit ballot) with one workgroup vote per batch.
#include "blend_common.h"

namespace goi {

namespace {

// ------------------------------------------------------------------------------------------------
// forward (TRACE = false) and trace (TRACE = true)
// ------------------------------------------------------------------------------------------------
template <int S4, bool TRACE>
__global__ __launch_bounds__(256) void render_fwd_k(const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list, int W, int H, int gx, int S,
                                                    const GaussRec* __restrict__ rec, const float* __restrict__ semantics,
                                                    const float* __restrict__ bg, float* __restrict__ out_color,
                                                    float* __restrict__ out_sem, float* __restrict__ out_depth,
                                                    float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
                                                    const float* __restrict__ img_sem, float* __restrict__ gau_sem,
                                                    int* __restrict__ num_gsem) {
    constexpr int NF4 = TRACE ? 1 : 1 + S4;  // float4 words staged per Gaussian: (r,g,b,depth) + semantics
    constexpr int NSEM = TRACE ? 0 : 4 * S4;
    __shared__ float4 s_geo[256];
    __shared__ float2 s_geo2[256];
    __shared__ float4 s_feat[256 * NF4];
    __shared__ unsigned long long s_mask[4][4];
    __shared__ uint32_t s_id[TRACE ? 256 : 1];

    const TileGeom t = tile_geom(W, H, gx);
    const uint2 range = ranges[t.tile];
    const int len = (int)(range.y - range.x);
    const int rounds = (len + 255) / 256;
    const size_t HW = (size_t)W * H;
    const size_t pix_id = (size_t)W * t.py + t.px;
    const float X0 = (float)(t.tx * TILE), Y0 = (float)(t.ty * TILE);

    bool done = !t.inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C[4] = {0.f, 0.f, 0.f, 0.f};  // r, g, b, depth
    float Cs[NSEM > 0 ? NSEM : 1];
#pragma unroll
    for (int i = 0; i < NSEM; i++) Cs[i] = 0.f;

    for (int b = 0; b < rounds; b++) {
        const int wave_done = __all(done) ? 1 : 0;
        if (__syncthreads_and(wave_done)) break;

        // ---- stage one batch: geometry, feature row, per-quadrant hit masks
        {
            const int k = b * 256 + (int)threadIdx.x;
            bool h0 = false, h1 = false, h2 = false, h3 = false;
            if (k < len) {
                const uint32_t id = point_list[range.x + k];
                const float4* r4 = reinterpret_cast<const float4*>(rec + id);
                const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
                s_geo[threadIdx.x] = q0;
                s_geo2[threadIdx.x] = make_float2(q1.x, q1.y);
                s_feat[threadIdx.x * NF4] = make_float4(q1.w, q2.x, q2.y, q1.z);
                if constexpr (TRACE) {
                    s_id[threadIdx.x] = id;
                } else {
                    const float* srow = semantics + (size_t)id * S;
                    if ((S & 3) == 0) {
#pragma unroll
                        for (int i = 0; i < S4; i++)
                            s_feat[threadIdx.x * NF4 + 1 + i] = reinterpret_cast<const float4*>(srow)[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < S4; i++) {
                            float4 v;
                            v.x = (4 * i + 0 < S) ? srow[4 * i + 0] : 0.f;
                            v.y = (4 * i + 1 < S) ? srow[4 * i + 1] : 0.f;
                            v.z = (4 * i + 2 < S) ? srow[4 * i + 2] : 0.f;
                            v.w = (4 * i + 3 < S) ? srow[4 * i + 3] : 0.f;
                            s_feat[threadIdx.x * NF4 + 1 + i] = v;
                        }
                    }
                }
                h0 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0);
                h1 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0);
                h2 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0, Y0 + 8.f);
                h3 = box_hits_quadrant(q0.x, q0.y, q2.z, q2.w, X0 + 8.f, Y0 + 8.f);
            }
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if (t.lane == 0) {
                s_mask[0][t.w] = m0;
                s_mask[1][t.w] = m1;
                s_mask[2][t.w] = m2;
                s_mask[3][t.w] = m3;
            }
        }
        __syncthreads();

        // ---- consume: this wave walks the set bits of its own quadrant's masks, front to back
        if (!wave_done) {
            for (int sw = 0; sw < 4; sw++) {
                unsigned long long m = s_mask[t.w][sw];
                m = uniform_u64(m);
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const int gi = sw * 64 + j;
                    const float4 g = s_geo[gi];
                    const float2 g2 = s_geo2[gi];
                    const PairEval e = eval_pair(g.x, g.y, g.z, g.w, g2.x, g2.y, t.pxf, t.pyf);
                    bool c = !done && e.hit;
                    const float test_T = T * (1.f - e.alpha);
                    if (c && test_T < kTMin) {
                        done = true;
                        c = false;
                    }
                    if (__any(c)) {
                        const float wgt = c ? e.alpha * T : 0.f;
                        const float4 f0 = s_feat[gi * NF4];
                        C[0] += f0.x * wgt;
                        C[1] += f0.y * wgt;
                        C[2] += f0.z * wgt;
                        C[3] += f0.w * wgt;
                        if constexpr (!TRACE) {
#pragma unroll
                            for (int i = 0; i < S4; i++) {
                                const float4 f = s_feat[gi * NF4 + 1 + i];
                                Cs[4 * i + 0] += f.x * wgt;
                                Cs[4 * i + 1] += f.y * wgt;
                                Cs[4 * i + 2] += f.z * wgt;
                                Cs[4 * i + 3] += f.w * wgt;
                            }
                        } else {
                            if (c && (double)e.alpha > 0.005) {
                                const uint32_t id = s_id[gi];
                                for (int ch = 0; ch < S; ch++)
                                    atomicAdd(&gau_sem[(size_t)id * S + ch], img_sem[ch * HW + pix_id]);
                                atomicAdd(&num_gsem[id], S);
                            }
                        }
                        if (c) {
                            T = test_T;
                            last_contributor = (uint32_t)(b * 256 + gi + 1);
                        }
                    }
                    if (__all(done)) {
                        m = 0;
                        sw = 4;
                    }
                }
            }
        }
    }

    if (t.inside) {
        n_contrib[pix_id] = last_contributor;
        out_color[0 * HW + pix_id] = C[0] + T * bg[0];
        out_color[1 * HW + pix_id] = C[1] + T * bg[1];
        out_color[2 * HW + pix_id] = C[2] + T * bg[2];
        if constexpr (!TRACE) {
#pragma unroll
            for (int ch = 0; ch < NSEM; ch++)
                if (ch < S) out_sem[ch * HW + pix_id] = Cs[ch];
            out_alpha[pix_id] = 1.f - T;
            out_depth[pix_id] = C[3];
        }
    }
}

template <int S4>
void launch_fwd_s4(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                   float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_fwd_k<S4, false><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec,
                                                              sc.semantics, sc.bg, out_color, out_sem, out_depth,
                                                              out_alpha, im.n_contrib, nullptr, nullptr, nullptr);
}

}  // namespace

void launch_render_fwd(const GoiRasterScene& sc, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                       float* out_color, float* out_sem, float* out_depth, float* out_alpha, hipStream_t s) {
#define GOI_CALL(N) launch_fwd_s4<N>(sc, g, im, point_list, out_color, out_sem, out_depth, out_alpha, s)
    GOI_DISPATCH_S4(sc.S, GOI_CALL)
#undef GOI_CALL
}

void launch_trace_fwd(const GoiRasterScene& sc, const float* img_sem, const GeomView& g, const ImageView& im,
                      const uint32_t* point_list, float* out_color, float* gau_sem, int* num_gsem, hipStream_t s) {
    const int gx = (sc.W + TILE - 1) / TILE, gy = (sc.H + TILE - 1) / TILE;
    render_fwd_k<1, true><<<dim3(gx * gy), dim3(256), 0, s>>>(im.ranges, point_list, sc.W, sc.H, gx, sc.S, g.rec, nullptr,
                                                            sc.bg, out_color, nullptr, nullptr, nullptr, im.n_contrib,
                                                            img_sem, gau_sem, num_gsem);
}

}  // namespace goi
