// Lane utilisation of the blend kernels, MEASURED from what a forward left behind (diagnostic; goi_raster_blend_stats).
//
// Both blend kernels give one wave to an 8x8 quadrant and one loop trip to a (quadrant, Gaussian) pair
// (render_fwd.hip, render_bwd.hip; the reference's loops are CR/forward.cu:330-372 and CR/backward.cu:523-589, one thread
// per pixel of a 16x16 tile).  How many of the 64 lanes do useful work in such a trip is a property of the scene, and
// this kernel counts it exactly: it re-walks every quadrant's list over the positions the backward walks (up to the
// quadrant's last contributor, ImageView::qcost), re-evaluates every MEMBER pair (the masks the forward recorded) with the
// same two functions the blend kernels use, and counts the lanes whose pixel composited that Gaussian (position below the
// pixel's n_contrib and both alpha guards passed -- the backward's own condition, and exactly the pairs the forward
// accumulated).  The same pass counts what denser mappings would see -- a wave split into 8x4, 4x4 or 2x2 pixel blocks
// that each walk their own list: the number of (block, Gaussian) pairs with at least one live lane -- and how many
// candidates pass the forward's quadrant hit test (its evaluated pairs).
#include "blend_common.h"

namespace goi {

namespace {

__global__ __launch_bounds__(64) void blend_stats_k(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                    int W, int H, int gx, int n_quads, const GaussRec* __restrict__ rec,
                                                    const uint32_t* __restrict__ n_contrib,
                                                    const uint32_t* __restrict__ qcost,
                                                    const unsigned long long* __restrict__ qmask0,
                                                    const unsigned long long* __restrict__ qmask,
                                                    unsigned long long* __restrict__ out) {
    __shared__ f32x4 s_geo[64];
    __shared__ f32x4 s_geo2[64];
    const QuadGeom t = quad_geom(W, H, gx, n_quads);
    if (t.tile < 0) return;
    const int tq = quad_slot();
    const int lane = t.lane;
    const uint2 range = ranges[t.tile];
    const int len = (int)(range.y - range.x);
    const int n_proc = min((int)qcost[tq], len);
    const size_t pix_id = (size_t)W * t.py + t.px;
    const uint32_t nc = t.inside ? n_contrib[pix_id] : 0u;
    const float QCX = t.QX0 + 3.5f, QCY = t.QY0 + 3.5f;
    const f32x2 uv = {t.pxf - QCX, t.pyf - QCY};
    unsigned long long c_rounds = 0, c_cand = 0, c_hits = 0, c_members = 0, c_live = 0, c_m84 = 0, c_m44 = 0, c_m22 = 0,
                       c_dead = 0;
    const int rounds = (n_proc + 63) / 64;
    for (int b = 0; b < rounds; b++) {
        const int pos = b * 64 + lane;
        const bool present = pos < n_proc;
        float4 q0 = make_float4(0, 0, 0, 0), q1 = make_float4(1.f, 0.f, -1.f, -1.f);
        if (present) {
            const uint32_t id = point_list[range.x + (uint32_t)pos];
            const float4* r4 = reinterpret_cast<const float4*>(rec + id);
            q0 = r4[0];
            q1 = r4[1];
        }
        const bool hit = ellipse_hits_quadrant(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, t.QX0, t.QY0);
        unsigned long long members = *member_mask_ptr(const_cast<unsigned long long*>(qmask0),
                                                      const_cast<unsigned long long*>(qmask), t.tile, t.q, range.x, b);
        members = uniform_u64(members);
        if (n_proc - b * 64 < 64) members &= (1ull << (n_proc - b * 64)) - 1ull;
        c_cand += (unsigned long long)__popcll(__ballot(present));
        c_hits += (unsigned long long)__popcll(__ballot(hit));
        if ((members >> lane) & 1ull) {
            const PolyCoef pc = poly_coefs(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, QCX, QCY);
            s_geo[lane] = f32x4{pc.A35.x, pc.A35.y, pc.A12.x, pc.A12.y};
            s_geo2[lane] = f32x4{pc.A0, pc.A4, pc.lim, 0.f};
        }
        __builtin_amdgcn_wave_barrier();
        c_rounds++;
        c_members += (unsigned long long)__popcll(members);
        unsigned long long m = members;
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const f32x4 g = s_geo[j], g2 = s_geo2[j];
            const PairEval e = eval_poly(g.xy, g.zw, g2.x, g2.y, g2.z, uv);
            const bool live = e.hit && (uint32_t)(b * 64 + j) < nc;
            const unsigned long long lv = __ballot(live);
            c_live += (unsigned long long)__popcll(lv);
            c_dead += lv == 0 ? 1ull : 0ull;
            c_m84 += ((lv & 0xFFFFFFFFull) != 0) + ((lv >> 32) != 0);
            c_m44 += ((lv & 0x000000000F0F0F0Full) != 0) + ((lv & 0x00000000F0F0F0F0ull) != 0) +
                     ((lv & 0x0F0F0F0F00000000ull) != 0) + ((lv & 0xF0F0F0F000000000ull) != 0);
            unsigned long long f = lv | (lv >> 1);
            f |= f >> 8;
            c_m22 += (unsigned long long)__popcll(f & 0x0055005500550055ull);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long c_pix = (unsigned long long)__popcll(__ballot(t.inside));
    unsigned long long c_nc = (unsigned long long)nc;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c_nc += __shfl_xor(c_nc, d, 64);
    if (lane == 0) {
        atomicAdd(&out[0], n_proc > 0 ? 1ull : 0ull);
        atomicAdd(&out[1], c_rounds);
        atomicAdd(&out[2], c_cand);
        atomicAdd(&out[3], c_hits);
        atomicAdd(&out[4], c_members);
        atomicAdd(&out[5], c_live);
        atomicAdd(&out[6], c_m84);
        atomicAdd(&out[7], c_m44);
        atomicAdd(&out[8], c_m22);
        atomicAdd(&out[9], c_dead);
        atomicAdd(&out[10], c_pix);
        atomicAdd(&out[11], c_nc);
    }
}

}  // namespace

void launch_blend_stats(int W, int H, const GeomView& g, const ImageView& im, const uint32_t* point_list,
                        const unsigned long long* qmask, unsigned long long* out, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int n_quads = gx * gy * 4;
    blend_stats_k<<<dim3(quad_grid(n_quads)), dim3(64), 0, s>>>(im.ranges, point_list, W, H, gx, n_quads, g.rec, im.n_contrib,
                                                                 im.qcost, im.qmask0, qmask, out);
}

}  // namespace goi
