// Helpers shared by the tile blend kernels (forward, trace, backward).
#pragma once
#include "common.h"

namespace goi {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 0.0001f;
constexpr float kAlphaMax = 0.99f;

// alpha evaluation shared by forward, backward and trace so that all three agree on which
// (pixel, Gaussian) pairs contribute (guards of CR/forward.cu:341-351, CR/backward.cu:535-542).
struct PairEval {
    float dx, dy, power, G, alpha;
    bool hit;
};
__device__ __forceinline__ PairEval eval_pair(float gx_, float gy_, float ca, float cb, float cc, float o, float pxf,
                                              float pyf) {
    PairEval e;
    e.dx = gx_ - pxf;
    e.dy = gy_ - pyf;
    e.power = -0.5f * (ca * e.dx * e.dx + cc * e.dy * e.dy) - cb * e.dx * e.dy;
    e.G = __expf(e.power);
    e.alpha = fminf(kAlphaMax, o * e.G);
    e.hit = (e.power <= 0.0f) && (e.alpha >= kAlphaMin);
    return e;
}

// Does the Gaussian's exact contribution box (GaussRec hx/hy) touch the 8x8 quadrant whose first
// pixel is (X0, Y0)?  hx < 0: the Gaussian can never reach alpha >= 1/255.
__device__ __forceinline__ bool box_hits_quadrant(float x, float y, float hx, float hy, float X0, float Y0) {
    return (hx >= 0.f) && (x - hx <= X0 + 7.f) && (x + hx >= X0) && (y - hy <= Y0 + 7.f) && (y + hy >= Y0);
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
__device__ __forceinline__ QuadGeom quad_geom(int W, int H, int gx, int n_quads) {
    QuadGeom t;
    const int per = (int)(gridDim.x >> 3);
    const int tq = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    t.lane = threadIdx.x & 63;
    if (tq >= n_quads) {
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
