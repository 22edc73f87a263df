// Small device-side vector/matrix helpers for the per-Gaussian kernels.
// M3 follows the column-major convention the reference's math library uses (m.c[col][row]; the
// nine-scalar constructor fills columns), so products are evaluated in the same order and the
// per-Gaussian geometry (radii, tile rectangles, depth keys) comes out bit-identical to the
// oracle when this code is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace goi {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}

struct M3 {
    float c[3][3];
};
__device__ __forceinline__ M3 make_m3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1,
                                      float c2) {
    M3 m;
    m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
    m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
    m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2;
    return m;
}
__device__ __forceinline__ M3 transpose(const M3& a) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.c[i][j] = a.c[j][i];
    return r;
}
__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
        for (int row = 0; row < 3; row++)
            r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
    return r;
}
__device__ __forceinline__ V3 column(const M3& a, int i) { return {a.c[i][0], a.c[i][1], a.c[i][2]}; }

// Row-vector transforms with the 16-float transposed matrices (CR/auxiliary.h:58-97).
__device__ __forceinline__ V3 xform_point_4x3(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ float4 xform_point_4x4(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
__device__ __forceinline__ V3 xform_vec_4x3_t(V3 p, const float* m) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// Real spherical-harmonics basis constants, degrees 0..3.
__device__ constexpr float kSH0 = 0.28209479177387814f;
__device__ constexpr float kSH1 = 0.4886025119029199f;
__device__ constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                      -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                      0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                      -0.5900435899266435f};

}  // namespace goi
