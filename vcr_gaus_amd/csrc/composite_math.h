// Device helpers of the compositing kernels (composite.hip; also used by the row-stream experiment kept under
// profiles/experiments/r2_composite_rows.hip).  Everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include "vcr_common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// Sum over the 64 lanes, result valid in every lane.
__device__ __forceinline__ float wave_sum(float x) {
    x = dpp_add<0x128>(x);   // row_ror:8
    x = dpp_add<0x124>(x);   // row_ror:4
    x = dpp_add<0x122>(x);   // row_ror:2
    x = dpp_add<0x121>(x);   // row_ror:1
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

// Maximum over the 64 lanes (non-negative values), result valid in every lane.
__device__ __forceinline__ int wave_max_i32(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false));   // row_ror:8
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false));   // row_ror:4
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false));   // row_ror:2
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false));   // row_ror:1
    x = max(x, __shfl_xor(x, 16));
    x = max(x, __shfl_xor(x, 32));
    return x;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row_sum16(float x) {
    x = dpp_add<0x128>(x);
    x = dpp_add<0x124>(x);
    x = dpp_add<0x122>(x);
    return dpp_add<0x121>(x);
}


__device__ __forceinline__ float edge_min(float a2, float c2, float b, float xe, float ic, float y0, float y1) {
    // min over y in [y0,y1] of 0.5*(a2*xe^2... ) with a2=A, c2=C: Q = 0.5*A*xe^2 + B*xe*y + 0.5*C*y^2
    const float ys = fminf(y1, fmaxf(y0, -b * xe * ic));
    return 0.5f * a2 * xe * xe + b * xe * ys + 0.5f * c2 * ys * ys;
}


// Can this Gaussian reach alpha >= 1/255 at any pixel centre of the rectangle [X0,X0+bw]x[Y0,Y0+bh]?  Conservative.
// (`tau` = ln(255 opacity) >= 0 and the raw conic as arguments: the grouped backward keeps them per staged survivor and repeats
//  the test on its four 4x4 sub-blocks with exactly this arithmetic.)
__device__ __forceinline__ bool quad_touch_tau(float px, float py, float A, float B, float C, float tau, float X0, float Y0,
                                               float bw = 7.f, float bh = 7.f) {
    if (!(tau >= 0.f)) return false;                      // opacity below 1/255 never contributes
    if (!(A > 0.f) || !(C > 0.f)) return true;            // degenerate conic: leave it to the per-pixel test
    const float x0 = X0 - px, x1 = X0 + bw - px, y0 = Y0 - py, y1 = Y0 + bh - py;
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return true;
    const float ia = 1.f / A, ic = 1.f / C;
    float qm = edge_min(A, C, B, x0, ic, y0, y1);
    qm = fminf(qm, edge_min(A, C, B, x1, ic, y0, y1));
    qm = fminf(qm, edge_min(C, A, B, y0, ia, x0, x1));
    qm = fminf(qm, edge_min(C, A, B, y1, ia, x0, x1));
    const float mx = fmaxf(x0 * x0, x1 * x1), my = fmaxf(y0 * y0, y1 * y1);
    return qm <= tau + 0.05f + 2e-6f * (A * mx + C * my);
}
__device__ __forceinline__ bool quad_touch(const float4 q0, const float4 q1, float X0, float Y0, float bw = 7.f,
                                           float bh = 7.f) {
    return quad_touch_tau(q0.x, q0.y, q1.x, q1.y, q1.z, __logf(255.f * q0.w), X0, Y0, bw, bh);
}

// Can the Gaussian reach alpha >= 1/255 at a pixel centre of tile (tx, ty)?  The same conservative conic-minimum test the
// compositing kernels apply per 8x8 quad, on the tile's 16x16 pixel rectangle: a tile rejected here would have been culled
// by all four of its quad waves, so the rendered image, the n_contrib semantics (positions are relative to the tile's own
// list, forward and backward alike) and every gradient are unchanged -- only the lists, the tile sort and the chunk
// culling get shorter.  Used by the projection kernel (exact per-tile rejection, preprocess.hip).
__device__ __forceinline__ bool tile_touch(float4 q0, float4 q1, int tx, int ty) {
    return quad_touch(q0, q1, (float)(tx * VCR_TILE), (float)(ty * VCR_TILE), (float)(VCR_TILE - 1), (float)(VCR_TILE - 1));
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue two fp32 lanes-worth per instruction on CDNA3/4):
// the shading loops are VALU-issue-bound, so everything that comes in natural pairs is written on this type.
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }

// Exponent of one Gaussian at one pixel in base 2: alpha_raw = opacity * exp(power) = 2^e with
//   e = log2(opacity) + 0.5 * (u . d),  u = s d,  s = -log2(e) * conic (scaled once per Gaussian by its culler lane),
// d = (gaussian centre - pixel).  `u` (= d e / d d) is what the backward needs anyway; `hs` = u . d has the sign of the
// reference's `power`.  Forward and backward share this function so that their hit decisions agree bit for bit.
__device__ __forceinline__ float gauss_exponent(f2 d, f2 sAC, float sB, float lop, f2& u, float& hs) {
    const f2 t = splat(sB) * d.yx;
    u = pk_fma(sAC, d, t);
    hs = fmaf(u.x, d.x, u.y * d.y);       // (spelled out: every kernel that evaluates a pair must round it the same way)
    return fmaf(0.5f, hs, lop);
}
#define VCR_L2E 1.4426950408889634f
#define VCR_LN2 0.6931471805599453f
#define VCR_LOG2_255 7.994353436858858f

}  // namespace
