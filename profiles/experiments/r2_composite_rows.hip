// K6 in "row-stream" form: the four 16-lane DPP rows of a wave shade FOUR DIFFERENT Gaussians at a time, each on its own
// 4x4 pixel block of the wave's 8x8 quad.
//
// Why: in composite.hip every survivor of the quad-level culling is evaluated on all 64 pixels of the
// quad, but a Gaussian of the metric workload reaches 11 of them (1.9 of the four 4x4 blocks).  Here the survivors of
// several 64-entry chunks are first compacted into a 128-slot LDS buffer, then tested against the four blocks' live
// rectangles by 64 lanes at once (one survivor per lane), which yields one slot list per block; the drain loop then runs
// max_r(len_r) iterations in which row r shades the r-th block with ITS next survivor.  Per-pixel blending order is the list
// order within each block, i.e. exactly the order of composite.hip, and both kernels share gauss_exponent(), so the
// results (hit decisions included) are identical.
#include "vcr_common.h"
#include "composite_math.h"
#include <stdlib.h>

namespace {

#ifdef VCR_ROWS_DEBUG
__device__ unsigned long long g_rows_dbg[8];       // [0] drain iterations, [1] compacted survivors, [2] drains, [3] block entries, [4] chunks
#endif
constexpr int RS_SLOTS = 128;            // compacted survivors per batch (a batch is drained once it holds >= 64)

struct RowPixel { int x, y, pix; bool inside; };

// lane -> pixel: DPP row r = lane >> 4 owns the 4x4 block (r & 1, r >> 1) of the quad, lane & 15 = 4 * y + x inside it
__device__ __forceinline__ RowPixel row_pixel(int tile, int quad, int gx, int W, int H) {
    const int lane = threadIdx.x & 63, r = lane >> 4, j = lane & 15;
    RowPixel p;
    p.x = (tile % gx) * VCR_TILE + (quad & 1) * 8 + (r & 1) * 4 + (j & 3);
    p.y = (tile / gx) * VCR_TILE + (quad >> 1) * 8 + (r >> 1) * 4 + (j >> 2);
    p.inside = p.x < W && p.y < H;
    p.pix = p.y * W + p.x;
    return p;
}

// bounding box (block-local pixel units, inclusive) of the live pixels of one block; false if none is live
__device__ __forceinline__ bool block_box(unsigned m16, int& x0, int& x1, int& y0, int& y1) {
    if (m16 == 0) return false;
    const unsigned cols = (m16 | (m16 >> 4) | (m16 >> 8) | (m16 >> 12)) & 0xFu;
    x0 = __builtin_ctz(cols); x1 = 31 - __builtin_clz(cols);
    y0 = __builtin_ctz(m16) >> 2; y1 = (31 - __builtin_clz(m16)) >> 2;
    return true;
}

// Can the staged (pre-scaled) Gaussian reach alpha >= 1/255 on the rectangle [X0,X0+bw]x[Y0,Y0+bh]?  Same test as
// quad_touch() in base-2 units: p0 = (gx, gy, sA, sC), sB, lop with s = -log2(e) * conic, lop = log2(opacity).
__device__ __forceinline__ bool block_touch(const float4 p0, float sB, float lop, float X0, float Y0, float bw, float bh) {
    const float A = -p0.z, C = -p0.w, B = -sB;
    const float tau = lop + VCR_LOG2_255;
    if (!(tau >= 0.f)) return false;
    if (!(A > 0.f) || !(C > 0.f)) return true;
    const float x0 = X0 - p0.x, x1 = X0 + bw - p0.x, y0 = Y0 - p0.y, y1 = Y0 + bh - p0.y;
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return true;
    const float ia = fast_rcp(A), ic = fast_rcp(C);
    float qm = edge_min(A, C, B, x0, ic, y0, y1);
    qm = fminf(qm, edge_min(A, C, B, x1, ic, y0, y1));
    qm = fminf(qm, edge_min(C, A, B, y0, ia, x0, x1));
    qm = fminf(qm, edge_min(C, A, B, y1, ia, x0, x1));
    const float mx = fmaxf(x0 * x0, x1 * x1), my = fmaxf(y0 * y0, y1 * y1);
    return qm <= tau + 0.05f * VCR_L2E + 4e-6f * (A * mx + C * my);
}

#define VCR_LOAD_ID(POS, END, ID, VALID) \
    do { const uint32_t _p = (POS); VALID = _p < (END); ID = VALID ? point_list[_p] : 0u; } while (0)
#define VCR_GATHER_REC(ID, Q0, Q1, Q2, Q3)                                          \
    do {                                                                            \
        const float4* _src = reinterpret_cast<const float4*>(rec + (ID));           \
        Q0 = _src[0]; Q1 = _src[1]; Q2 = _src[2]; Q3 = _src[3];                      \
    } while (0)

template <int S, bool ISECT, int ND>
__global__ void __launch_bounds__(64) composite_fwd_rows_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                                const float* __restrict__ semv,
                                                                const uint32_t* __restrict__ point_list,
                                                                const uint2* __restrict__ ranges,
                                                                const uint32_t* __restrict__ tile_order, int num_tiles,
                                                                float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                                float* __restrict__ moments, float* __restrict__ out) {
    __shared__ float4 srec[4][RS_SLOTS];                  // compacted survivors: 4 planes x 128 slots x 16 B
    __shared__ uint8_t slist[4][RS_SLOTS];                // per block: the slots of the survivors that reach it, in list order
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    // workgroup -> (tile, quad): consecutive workgroups go to different XCDs (8, each with its own L2), so the four quads
    // of a tile are given to the SAME XCD: xcd = b & 7 takes the tiles order[8 i + xcd], four consecutive slots each
    const uint32_t slot_x = blockIdx.x >> 3, tidx = (slot_x >> 2) * 8u + (blockIdx.x & 7u);
    if (tidx >= (uint32_t)num_tiles) return;
    const int tile = (int)tile_order[tidx], quad = (int)(slot_x & 3u);
    const RowPixel pm = row_pixel(tile, quad, gx, a.W, a.H);
    const uint2 range = ranges[tile];
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63, row = lane >> 4;
    const float QX = (float)((tile % gx) * VCR_TILE + (quad & 1) * 8), QY = (float)((tile / gx) * VCR_TILE + (quad >> 1) * 8);
    const f2 fxy = {(float)pm.x, (float)pm.y};
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float T = 1.f;
    f2 acc_c01 = {0.f, 0.f}, acc_c2n = {0.f, 0.f}, acc_n12 = {0.f, 0.f}, acc_da = {0.f, 0.f};
    float SM[S > 0 ? S : 1];
#pragma unroll
    for (int k = 0; k < S; ++k) SM[k] = 0.f;
    float M1 = 0.f, M2 = 0.f;
    const float zc_map = VCR_ZFAR / (VCR_ZFAR - VCR_ZNEAR);
    uint32_t last = 0;
    bool done = !pm.inside;

    // live rectangles: per block (for the block-level test at drain time) and their union (quad-level test per chunk);
    // `done` only changes inside a drain, so they are recomputed after every drain
    float bX0[4], bY0[4], bW[4], bH[4];
    bool bLive[4];
    float qx0, qy0, qw, qh;
    bool anyLive;
    auto boxes = [&]() {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(!done);
        int ux0 = 8, ux1 = -1, uy0 = 8, uy1 = -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int x0 = 0, x1 = 3, y0 = 0, y1 = 3;
            bLive[r] = block_box((unsigned)(live >> (16 * r)) & 0xFFFFu, x0, x1, y0, y1);
            const int ox = (r & 1) * 4, oy = (r >> 1) * 4;
            bX0[r] = QX + (float)(ox + x0); bY0[r] = QY + (float)(oy + y0); bW[r] = (float)(x1 - x0); bH[r] = (float)(y1 - y0);
            if (bLive[r]) { ux0 = min(ux0, ox + x0); ux1 = max(ux1, ox + x1); uy0 = min(uy0, oy + y0); uy1 = max(uy1, oy + y1); }
        }
        anyLive = live != 0;
        qx0 = QX + (float)ux0; qy0 = QY + (float)uy0; qw = (float)(ux1 - ux0); qh = (float)(uy1 - uy0);
    };
    boxes();

    uint32_t used = 0;                                     // slots of the batch in use (wave-uniform)
    uint32_t pos = range.x;
    uint32_t id, nid; float4 q0, q1, q2, q3; bool valid, nvalid;
    VCR_LOAD_ID(pos + lane, range.y, id, valid);
    VCR_GATHER_REC(id, q0, q1, q2, q3);
    VCR_LOAD_ID(pos + 64 + lane, range.y, nid, nvalid);
    while (pos < range.y && anyLive) {
        uint32_t nnid; float4 nq0, nq1, nq2, nq3; bool nnvalid;
        const uint32_t npos = pos + 64;
        VCR_GATHER_REC(nid, nq0, nq1, nq2, nq3);                     // records of the next chunk
        VCR_LOAD_ID(npos + 64 + lane, range.y, nnid, nnvalid);       // ids of the chunk after that
        const bool keep = valid && quad_touch(q0, q1, qx0, qy0, qw, qh);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {                                        // compacted staging, pre-scaled for the shading loop
            const uint32_t slot = used + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            srec[0][slot] = make_float4(q0.x, q0.y, -VCR_L2E * q1.x, -VCR_L2E * q1.z);
            srec[1][slot] = make_float4(-VCR_L2E * q1.y, __builtin_amdgcn_logf(q0.w), q0.z, q1.w);
            srec[2][slot] = make_float4(q2.x, q2.y, q2.z, q3.x);
            srec[3][slot] = make_float4(q3.y, q3.z, __uint_as_float(id), __uint_as_float(pos - range.x + (uint32_t)lane + 1u));
        }
        used += (uint32_t)__popcll(m);
#ifdef VCR_ROWS_DEBUG
        if (lane == 0) atomicAdd(&g_rows_dbg[4], 1ull);
#endif
        if (used >= 64u || npos >= range.y) {
            if (used > 0) {
                // ---- block-level test of the compacted survivors: one survivor per lane, four blocks each ----
                __builtin_amdgcn_wave_barrier();
                uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
                for (uint32_t base = 0; base < used; base += 64u) {
                    const uint32_t s = base + (uint32_t)lane;
                    const bool sv = s < used;
                    const float4 p0 = srec[0][s & (RS_SLOTS - 1)], p1 = srec[1][s & (RS_SLOTS - 1)];
#define VCR_BLOCK(R, CNT)                                                                                            \
                    {                                                                                                \
                        const bool t = sv && bLive[R] && block_touch(p0, p1.x, p1.y, bX0[R], bY0[R], bW[R], bH[R]);   \
                        const unsigned long long bm = __builtin_amdgcn_ballot_w64(t);                                \
                        if (t) slist[R][(CNT + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),                        \
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))) & (RS_SLOTS - 1)] = (uint8_t)s; \
                        CNT += (uint32_t)__popcll(bm);                                                               \
                    }
                    VCR_BLOCK(0, cnt0) VCR_BLOCK(1, cnt1) VCR_BLOCK(2, cnt2) VCR_BLOCK(3, cnt3)
#undef VCR_BLOCK
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t n = max(max(cnt0, cnt1), max(cnt2, cnt3));
#ifdef VCR_ROWS_DEBUG
                if (lane == 0) {
                    atomicAdd(&g_rows_dbg[0], (unsigned long long)n); atomicAdd(&g_rows_dbg[1], (unsigned long long)used);
                    atomicAdd(&g_rows_dbg[2], 1ull); atomicAdd(&g_rows_dbg[3], (unsigned long long)(cnt0 + cnt1 + cnt2 + cnt3));
                }
#endif
                // tails of the lists point at slot 0 (a valid record of this batch): rows that have run out of survivors keep
                // reading finite data while their lanes are masked
                for (uint32_t i = (uint32_t)lane; i < (uint32_t)RS_SLOTS; i += 64u) {
                    if (i >= cnt0) slist[0][i] = 0; if (i >= cnt1) slist[1][i] = 0;
                    if (i >= cnt2) slist[2][i] = 0; if (i >= cnt3) slist[3][i] = 0;
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t cntrow = row == 0 ? cnt0 : (row == 1 ? cnt1 : (row == 2 ? cnt2 : cnt3));
                const uint8_t* const lst = slist[row];
                // ---- drain: iteration k, row r shades its block with its k-th survivor ----
#define VCR_FETCH(R, SLOT)                                                                       \
                do {                                                                             \
                    const uint32_t _s = (SLOT) & (RS_SLOTS - 1);                                 \
                    R##0 = srec[0][_s]; R##1 = srec[1][_s]; R##2 = srec[2][_s]; R##3 = srec[3][_s]; \
                    asm volatile("" ::: "memory");                                               \
                } while (0)
#define VCR_SHADE(R, K)                                                                                                  \
                do {                                                                                                     \
                    const float4 r0 = R##0, r1 = R##1, r2 = R##2, r3 = R##3;                                             \
                    const f2 gxy = {r0.x, r0.y}, sAC = {r0.z, r0.w};                                                     \
                    f2 u; float hs;                                                                                      \
                    const float e = gauss_exponent(gxy - fxy, sAC, r1.x, r1.y, u, hs);                                   \
                    const float alpha = fminf(VCR_ALPHA_MAX, __builtin_amdgcn_exp2f(e));                                 \
                    bool hit = (K) < cntrow && !done && hs <= 0.f && alpha >= VCR_ALPHA_MIN;                             \
                    const float test_T = fmaf(-alpha, T, T);                                                             \
                    if (hit && test_T < VCR_T_EPS) { done = true; hit = false; }                                         \
                    const float w = hit ? alpha * T : 0.f;                                                               \
                    const f2 c01 = {r2.x, r2.y}, c2n = {r2.z, r2.w}, n12 = {r3.x, r3.y};                                 \
                    float dep = r1.z;                                                                                    \
                    if (ISECT) {                                                                                         \
                        const float den = c2n.y * rx + n12.x * ry + n12.y * rz;                                          \
                        if (den > VCR_PLANE_EPS) dep = r1.w * fast_rcp(den) * rz;                                        \
                    }                                                                                                    \
                    const f2 ww = splat(w);                                                                              \
                    acc_c01 = pk_fma(ww, c01, acc_c01);                                                                  \
                    acc_c2n = pk_fma(ww, c2n, acc_c2n);                                                                  \
                    acc_n12 = pk_fma(ww, n12, acc_n12);                                                                  \
                    acc_da = pk_fma(ww, f2{dep, 1.f}, acc_da);                                                           \
                    if (ND == 2) M2 += hit ? w * dep * dep : 0.f;                                                        \
                    if (ND == 1) {                                                                                       \
                        const float md = hit ? -zc_map * VCR_ZNEAR * fast_rcp(dep) : 0.f;                                \
                        M1 += w * md; M2 += w * md * md;                                                                 \
                    }                                                                                                    \
                    if (S > 0) {                                                                                         \
                        const uint32_t gid = __float_as_uint(r3.z);                                                      \
_Pragma("unroll")                                                                                                        \
                        for (int k = 0; k < S; ++k) SM[k] += w * (hit ? semv[(size_t)gid * S + k] : 0.f);                \
                    }                                                                                                    \
                    T = hit ? test_T : T;                                                                                \
                    last = hit ? __float_as_uint(r3.w) : last;                                                           \
                } while (0)
                if (n > 0) {
                    float4 A0, A1, A2, A3, B0, B1, B2, B3;
                    uint32_t s_next = lst[1];
                    VCR_FETCH(A, (uint32_t)lst[0]);
                    uint32_t k = 0;
                    for (;;) {                               // ping-pong: the other buffer is in flight while one is shaded
                        uint32_t s_n2 = lst[(k + 2) & (RS_SLOTS - 1)];
                        VCR_FETCH(B, s_next);
                        VCR_SHADE(A, k);
                        if (++k >= n) break;
                        s_next = lst[(k + 2) & (RS_SLOTS - 1)];
                        VCR_FETCH(A, s_n2);
                        VCR_SHADE(B, k);
                        if (++k >= n) break;
                    }
                }
#undef VCR_FETCH
#undef VCR_SHADE
                used = 0;
                boxes();
            }
        }
        pos = npos; id = nid; q0 = nq0; q1 = nq1; q2 = nq2; q3 = nq3; valid = nvalid; nid = nnid; nvalid = nnvalid;
    }
    const float C0 = acc_c01.x, C1 = acc_c01.y, C2 = acc_c2n.x, N0 = acc_c2n.y, N1 = acc_n12.x, N2 = acc_n12.y;
    const float D = acc_da.x, A = acc_da.y;
    if (pm.inside) {
        final_T[pm.pix] = T;
        n_contrib[pm.pix] = last;
        out[0 * (size_t)P + pm.pix] = C0 + T * a.bg[0];
        out[1 * (size_t)P + pm.pix] = C1 + T * a.bg[1];
        out[2 * (size_t)P + pm.pix] = C2 + T * a.bg[2];
        out[3 * (size_t)P + pm.pix] = D;
        out[4 * (size_t)P + pm.pix] = N0;
        out[5 * (size_t)P + pm.pix] = N1;
        out[6 * (size_t)P + pm.pix] = N2;
        out[7 * (size_t)P + pm.pix] = A;
#pragma unroll
        for (int k = 0; k < S; ++k) out[(8 + k) * (size_t)P + pm.pix] = SM[k];
        if (ND == 2) {
            out[(8 + S) * (size_t)P + pm.pix] = D;
            out[(9 + S) * (size_t)P + pm.pix] = M2;
        }
        if (ND == 1) {
            out[(8 + S) * (size_t)P + pm.pix] = A * M2 - M1 * M1;
            moments[pm.pix] = M1; moments[P + pm.pix] = M2;
        }
    }
}

template <bool ISECT, int ND>
int launch_rows_fwd(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o, int tiles, hipStream_t st) {
#define VCR_FWDR(SS)                                                                                                   \
    hipLaunchKernelGGL((composite_fwd_rows_kernel<SS, ISECT, ND>), dim3(32 * ((tiles + 7) / 8)), dim3(64), 0, st, a, g.rec, g.sem,   \
                       b.point_list, b.ranges, b.tile_order, tiles, im.final_T, im.n_contrib, im.moments, o.out)
    switch (a.S) {
        case 0: VCR_FWDR(0); break;
        case 1: VCR_FWDR(1); break;
        case 2: VCR_FWDR(2); break;
        case 3: VCR_FWDR(3); break;
        default: VCR_FWDR(4); break;
    }
#undef VCR_FWDR
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// f_count == 0 only (the count / visibility modes keep the per-wave kernel of composite.hip)
int vcr_launch_composite_forward_rows(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o,
                                      hipStream_t st) {
    const int tiles = ((a.W + VCR_TILE - 1) / VCR_TILE) * ((a.H + VCR_TILE - 1) / VCR_TILE);
    const bool isect = a.dirs != nullptr && a.normals_precomp != nullptr;
    if (a.num_dist == 2) return isect ? launch_rows_fwd<true, 2>(a, g, b, im, o, tiles, st) : launch_rows_fwd<false, 2>(a, g, b, im, o, tiles, st);
    if (a.num_dist == 1) return isect ? launch_rows_fwd<true, 1>(a, g, b, im, o, tiles, st) : launch_rows_fwd<false, 1>(a, g, b, im, o, tiles, st);
    return isect ? launch_rows_fwd<true, 0>(a, g, b, im, o, tiles, st) : launch_rows_fwd<false, 0>(a, g, b, im, o, tiles, st);
}

#ifdef VCR_ROWS_DEBUG
extern "C" int vcr_rows_debug_read(unsigned long long* out8, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rows_dbg), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_rows_dbg), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif
