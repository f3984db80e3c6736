// K6 / K7: per-tile front-to-back alpha compositing and its adjoint for gfx950 (wave64).
//
// Contract: SURVEY.md Appendix A.3 step 9-10 / A.4, restated in oracle/raster_torch.py::composite_tile.
// Output channel layout consumed by gaussian_renderer/__init__.py:122-123,150,155-161:
//   colour3 | depth1 | camera-space normal3 | alpha1 | semantics S.
//
// Geometry: one 256-thread workgroup per 16x16 tile; each 64-lane wave owns one 8x8 pixel quad and runs
// independently, so the early-out (T < 1e-4) and the "nobody in this wave is touched" skip are wave-uniform.
// (A first LDS-staged, barrier-synchronised per-tile version of these kernels measured 1.55 / 1.87 ms fwd/bwd on
// the metric workload against 0.42 / 0.79 ms for this design)
#include "vcr_common.h"
#include "composite_math.h"
#include <stdlib.h>

namespace {

// Reduce 16 per-lane values (8 pairs) over the wave with a halving butterfly: v_permlane32_swap / v_permlane16_swap
// exchange half of the live values per step (8+4 swaps, the sums as packed adds), then 4 row rotations finish the
// remaining 4.  On return lane l holds, in out[0..3], the wave totals of slots 8*(l>>5) + 4*((l>>4)&1) + 0..3
// (slot 2j = v[j].x, slot 2j+1 = v[j].y).
__device__ __forceinline__ void wave_reduce16(const f2 v[8], float out[4]) {
    f2 u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j].x), __float_as_uint(v[4 + j].x), false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j].y), __float_as_uint(v[4 + j].y), false, false);
        u[j] = f2{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + f2{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[j].x), __float_as_uint(u[2 + j].x), false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[j].y), __float_as_uint(u[2 + j].y), false, false);
        const f2 x = f2{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + f2{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
        out[2 * j] = row_sum16(x.x);
        out[2 * j + 1] = row_sum16(x.y);
    }
}

struct PixelMap {
    int x, y, pix;
    bool inside;
};

// `sub` < 0: the wave owns the whole 8x8 quad `wv` of the tile.  sub = 0..3 ("split" work items of the heaviest tiles): it
// owns only the 4x4 sub-block (sub & 1, sub >> 1) of that quad -- lanes keep the 8x8 numbering relative to the sub-block's
// origin and the 48 lanes outside it are masked like out-of-image pixels, so the culling rectangle (bounding box of the
// live lanes) shrinks to the sub-block by itself.  Four waves then share a quad's list: 1.9x its shading work, but less
// than half its serial length, which is what bounds the launch.
__device__ __forceinline__ PixelMap pixel_of_quad(int tile, int gx, int W, int H, int sub, int wv) {
    const int lane = threadIdx.x & 63;
    PixelMap p;
    const int ox = sub < 0 ? 0 : (sub & 1) * 4, oy = sub < 0 ? 0 : (sub >> 1) * 4;
    p.x = (tile % gx) * VCR_TILE + (wv & 1) * 8 + ox + (lane & 7);
    p.y = (tile / gx) * VCR_TILE + (wv >> 1) * 8 + oy + (lane >> 3);
    p.inside = p.x < W && p.y < H && (sub < 0 || ((lane & 7) < 4 && (lane >> 3) < 4));
    p.pix = p.y * W + p.x;
    return p;
}
__device__ __forceinline__ PixelMap pixel_of_thread(int tile, int gx, int W, int H, int sub) {
    return pixel_of_quad(tile, gx, W, H, sub, (int)(threadIdx.x >> 6));
}

// blockIdx -> (tile, sub): the first 4 * S blocks are the split items of the S heaviest tiles (tile_order is longest-first,
// S = meta[0] is decided on the device by tile_order_kernel), then one block per remaining tile; the grid is sized for the
// largest S, surplus blocks return -1.
__device__ __forceinline__ int work_item(const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ meta, int num_tiles,
                                         int& sub) {
    const int b = (int)blockIdx.x, n_split = (int)meta[0];
    if (b < 4 * n_split) { sub = b & 3; return (int)tile_order[b >> 2]; }
    sub = -1;
    const int i = b - 3 * n_split;
    return i < num_tiles ? (int)tile_order[i] : -1;
}

// [begin, end) of the list a wave walks: the tile's (all four quad-waves share it), or -- quad-list mode, gxc = 8x8 cells per
// row (a kernel argument: the host knows the mode of the call; reading it from BinState::meta would put a second dependent
// load in front of every wave, 7 us per launch with 32 k mostly empty waves) -- the list of the wave's own quad.
template <bool QL>       // (compile-time: the per-tile kernels keep their one `ranges[tile]` load and nothing else -- a run-time
__device__ __forceinline__ uint2 list_range(const uint2* __restrict__ ranges, int gxc, int tile, int gx, int quad) {   // branch here
    if (!QL) return ranges[tile];                                                          // measured +7 us per forward launch)
    return ranges[(2 * (tile / gx) + (quad >> 1)) * gxc + 2 * (tile % gx) + (quad & 1)];
}

// Transmittance checkpoints (round 5).  The forward stores every pixel's T at the START of each 64-entry chunk c >= 1 of the list
// it walks (c = 0 starts at T = 1); the backward, which recovers T back to front by T_i = T_{i+1} / (1 - alpha_i), replaces its
// recovered value by the forward's own at every chunk boundary, so the rounding of that division chain is bounded by the few
// contributors a pixel has inside ONE chunk instead of growing over the whole list (profiles/r5_bwd_algorithm_emulation.txt:
// a +-1 ulp reciprocal in an un-anchored chain doubles the gradient error of an fp32 evaluation; anchored it is back at 1.0x).
// Record of chunk c of list l (tile, or 8x8 cell in quad-list mode) = slot (begin_l >> 6) + c + l: lists are laid out in index
// order, begin_{l'} >= begin_l + len_l and l' >= l + 1 for a later non-empty list, so floor(begin_{l'} / 64) + l' >=
// floor(begin_l / 64) + ceil(len_l / 64) + l -- slots never collide and stay below E / 64 + lists.  A slot holds the 256 pixels
// of the tile (64 per quad wave, quad-local row-major) or, in quad-list mode, the 64 pixels of the cell.
template <bool QL>
__device__ __forceinline__ size_t ckpt_base(uint32_t begin, int gxc, int tile, int gx, int quad) {
    if (!QL) return ((size_t)(begin >> 6) + (size_t)tile) * 256u + (size_t)quad * 64u;
    return ((size_t)(begin >> 6) + (size_t)((2 * (tile / gx) + (quad >> 1)) * gxc + 2 * (tile % gx) + (quad & 1))) * 64u;
}
#define VCR_CKPT_STRIDE (QL ? 64u : 256u)

// ================= one wave = one 8x8 quad, no LDS, no barriers ==========================================
// Each lane first acts as a CULLER for one Gaussian of the tile list (exact minimum of the conic form over
// the quad's pixel rectangle against the alpha >= 1/255 threshold), a 64-bit ballot compacts the survivors,
// and the wave then walks the set bits in depth order; the survivor's record is broadcast from its owner
// lane with v_readlane (SGPR operands for the per-pixel VALU work).  The next 64 records are gathered from
// HBM/L2 while the current survivors are shaded.

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Bounding box (in quad-local pixel units) of the lanes set in `live` (lane = 8*y + x): the culling rectangle
// shrinks to the pixels that can still change -- once most of a quad has saturated (T < 1e-4) a long list is only
// walked for the Gaussians that reach the few remaining pixels, which is what bounds the slowest wave of a launch.
__device__ __forceinline__ void live_box(unsigned long long live, float X0, float Y0, float& bx0, float& by0, float& bw,
                                         float& bh) {
    if (live == 0) { bx0 = X0; by0 = Y0; bw = 7.f; bh = 7.f; return; }
    const int y0 = __builtin_ctzll(live) >> 3, y1 = (63 - __builtin_clzll(live)) >> 3;
    unsigned long long c = live | (live >> 32);
    c |= c >> 16; c |= c >> 8;
    const unsigned cols = (unsigned)c & 0xFFu;
    const int x0 = __builtin_ctz(cols), x1 = 31 - __builtin_clz(cols);
    bx0 = X0 + (float)x0; by0 = Y0 + (float)y0; bw = (float)(x1 - x0); bh = (float)(y1 - y0);
}

// Two-stage software pipeline over 64-entry chunks of the tile list: the id of chunk k+2 is loaded while
// the 64-byte records of chunk k+1 are gathered and chunk k is shaded.  Plain registers only (no structs), and
// out-of-range lanes read record 0 (never used) so that no select-of-pointers / scratch is generated.
#define VCR_LOAD_ID(POS, END, ID, VALID) \
    do { const uint32_t _p = (POS); VALID = _p < (END); ID = VALID ? point_list[_p] : 0u; } while (0)
// semantic features of the entry (S <= 4 floats from semv[N,S]); staged with the record so that the shading loops read them
// from LDS instead of issuing a dependent global load per survivor (S = 2: forward 0.55 -> ms)
#define VCR_GATHER_SEM(ID, QS)                                                                   \
    do {                                                                                         \
        if (S > 0) {                                                                             \
            const float* _sp = semv + (size_t)(ID) * S;                                          \
            QS.x = _sp[0]; QS.y = S > 1 ? _sp[1] : 0.f; QS.z = S > 2 ? _sp[2] : 0.f; QS.w = S > 3 ? _sp[3] : 0.f; \
        }                                                                                        \
    } while (0)
#define VCR_GATHER_REC(ID, Q0, Q1, Q2, Q3)                                          \
    do {                                                                            \
        const float4* _src = reinterpret_cast<const float4*>(rec + (ID));           \
        Q0 = _src[0]; Q1 = _src[1]; Q2 = _src[2]; Q3 = _src[3];                      \
    } while (0)

// Knock-out builds for bottleneck attribution (timing only, results are wrong): -DVCR_KO=1 skips the shading loops
// (gather + culling + staging remain), =2 keeps the backward's wave reduction but never issues its atomics, =4 issues
// the atomics without the reduction.  See DESIGN.md section 4.
#ifndef VCR_KO
#define VCR_KO 0
#endif
// Diagnostic builds for the gradient-error attribution of round 5 (profiles/r5_grad_ratio_table_*.txt; results are as valid
// as the default build's, only slower): -DVCR_DBG_PIX64 keeps the per-pixel recurrences of the backward (transmittance
// recovery, suffix sum, dL/dalpha) in fp64; -DVCR_DBG_ACC64 accumulates the per-Gaussian screen-space sums with fp64 atomics
// into a side buffer that is rounded to the GradRec once.
#ifdef VCR_DBG_ACC64
__device__ double* g_acc64 = nullptr;
#define VCR_GRAD_ATOMIC(GID, K, VAL) atomicAdd(g_acc64 + (size_t)(GID) * 16 + (K), (double)(VAL))
#else
#define VCR_GRAD_ATOMIC(GID, K, VAL) atomicAdd(reinterpret_cast<float*>(sgrad + (GID)) + (K), (VAL))
#endif
#ifndef VCR_DBG_PIX64
#define VCR_DBG_PIX64 0
#endif
// -DVCR_HITHIST: instrumented build that counts, per surviving (quad, Gaussian) pair, how many of the 64 pixels it hits
// (forward: bins 0..64, backward: 65..129); read with vcr_debug_hit_histogram (profiles/hit_histogram.py)
#ifdef VCR_HITHIST
__device__ unsigned int g_hithist[130];
#define VCR_COUNT_HITS(BASE, MASK) do { if (lane == 0) atomicAdd(&g_hithist[(BASE) + __popcll(MASK)], 1u); } while (0)
#else
#define VCR_COUNT_HITS(BASE, MASK) do { } while (0)
#endif
// -DVCR_TPSTATS: instrumented build of the two-phase forward -- per-wave shader-clock sums of its three parts and its work counts,
// read through vcr_debug_hit_histogram (out[2k], out[2k+1] = low / high word of counter k; profiles/r6_tp_stats.py):
//   0 waves with a non-empty list, 1 cycles in total, 2 culling + staging, 3 phase 1, 4 phase 2, 5 flushes, 6 phase-2 iterations,
//   7 survivors staged, 8 chunks, 9 candidates (bits handed to the pixel lanes), 10 longest wave (cycles), 11 hits
#ifdef VCR_TPSTATS
__device__ unsigned long long g_tpstats[16];
#define VCR_TPS(X) X
#else
#define VCR_TPS(X)
#endif
// Survivors that hit at most this many pixels of the quad skip the 16-value wave reduction of the backward: their few lanes
// add their 16 values to the GradRec directly (16 masked atomic instructions).  0 disables the path -- the default:
// measured with 2 on the metric workload (28 % of the survivors hit <= 2 pixels, profiles/r3_hit_histogram_metric.txt):
// compositing backward 0.76 ms against 0.51 -- sixteen atomic INSTRUCTIONS cost several times the ~195 SIMD cycles of the
// butterfly they replace (the atomics are free in bandwidth, not in issue).
#ifndef VCR_BWD_SPARSE_HITS
#define VCR_BWD_SPARSE_HITS 0
#endif
// -DVCR_T_ANCHOR=1 (vcr_common.h): per-chunk re-anchoring of the recovered transmittance on checkpoints the forward stores;
// -DVCR_RCP_NEWTON=1: one Newton step on the v_rcp_f32 of the recovery.  Both were built and A/B'd in round 5 (VERDICT r4 item
// 2) and are OFF: neither moves the gradient error (profiles/r5_grad_ratio_*.txt), because the excess error of rounds 1-4
// was not made in this kernel at all (profiles/r5_grad_stage_errors.txt) -- and the checkpoints cost 10 + 20 us per step.
#ifndef VCR_RCP_NEWTON
#define VCR_RCP_NEWTON 0
#endif


#ifndef VCR_ROWS_WAVES
#define VCR_ROWS_WAVES 3         // row-packed backward, waves per SIMD: 3 -> 0.388 ms at the metric scene, 4 -> 0.399, 5 -> 0.47 (spills);
                                 // c5 0.893 / 0.960, dense 0.551 / 0.570 for 3 / 4 (profiles/r3_ab_waves.sh)
#endif
#ifndef VCR_FWD_WAVES
#define VCR_FWD_WAVES 0          // 0: compiler's choice (79 VGPRs, 6 waves per SIMD)
#endif
#if VCR_FWD_WAVES > 0
#define VCR_FWD_ATTR __attribute__((amdgpu_waves_per_eu(VCR_FWD_WAVES)))
#else
#define VCR_FWD_ATTR
#endif
// per wave: 4 (+1 with semantics outside the record) planes x 64 slots x 16 B  (conflict-free b128 writes)
#define VCR_V2_WREC ((S > 0 && !(S <= 2 && FC == 0)) ? 320 : 256)
// One work item of the uniform-loop forward: quad `wv` of `tile` (split items: 4x4 sub-block `sub`); `srec`: the wave's LDS planes.
template <int S, bool ISECT, int FC, int ND, bool QL>
__device__ __forceinline__ void fwd_v2_item(const VcrRasterArgs& a, const GeomRec* __restrict__ rec, const float* __restrict__ semv,
                                            const uint32_t* __restrict__ point_list, const uint2* __restrict__ ranges, int gxc,
                                            float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                            float* __restrict__ moments, float* __restrict__ out,
                                            int32_t* __restrict__ count, float* __restrict__ score, float* __restrict__ ckpt,
                                            int tile, int sub, int wv, float4* const srec) {
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const PixelMap pm = pixel_of_quad(tile, gx, a.W, a.H, sub, wv);
    const uint2 range = list_range<QL>(ranges, gxc, tile, gx, wv);
    // this lane's slot in a checkpoint record: quad-local row-major pixel index
    float* const ck = FC == 0 ? ckpt + ckpt_base<QL>(range.x, gxc, tile, gx, wv) +
                                    (size_t)((pm.y & 7) * 8 + (pm.x & 7)) : nullptr;
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63;
    // S <= 2 without count mode: the semantic features come with the record (GeomRec pad slots) and take the place of the id
    constexpr bool SEM_IN_REC = S > 0 && S <= 2 && FC == 0;
    const float X0 = (float)((tile % gx) * VCR_TILE + (wv & 1) * 8 + (sub < 0 ? 0 : (sub & 1) * 4));
    const float Y0 = (float)((tile / gx) * VCR_TILE + (wv >> 1) * 8 + (sub < 0 ? 0 : (sub >> 1) * 4));
    const f2 fxy = {(float)pm.x, (float)pm.y};
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float T = 1.f;
    f2 acc_c01 = {0.f, 0.f}, acc_c2n = {0.f, 0.f}, acc_n12 = {0.f, 0.f}, acc_da = {0.f, 0.f};   // (C0,C1) (C2,N0) (N1,N2) (D,A)
    float SM[S > 0 ? S : 1];
#pragma unroll
    for (int k = 0; k < S; ++k) SM[k] = 0.f;
    float M1 = 0.f, M2 = 0.f;   // ND == 2: sum w d^2 (sum w d is the depth channel); ND == 1: sum w m, sum w m^2
    const float zc_map = VCR_ZFAR / (VCR_ZFAR - VCR_ZNEAR);
    uint32_t last = 0;
    bool done = !pm.inside;

    uint32_t pos = range.x;
    // count-only modes (FC >= 3) shade no channel: they gather and stage the first 32 bytes of a record only
#define VCR_GATHER_FWD(ID, Q0, Q1, Q2, Q3)                                          \
    do {                                                                            \
        const float4* _src = reinterpret_cast<const float4*>(rec + (ID));           \
        Q0 = _src[0]; Q1 = _src[1];                                                 \
        if (FC < 3) { Q2 = _src[2]; Q3 = _src[3]; }                                 \
    } while (0)
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    uint32_t id, nid; float4 q0, q1, q2 = zero4, q3 = zero4, qs = zero4; bool valid, nvalid;
    VCR_LOAD_ID(pos + lane, range.y, id, valid);
    VCR_GATHER_FWD(id, q0, q1, q2, q3);
    if (!SEM_IN_REC) VCR_GATHER_SEM(id, qs);
    VCR_LOAD_ID(pos + 64 + lane, range.y, nid, nvalid);
    while (pos < range.y) {
        uint32_t nnid; float4 nq0, nq1, nq2 = zero4, nq3 = zero4, nqs = zero4; bool nnvalid;
        const uint32_t npos = pos + 64;
        if (VCR_T_ANCHOR && FC == 0 && pos != range.x && pm.inside) ck[(size_t)((pos - range.x) >> 6) * VCR_CKPT_STRIDE] = T;
        VCR_GATHER_FWD(nid, nq0, nq1, nq2, nq3);                     // records of the next chunk
        if (!SEM_IN_REC) VCR_GATHER_SEM(nid, nqs);
        VCR_LOAD_ID(npos + 64 + lane, range.y, nnid, nnvalid);       // ids of the chunk after that
        float bx0, by0, bw, bh;
        live_box(__builtin_amdgcn_ballot_w64(!done), X0, Y0, bx0, by0, bw, bh);
        const bool keep = valid && quad_touch(q0, q1, bx0, by0, bw, bh);
        unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        int chunk_cnt = 0; unsigned long long chunk_seen = 0;       // FC 3 / 4 (see below)
        (void)chunk_cnt; (void)chunk_seen;
        // Survivors stage their record -- rescaled for the shading loop, see gauss_exponent() -- in this wave's private LDS
        // planes; the shading loop then fetches one survivor per iteration with four wave-uniform ds_read_b128 (LDS
        // broadcast reads, 4 CU cycles each) one survivor ahead.  v_readlane_b32 costs 8.3 SIMD cycles on gfx950
        // (profiles/microbench), so the 14 register broadcasts it replaces were half of this loop.
        if (keep) {
            srec[0 * 64 + lane] = make_float4(q0.x, q0.y, -VCR_L2E * q1.x, -VCR_L2E * q1.z);
            srec[1 * 64 + lane] = make_float4(-VCR_L2E * q1.y, __builtin_amdgcn_logf(q0.w), q0.z, q1.w);   // v_log_f32 = log2
            if (FC < 3) {
                srec[2 * 64 + lane] = make_float4(q2.x, q2.y, q2.z, q3.x);
                srec[3 * 64 + lane] = SEM_IN_REC ? make_float4(q3.y, q3.z, q2.w, q3.w) : make_float4(q3.y, q3.z, __uint_as_float(id), 0.f);
            }
            if (S > 0 && !SEM_IN_REC) srec[4 * 64 + lane] = qs;
        }
        __builtin_amdgcn_wave_barrier();          // same wave, DS ops execute in order: no s_barrier needed
        // Shading of one survivor whose staged record sits in R0..R3 (a macro, not a lambda: captured-by-reference bools
        // become byte-sized phis in VGPRs instead of lane masks in SGPRs).  No early skip when nobody is hit (8 % of
        // survivors): the shading is then a no-op with w = 0 and a single-block body spares the phi copies of the
        // accumulators; the "whole quad saturated" exit is tested once per chunk (below), not per survivor.
#define VCR_SHADE_FWD(R, B)                                                                                              \
        do {                                                                                                             \
            const float4 r0 = R##0, r1 = R##1, r2 = R##2, r3 = R##3; const int sb_ = (B);                                  \
            const f2 gxy = {r0.x, r0.y}, sAC = {r0.z, r0.w};                                                             \
            f2 u; float hs;                                                                                              \
            const float e = gauss_exponent(gxy - fxy, sAC, r1.x, r1.y, u, hs);                                           \
            const float alpha = fminf(VCR_ALPHA_MAX, __builtin_amdgcn_exp2f(e));                                         \
            bool hit = !done && hs <= 0.f && alpha >= VCR_ALPHA_MIN;                                                     \
            const float test_T = fmaf(-alpha, T, T);                                                                     \
            if (hit && test_T < VCR_T_EPS) { done = true; hit = false; }                                                 \
            const float w = hit ? alpha * T : 0.f;                                                                       \
            { const unsigned long long hm_ = __builtin_amdgcn_ballot_w64(hit); VCR_COUNT_HITS(0, hm_); (void)hm_; }      \
            if (FC == 1) {                                                                                               \
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);                                          \
                if (hm != 0) {                                                                                           \
                    const float ws = wave_sum(w);                                                                        \
                    if (lane == 0) {                                                                                     \
                        const uint32_t gid = __float_as_uint(r3.z);                                                      \
                        atomicAdd(count + gid, (int)__popcll(hm));                                                       \
                        atomicAdd(score + gid, ws);                                                                      \
                    }                                                                                                    \
                }                                                                                                        \
            }                                                                                                            \
            if (FC == 3) {      /* count only: the survivor's pixel count goes to ITS slot of the chunk (lane sb_), one atomic    */ \
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);   /* instruction per chunk deposits them all  */ \
                chunk_cnt = lane == sb_ ? (int)__popcll(hm) : chunk_cnt;                                                 \
            }                                                                                                            \
            if (FC == 4) chunk_seen |= (unsigned long long)(__builtin_amdgcn_ballot_w64(hit) != 0) << sb_;               \
            const f2 c01 = {r2.x, r2.y}, c2n = {r2.z, r2.w}, n12 = {r3.x, r3.y};                                         \
            float dep = r1.z;                                                                                            \
            if (ISECT) {                                                                                                 \
                const float den = fmaf(c2n.y, rx, fmaf(n12.x, ry, n12.y * rz));   /* (explicit: both copies of the macro must round alike) */                                                  \
                if (den > VCR_PLANE_EPS) dep = r1.w * fast_rcp(den) * rz;                                                \
            }                                                                                                            \
            const f2 ww = splat(w);                                                                                      \
            acc_c01 = pk_fma(ww, c01, acc_c01);                                                                          \
            acc_c2n = pk_fma(ww, c2n, acc_c2n);                                                                          \
            acc_n12 = pk_fma(ww, n12, acc_n12);                                                                          \
            acc_da = pk_fma(ww, f2{dep, 1.f}, acc_da);                                                                   \
            if (ND == 2) M2 = fmaf(w * dep, dep, M2);                                                                    \
            if (ND == 1) {                                                                                               \
                const float md = -zc_map * VCR_ZNEAR * fast_rcp(dep);                                                    \
                M1 = fmaf(w, md, M1); M2 = fmaf(w * md, md, M2);                                                         \
            }                                                                                                            \
            if (S > 0) {                                                                                                 \
                const float4 r4_ = SEM_IN_REC ? make_float4(r3.z, r3.w, 0.f, 0.f) : R##4;                                \
                const float sv_[4] = {r4_.x, r4_.y, r4_.z, r4_.w};                                                       \
_Pragma("unroll")                                                                                                        \
                for (int k = 0; k < S; ++k) SM[k] = fmaf(w, sv_[k], SM[k]);                                              \
            }                                                                                                            \
            T = hit ? test_T : T;                                                                                        \
            last = hit ? pos - range.x + (uint32_t)sb_ + 1u : last;                                                        \
        } while (0)
        // the empty asm (memory clobber) keeps the compiler from sinking the prefetch of the other buffer below the
        // shading of the current one; it does not wait for the data
#define VCR_LDS_FETCH(R, B)                                                                                   \
        do {                                                                                                  \
            R##0 = srec[(B)]; R##1 = srec[64 + (B)];                                                          \
            if (FC < 3) { R##2 = srec[128 + (B)]; R##3 = srec[192 + (B)]; }                                   \
            if (S > 0 && !SEM_IN_REC) R##4 = srec[256 + (B)];                                                 \
            asm volatile("" ::: "memory");                                                                    \
        } while (0)
        if (m && !(VCR_KO & 1)) {
            float4 A0, A1, A2 = zero4, A3 = zero4, A4 = zero4, B0, B1, B2 = zero4, B3 = zero4, B4 = zero4;
            int b = __builtin_ctzll(m);
            m &= m - 1;
            VCR_LDS_FETCH(A, b);
            for (;;) {                                   // ping-pong: the other buffer is in flight while one is shaded
                int nb = m ? __builtin_ctzll(m) : 0;
                bool more = m != 0;
                m &= m - 1;
                VCR_LDS_FETCH(B, nb);
                VCR_SHADE_FWD(A, b);
                if (!more) break;
                b = nb;
                nb = m ? __builtin_ctzll(m) : 0;
                more = m != 0;
                m &= m - 1;
                VCR_LDS_FETCH(A, nb);
                VCR_SHADE_FWD(B, b);
                if (!more) break;
                b = nb;
            }
        }
        // count modes 3 / 4: lane j holds what entry j of this chunk collected -- ONE atomic / store instruction per chunk
        // instead of one per survivor (`id` is this lane's entry of the current chunk)
        if (FC == 3 && chunk_cnt != 0) atomicAdd(count + id, chunk_cnt);
        if (FC == 4 && ((chunk_seen >> lane) & 1ull)) count[id] = 1;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;        // every pixel of the quad has T < 1e-4
        pos = npos; id = nid; q0 = nq0; q1 = nq1; q2 = nq2; q3 = nq3; qs = nqs; valid = nvalid; nid = nnid; nvalid = nnvalid;
    }
    const float C0 = acc_c01.x, C1 = acc_c01.y, C2 = acc_c2n.x, N0 = acc_c2n.y, N1 = acc_n12.x, N2 = acc_n12.y;
    const float D = acc_da.x, A = acc_da.y;
    if (pm.inside) {
        if (FC == 0) {                      // (the image state is the backward's; the count modes have none)
            final_T[pm.pix] = T;
            n_contrib[pm.pix] = last;
        }
        if (FC != 3 && FC != 4) {
            out[0 * (size_t)P + pm.pix] = C0 + T * a.bg[0];
            out[1 * (size_t)P + pm.pix] = C1 + T * a.bg[1];
            out[2 * (size_t)P + pm.pix] = C2 + T * a.bg[2];
        }
        if (FC == 0) {
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
            if (ND == 1) {          // 2DGS distortion sum_{i,j} w_i w_j (m_i-m_j)^2 / 2 = A*M2 - M1^2
                out[(8 + S) * (size_t)P + pm.pix] = A * M2 - M1 * M1;
                moments[pm.pix] = M1; moments[P + pm.pix] = M2;
            }
        }
    }
}

template <int S, bool ISECT, int FC, int ND, bool QL>
__global__ void __launch_bounds__(256) VCR_FWD_ATTR composite_fwd_v2_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                               const float* __restrict__ semv,
                                                               const uint32_t* __restrict__ point_list,
                                                               const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ meta,
                                                               int num_tiles, int gxc, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                               float* __restrict__ moments, float* __restrict__ out,
                                                               int32_t* __restrict__ count, float* __restrict__ score,
                                                               float* __restrict__ ckpt) {
    int sub;
    const int tile = work_item(tile_order, meta, num_tiles, sub);
    if (tile < 0) return;
    const int wv = threadIdx.x >> 6;
    __shared__ float4 s_rec_all[4 * VCR_V2_WREC];
    fwd_v2_item<S, ISECT, FC, ND, QL>(a, rec, semv, point_list, ranges, gxc, final_T, n_contrib, moments, out, count, score, ckpt,
                                      tile, sub, wv, s_rec_all + wv * VCR_V2_WREC);
}

// ================= forward, TWO-PHASE (round 6): every lane walks the list of ITS OWN candidates =============================
// The loop above shades one survivor per iteration on all 64 lanes, and 10-12 of them are hit (profiles/r3_hit_histogram_*.txt,
// profiles/r6_twophase_sim.txt).  Here the survivors of the culling are first collected -- densely, over several 64-entry chunks --
// in a group of up to VCR_TP_CAP staged records.  When the group is full (or the list ends):
//   phase 1 (Gaussian x row parallel): lane (j, r) of a step solves the conic of survivor 8 * step + j on pixel row r of the quad
//     for the interval of columns where alpha can reach 1/255 -- a conservative SUPERSET (slack of quad_touch + a margin): 64
//     (survivor, row) pairs per ~25-instruction step instead of one survivor per ~30-instruction iteration.  The eight row bytes of a
//     survivor ARE its 64-pixel candidate mask; a 64-byte transpose through LDS hands pixel lane (x, y) bit x of row y of the
//     eight survivors of the step, which it appends to its own 64-bit candidate word;
//   phase 2 (pixel parallel, lists per lane): every lane pops ITS next candidate (v_ffbl), fetches that record from the staged
//     group with per-lane LDS addresses and runs the v2 loop's arithmetic on it -- exact hit test included, so the superset costs
//     time, never a result.  An iteration shades 64 different (pixel, Gaussian) pairs; the loop runs max_lane(own candidates)
//     times per 32-slot half of the group: 0.46-0.53 iterations per survivor on the metric scene instead of 1.
// Every pixel still sees its contributors in list order with the same fp32 operations as in the v2 kernel: the image, final_T and
// n_contrib are bit-identical to it (tests/test_raster_parity_gpu.py::test_two_phase_forward_is_bit_identical...).
#ifndef VCR_FWD_TP
#define VCR_FWD_TP 1
#endif
#define VCR_TP_CAP 64
// Plane stride of the staged group: one slot more than the group holds.  Slot VCR_TP_CAP is a DUMMY record whose alpha is 0 (log2 of its
// opacity = -1e30): a lane without a candidate left walks it, so phase 2 carries no "this lane is active" flag from the pop to the hit test
// (two VALU and two scalar instructions per iteration less; same results -- the dummy hits nothing and contributes w = 0).
#define VCR_TP_STRIDE (VCR_TP_CAP + 1)
#ifndef VCR_TP_MAX_TILES_PER_GAUSSIAN
#define VCR_TP_MAX_TILES_PER_GAUSSIAN 5      // frames with more 3-sigma tiles per visible Gaussian keep the v2 loop (vcr_forward_two_phase)
#endif
#ifndef VCR_TP_WAVES
#define VCR_TP_WAVES 0           // 0: compiler's choice
#endif
#if VCR_TP_WAVES > 0
#define VCR_TP_ATTR __attribute__((amdgpu_waves_per_eu(VCR_TP_WAVES)))
#else
#define VCR_TP_ATTR
#endif
#ifndef VCR_TP_PIPE
#define VCR_TP_PIPE 1            // 1: phase 2 fetches a lane's next candidate while it shades the current one
#endif

// What phase 1 needs of a survivor, computed once by its culler lane: with the raw conic (A, B, C), tau = ln(255 opacity) and
// d = centre - pixel, alpha >= 1/255  <=>  A dx^2 + 2 B dx dy + C dy^2 <= 2 tau  <=>  (dx + (B/A) dy)^2 <= 2 tau / A - (det / A^2) dy^2.
// -> (B / A, 2 tau' / A, det' / A^2): tau' carries the slack of quad_touch, det' is a lower bound of A C - B^2 under fp32
// cancellation (needle-shaped footprints), so that rounding can only widen an interval.  Degenerate conics get every column.
__device__ __forceinline__ float4 span_params(const float4 q0, const float4 q1, float X0, float Y0, uint32_t pos1) {
    const float A = q1.x, B = q1.y, C = q1.z;
    float4 p;
    p.w = __uint_as_float(pos1);
    if (!(A > 0.f) || !(C > 0.f)) { p.x = 0.f; p.y = __builtin_inff(); p.z = 0.f; return p; }
    const float tau = __logf(255.f * q0.w);
    const float x0 = X0 - q0.x, x1 = X0 + 7.f - q0.x, y0 = Y0 - q0.y, y1 = Y0 + 7.f - q0.y;
    const float mx = fmaxf(x0 * x0, x1 * x1), my = fmaxf(y0 * y0, y1 * y1);
    const float taus = tau + 0.05f + 2e-6f * (A * mx + C * my);
    const float ia = fast_rcp(A);
    const float ac = A * C, bb = B * B;
    p.x = B * ia;
    p.y = 2.00002f * taus * ia;
    p.z = ((ac - bb) - 4e-7f * (ac + bb)) * ia * ia * 0.99998f;
    return p;
}

// One work item of the two-phase forward: quad `wv` of `tile` (or its 4x4 sub-block `sub`), walked by the calling wave; `srec` / `rbq`
// are the wave's private LDS planes (5 x VCR_TP_STRIDE float4 and 32 uint4).
template <int S, bool ISECT, int ND, bool QL>
__device__ __forceinline__ void fwd_tp_item(const VcrRasterArgs& a, const GeomRec* __restrict__ rec, const uint32_t* __restrict__ point_list,
                                            const uint2* __restrict__ ranges, int gxc, float* __restrict__ final_T,
                                            uint32_t* __restrict__ n_contrib, float* __restrict__ moments, float* __restrict__ out,
                                            int tile, int sub, int wv, float4* const srec, uint4* const rbq) {
    static_assert(S <= 2, "the two-phase forward keeps the semantic features in the record's pad words");
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const PixelMap pm = pixel_of_quad(tile, gx, a.W, a.H, sub, wv);
    const uint2 range = list_range<QL>(ranges, gxc, tile, gx, wv);
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63;
    uint8_t* const rbb = reinterpret_cast<uint8_t*>(rbq);
    if (range.x == range.y) {                              // (wave-uniform) nothing reaches this quad -- 84 % of the metric frame's waves:
        if (pm.inside) {                                   // background, T = 1, no contributor; exactly what the general path would write
            final_T[pm.pix] = 1.f;
            n_contrib[pm.pix] = 0u;
            out[0 * (size_t)P + pm.pix] = 0.f + 1.f * a.bg[0];
            out[1 * (size_t)P + pm.pix] = 0.f + 1.f * a.bg[1];
            out[2 * (size_t)P + pm.pix] = 0.f + 1.f * a.bg[2];
#pragma unroll
            for (int k = 3; k < 8 + S + (ND == 2 ? 2 : (ND == 1 ? 1 : 0)); ++k) out[k * (size_t)P + pm.pix] = 0.f;
            if (ND == 1) { moments[pm.pix] = 0.f; moments[P + pm.pix] = 0.f; }
        }
        return;
    }
    // the dummy record: alpha = 2^-1e30 = 0, everything else finite -- depth 1, so that the mapped depth of the distortion channel
    // (num_dist = 1: 1 / depth) stays finite and w = 0 times it stays 0
    if (lane < 4) srec[lane * VCR_TP_STRIDE + VCR_TP_CAP] = make_float4(0.f, lane == 1 ? -1e30f : 0.f, lane == 1 ? 1.f : 0.f, 0.f);
    const float X0 = (float)((tile % gx) * VCR_TILE + (wv & 1) * 8 + (sub < 0 ? 0 : (sub & 1) * 4));
    const float Y0 = (float)((tile / gx) * VCR_TILE + (wv >> 1) * 8 + (sub < 0 ? 0 : (sub >> 1) * 4));
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
    int fill = 0;                                          // staged survivors of the current group (wave-uniform)
    VCR_TPS(const long long ts_begin = clock64(); long long ts_p1 = 0; long long ts_p2 = 0; unsigned ts_flush = 0; unsigned ts_iter = 0;
            unsigned ts_surv = 0; unsigned ts_chunks = 0; unsigned ts_cand = 0; unsigned ts_hits = 0;)

    // one candidate of this lane: the staged record of slot SLOT (per-lane LDS addresses), the v2 loop's arithmetic.  The
    // centre / conic half of a lane's NEXT candidate is fetched while the current one is shaded; the colour / normal half is
    // fetched at the top of its own iteration (first used ~20 instructions later) into registers both buffers share.
#define VCR_TP_FETCH01(R, SLOT)                                                                               \
    do {                                                                                                      \
        const int s_ = (SLOT);                                                                                \
        R##0 = srec[s_]; R##1 = srec[VCR_TP_STRIDE + s_]; R##p = s_;                                             \
        asm volatile("" ::: "memory");                                                                        \
    } while (0)
#define VCR_TP_FETCH23(SLOT)                                                                                  \
    do {                                                                                                      \
        const int s_ = (SLOT);                                                                                \
        C2 = srec[2 * VCR_TP_STRIDE + s_]; C3 = srec[3 * VCR_TP_STRIDE + s_];                                       \
        asm volatile("" ::: "memory");                                                                        \
    } while (0)
#define VCR_TP_SHADE(R, ACT)                                                                                  \
    do {                                                                                                      \
        const float4 r0 = R##0, r1 = R##1, r2 = C2, r3 = C3;                                                  \
        const f2 gxy = {r0.x, r0.y}, sAC = {r0.z, r0.w};                                                      \
        f2 u; float hs;                                                                                       \
        const float e = gauss_exponent(gxy - fxy, sAC, r1.x, r1.y, u, hs);                                    \
        const float alpha = fminf(VCR_ALPHA_MAX, __builtin_amdgcn_exp2f(e));                                  \
        bool hit = (ACT) && !done && hs <= 0.f && alpha >= VCR_ALPHA_MIN;                                     \
        const float test_T = fmaf(-alpha, T, T);                                                              \
        if (hit && test_T < VCR_T_EPS) { done = true; hit = false; }                                          \
        const float w = hit ? alpha * T : 0.f;                                                                \
        const f2 c01 = {r2.x, r2.y}, c2n = {r2.z, r2.w}, n12 = {r3.x, r3.y};                                  \
        float dep = r1.z;                                                                                     \
        if (ISECT) {                                                                                          \
            const float den = fmaf(c2n.y, rx, fmaf(n12.x, ry, n12.y * rz));                                   \
            if (den > VCR_PLANE_EPS) dep = r1.w * fast_rcp(den) * rz;                                         \
        }                                                                                                     \
        const f2 ww = splat(w);                                                                               \
        acc_c01 = pk_fma(ww, c01, acc_c01);                                                                   \
        acc_c2n = pk_fma(ww, c2n, acc_c2n);                                                                   \
        acc_n12 = pk_fma(ww, n12, acc_n12);                                                                   \
        acc_da = pk_fma(ww, f2{dep, 1.f}, acc_da);                                                            \
        if (ND == 2) M2 = fmaf(w * dep, dep, M2);                                                             \
        if (ND == 1) {                                                                                        \
            const float md = -zc_map * VCR_ZNEAR * fast_rcp(dep);                                             \
            M1 = fmaf(w, md, M1); M2 = fmaf(w * md, md, M2);                                                  \
        }                                                                                                     \
        if (S > 0) {                                                                                          \
            const float sv_[2] = {r3.z, r3.w};                                                                \
_Pragma("unroll")                                                                                             \
            for (int k = 0; k < S; ++k) SM[k] = fmaf(w, sv_[k], SM[k]);                                       \
        }                                                                                                     \
        VCR_TPS(ts_hits += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(hit)); ++ts_iter;)                  \
        T = hit ? test_T : T;                                                                                 \
        lslot = hit ? R##p : lslot;       /* (its list position is looked up once per group)                */ \
    } while (0)
    // walk the candidates of one 32-slot half of the group, front to back; a lane whose word is empty walks the dummy slot
#define VCR_TP_POP(W_, B_, ACT_)                                                                              \
    do { ACT_ = true; B_ = W_ != 0u ? __builtin_ctz(W_) : dummy_rel_; W_ &= W_ - 1u; } while (0)
#if VCR_TP_PIPE
#define VCR_TP_WALK(WORD, BASE)                                                                               \
    do {                                                                                                      \
        uint32_t w_ = (WORD);                                                                                 \
        const int dummy_rel_ = VCR_TP_CAP - (BASE);       /* the dummy slot, relative to this half             */ \
        if (__builtin_amdgcn_ballot_w64(w_ != 0u) != 0) {                                                     \
            float4 A0, A1, B0, B1, C2, C3; int Ap, Bp;                                                        \
            int b_, nb_; bool act_, nact_;                                                                    \
            VCR_TP_POP(w_, b_, act_);                                                                         \
            VCR_TP_FETCH01(A, (BASE) + b_);                                                                   \
            for (;;) {                                                                                        \
                bool more_ = __builtin_amdgcn_ballot_w64(w_ != 0u) != 0;                                      \
                VCR_TP_FETCH23(Ap);                                                                           \
                VCR_TP_POP(w_, nb_, nact_);                                                                   \
                VCR_TP_FETCH01(B, (BASE) + nb_);                                                              \
                VCR_TP_SHADE(A, act_);                                                                        \
                if (!more_) break;                                                                            \
                more_ = __builtin_amdgcn_ballot_w64(w_ != 0u) != 0;                                           \
                VCR_TP_FETCH23(Bp);                                                                           \
                VCR_TP_POP(w_, b_, act_);                                                                     \
                VCR_TP_FETCH01(A, (BASE) + b_);                                                               \
                VCR_TP_SHADE(B, nact_);                                                                       \
                if (!more_) break;                                                                            \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#else
#define VCR_TP_WALK(WORD, BASE)                                                                               \
    do {                                                                                                      \
        uint32_t w_ = (WORD);                                                                                 \
        const int dummy_rel_ = VCR_TP_CAP - (BASE);                                                           \
        while (__builtin_amdgcn_ballot_w64(w_ != 0u) != 0) {                                                  \
            float4 A0, A1, C2, C3; int Ap;                                                                    \
            int b_; bool act_;                                                                                \
            VCR_TP_POP(w_, b_, act_);                                                                         \
            VCR_TP_FETCH01(A, (BASE) + b_);                                                                   \
            VCR_TP_FETCH23(Ap);                                                                               \
            VCR_TP_SHADE(A, act_);                                                                            \
        }                                                                                                     \
    } while (0)
#endif
    // Phase 1 for the `fill` staged survivors: all steps write their row bytes (row r of the quad owns 64 consecutive bytes, byte
    // 8 * step + j = survivor 8 * step + j), ONE LDS round trip, then pixel lane (x, y) pulls bit x out of the 64 bytes of row y:
    // per 4 bytes a shift, a mask and a v_dot4_u32_u8 with the weights (1, 2, 4, 8) / (16, 32, 64, 128).
#define VCR_TP_ROWBYTE(ST)                                                                                    \
    do {                                                                                                      \
        const int slot_ = 8 * (ST) + (lane >> 3);                                                             \
        const float4 g_ = srec[slot_], p_ = srec[4 * VCR_TP_STRIDE + slot_];                                     \
        const float dy_ = (g_.y - Y0) - (float)(lane & 7);                                                    \
        const float D_ = fmaf(-p_.z, dy_ * dy_, p_.y);                                                        \
        const float h_ = __builtin_amdgcn_sqrtf(fmaxf(D_, 0.f)) * 1.00001f + 0.01f;                           \
        const float xc_ = fmaf(p_.x, dy_, g_.x - X0);                                                         \
        const float lo_ = fmaxf(ceilf(xc_ - h_), 0.f), hi_ = fminf(floorf(xc_ + h_), 7.f);                    \
        const bool ok_ = D_ >= 0.f && lo_ <= hi_;                                                             \
        const uint32_t row_ = ok_ ? (2u << (int)hi_) - (1u << (int)lo_) : 0u;                                 \
        rbb[(lane & 7) * 64 + 8 * (ST) + (lane >> 3)] = (uint8_t)row_;                                        \
    } while (0)
#define VCR_TP_BITS(V0, V1, SH)                                                                               \
    (__builtin_amdgcn_udot4(((V1) >> (lane & 7)) & 0x01010101u, 0x80402010u,                                  \
                            __builtin_amdgcn_udot4(((V0) >> (lane & 7)) & 0x01010101u, 0x08040201u, 0u, false), false) << (SH))
#define VCR_TP_FLUSH()                                                                                        \
    do {                                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                                      \
        VCR_TPS(const long long tf0_ = clock64(); ++ts_flush;)                                                \
        const int steps_ = (fill + 7) >> 3;                                                                   \
        for (int st_ = 0; st_ < steps_; ++st_) VCR_TP_ROWBYTE(st_);                                           \
        __builtin_amdgcn_wave_barrier();                                                                      \
        uint32_t lo_w = 0u, hi_w = 0u;                                                                        \
        {                                                                                                     \
            const uint4 v0_ = rbq[(lane >> 3) * 4], v1_ = rbq[(lane >> 3) * 4 + 1];                           \
            lo_w = VCR_TP_BITS(v0_.x, v0_.y, 0) | VCR_TP_BITS(v0_.z, v0_.w, 8) | VCR_TP_BITS(v1_.x, v1_.y, 16) | VCR_TP_BITS(v1_.z, v1_.w, 24); \
        }                                                                                                     \
        if (steps_ > 4) {                                                                                     \
            const uint4 v2_ = rbq[(lane >> 3) * 4 + 2], v3_ = rbq[(lane >> 3) * 4 + 3];                       \
            hi_w = VCR_TP_BITS(v2_.x, v2_.y, 0) | VCR_TP_BITS(v2_.z, v2_.w, 8) | VCR_TP_BITS(v3_.x, v3_.y, 16) | VCR_TP_BITS(v3_.z, v3_.w, 24); \
        }                                                                                                     \
        __builtin_amdgcn_wave_barrier();                                                                      \
        /* slots >= fill hold the previous group's survivors: cut the words there; pixels that are finished have no candidates */ \
        const unsigned long long fm_ = fill >= 64 ? ~0ull : (1ull << fill) - 1ull;                            \
        lo_w = done ? 0u : lo_w & (uint32_t)fm_;                                                              \
        hi_w = done ? 0u : hi_w & (uint32_t)(fm_ >> 32);                                                      \
        int lslot = -1;                                                                                       \
        VCR_TPS(const long long tf1_ = clock64(); ts_p1 += tf1_ - tf0_;                                       \
                { unsigned c_ = (unsigned)(__popc(lo_w) + __popc(hi_w)); for (int o_ = 32; o_ > 0; o_ >>= 1) c_ += (unsigned)__shfl_xor((int)c_, o_); ts_cand += c_; }) \
        VCR_TP_WALK(lo_w, 0);                                                                                 \
        if (steps_ > 4) VCR_TP_WALK(hi_w, 32);                                                                \
        if (lslot >= 0) last = __float_as_uint(srec[4 * VCR_TP_STRIDE + lslot].w);                               \
        VCR_TPS(ts_p2 += clock64() - tf1_;)                                                                   \
        fill = 0;                                                                                             \
    } while (0)

    uint32_t pos = range.x;
    // Chunk pipeline.  Only the first half of a record -- centre, opacity, conic: what the culling reads -- is gathered ahead of its
    // chunk (VCR_TP_DEPTH chunks ahead, ids one further); the colour / normal half is fetched for the SURVIVORS only, when their
    // first half is staged (same 64-byte record: it comes from the L1 / L2), and written to LDS one iteration LATER, behind the
    // next chunk's culling -- waiting for it on the spot would also drain the head gathers behind it in the in-order vmcnt queue.
#define VCR_GATHER_HEAD(ID, Q0, Q1)                                                 \
    do {                                                                            \
        const float4* _src = reinterpret_cast<const float4*>(rec + (ID));           \
        Q0 = _src[0]; Q1 = _src[1];                                                 \
    } while (0)
#ifndef VCR_TP_DEPTH
#define VCR_TP_DEPTH 1           // measured: 2 costs the fifth wave per SIMD (106 VGPRs) and loses 4 % (profiles/r6_fwd_variants.txt)
#endif
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    uint32_t id, id1, id2 = 0u; float4 q0, q1, h1q0 = zero4, h1q1 = zero4; bool valid, valid1, valid2 = false;
    VCR_LOAD_ID(pos + lane, range.y, id, valid);
    VCR_GATHER_HEAD(id, q0, q1);
    VCR_LOAD_ID(pos + 64 + lane, range.y, id1, valid1);
    if (VCR_TP_DEPTH > 1) {
        VCR_GATHER_HEAD(id1, h1q0, h1q1);
        VCR_LOAD_ID(pos + 128 + lane, range.y, id2, valid2);
    }
    float4 tq2 = zero4, tq3 = zero4; int tslot = -1;       // colour / normal half in flight for the survivor staged in slot tslot
#define VCR_TP_LAND_TAILS()                                                                                   \
    do {                                                                                                      \
        if (tslot >= 0) {                                                                                     \
            srec[2 * VCR_TP_STRIDE + tslot] = make_float4(tq2.x, tq2.y, tq2.z, tq3.x);                           \
            srec[3 * VCR_TP_STRIDE + tslot] = make_float4(tq3.y, tq3.z, tq2.w, tq3.w);                           \
        }                                                                                                     \
        tslot = -1;                                                                                           \
    } while (0)
    while (pos < range.y) {
        const uint32_t npos = pos + 64;
        uint32_t idn; bool validn; float4 hnq0, hnq1;
        if (VCR_TP_DEPTH > 1) {
            VCR_GATHER_HEAD(id2, hnq0, hnq1);                             // heads of chunk c + 2
            VCR_LOAD_ID(npos + 128 + lane, range.y, idn, validn);        // ids of chunk c + 3
        } else {
            VCR_GATHER_HEAD(id1, hnq0, hnq1);                             // heads of chunk c + 1
            VCR_LOAD_ID(npos + 64 + lane, range.y, idn, validn);         // ids of chunk c + 2
        }
        float bx0, by0, bw, bh;
        live_box(__builtin_amdgcn_ballot_w64(!done), X0, Y0, bx0, by0, bw, bh);
        const bool keep = valid && quad_touch(q0, q1, bx0, by0, bw, bh);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const int cnt = __popcll(m);
        VCR_TP_LAND_TAILS();                                         // the previous chunk's survivors are complete now
        if (fill + cnt > VCR_TP_CAP) {                               // (wave-uniform)
            VCR_TP_FLUSH();
            if (__builtin_amdgcn_ballot_w64(!done) == 0) { pos = range.y; break; }        // every pixel of the quad has T < 1e-4
        }
        if (keep) {
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            tq2 = src[2]; tq3 = src[3];
            const int slot = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            tslot = slot;
            srec[slot] = make_float4(q0.x, q0.y, -VCR_L2E * q1.x, -VCR_L2E * q1.z);
            srec[VCR_TP_STRIDE + slot] = make_float4(-VCR_L2E * q1.y, __builtin_amdgcn_logf(q0.w), q0.z, q1.w);
            srec[4 * VCR_TP_STRIDE + slot] = span_params(q0, q1, X0, Y0, pos - range.x + (uint32_t)lane + 1u);
        }
        fill += cnt;
        VCR_TPS(ts_surv += (unsigned)cnt; ++ts_chunks;)
        pos = npos;
        if (VCR_TP_DEPTH > 1) {
            id = id1; q0 = h1q0; q1 = h1q1; valid = valid1;
            id1 = id2; h1q0 = hnq0; h1q1 = hnq1; valid1 = valid2;
            id2 = idn; valid2 = validn;
        } else {
            id = id1; q0 = hnq0; q1 = hnq1; valid = valid1;
            id1 = idn; valid1 = validn;
        }
    }
    VCR_TP_LAND_TAILS();
    if (fill > 0) VCR_TP_FLUSH();
#undef VCR_TP_LAND_TAILS
#ifdef VCR_TPSTATS
    if (lane == 0 && range.y > range.x) {
        const long long tot = clock64() - ts_begin;
        atomicAdd(&g_tpstats[0], 1ull); atomicAdd(&g_tpstats[1], (unsigned long long)tot);
        atomicAdd(&g_tpstats[2], (unsigned long long)(tot - ts_p1 - ts_p2)); atomicAdd(&g_tpstats[3], (unsigned long long)ts_p1);
        atomicAdd(&g_tpstats[4], (unsigned long long)ts_p2); atomicAdd(&g_tpstats[5], (unsigned long long)ts_flush);
        atomicAdd(&g_tpstats[6], (unsigned long long)ts_iter); atomicAdd(&g_tpstats[7], (unsigned long long)ts_surv);
        atomicAdd(&g_tpstats[8], (unsigned long long)ts_chunks); atomicAdd(&g_tpstats[9], (unsigned long long)ts_cand);
        atomicMax(&g_tpstats[10], (unsigned long long)tot); atomicAdd(&g_tpstats[11], (unsigned long long)ts_hits);
    }
#endif
#undef VCR_GATHER_HEAD
#undef VCR_TP_FLUSH
#undef VCR_TP_BITS
#undef VCR_TP_ROWBYTE
#undef VCR_TP_WALK
#undef VCR_TP_POP
#undef VCR_TP_SHADE
#undef VCR_TP_FETCH23
#undef VCR_TP_FETCH01
    if (pm.inside) {
        final_T[pm.pix] = T;
        n_contrib[pm.pix] = last;
        out[0 * (size_t)P + pm.pix] = acc_c01.x + T * a.bg[0];
        out[1 * (size_t)P + pm.pix] = acc_c01.y + T * a.bg[1];
        out[2 * (size_t)P + pm.pix] = acc_c2n.x + T * a.bg[2];
        out[3 * (size_t)P + pm.pix] = acc_da.x;
        out[4 * (size_t)P + pm.pix] = acc_c2n.y;
        out[5 * (size_t)P + pm.pix] = acc_n12.x;
        out[6 * (size_t)P + pm.pix] = acc_n12.y;
        out[7 * (size_t)P + pm.pix] = acc_da.y;
#pragma unroll
        for (int k = 0; k < S; ++k) out[(8 + k) * (size_t)P + pm.pix] = SM[k];
        if (ND == 2) {
            out[(8 + S) * (size_t)P + pm.pix] = acc_da.x;
            out[(9 + S) * (size_t)P + pm.pix] = M2;
        }
        if (ND == 1) {
            out[(8 + S) * (size_t)P + pm.pix] = acc_da.y * M2 - M1 * M1;
            moments[pm.pix] = M1; moments[P + pm.pix] = M2;
        }
    }
}

// Block form (rounds 2-6a): one workgroup per tile in `tile_order`, its four waves take the four quads.
template <int S, bool ISECT, int ND, bool QL>
__global__ void __launch_bounds__(256) VCR_TP_ATTR composite_fwd_tp_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                               const uint32_t* __restrict__ point_list,
                                                               const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ meta,
                                                               int num_tiles, int gxc, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                               float* __restrict__ moments, float* __restrict__ out) {
    int sub;
    const int tile = work_item(tile_order, meta, num_tiles, sub);
    if (tile < 0) return;
    const int wv = threadIdx.x >> 6;
    __shared__ float4 s_rec_all[4 * 5 * VCR_TP_STRIDE];    // per wave: 5 planes x (64 slots + the dummy) x 16 B
    __shared__ uint4 s_rb_all[4 * 32];                     // per wave: 8 pixel rows x 64 survivors, one byte each
    fwd_tp_item<S, ISECT, ND, QL>(a, rec, point_list, ranges, gxc, final_T, n_contrib, moments, out, tile, sub, wv,
                                  s_rec_all + wv * 5 * VCR_TP_STRIDE, s_rb_all + wv * 32);
}

// ---- shading macros of the compositing backward (one survivor, all 64 lanes = the whole 8x8 quad) ---------------------------
// Used by the row-packed kernel below for the chunks where packing does not pay (the round-2 kernel that was built from them
// alone, `composite_bwd_v2`, left the library in round 5: profiles/r3_bwd_rows_ab.txt holds its A/B).
// Gradient slots of one (pixel, Gaussian) pair: expects r1..r3 (staged record), d, u, araw and `hit` in scope; defines
// gid, v[8] (16 raw slots, GradRec order) and vs[S]; updates T and Bsuf.  Lanes without a hit produce zeros.
#define VCR_BWD_MATH(R)                                                                                                  \
    const f2 c01 = {r2.x, r2.y}, c2n = {r2.z, r2.w}, n12 = {r3.x, r3.y};                                         \
    const float zc = r1.z, pl = r1.w;                                                                            \
    const uint32_t gid = __float_as_uint(r3.z);                                                                  \
    f2 v[8];                                                                                                     \
    float vs[S > 0 ? S : 1];                                                                                     \
    {                                                                                                            \
        const float ah = hit ? araw : 0.f;                                                                       \
        const float alpha = fminf(VCR_ALPHA_MAX, ah);                                                            \
        const float oma_ = 1.f - alpha;                                                                          \
        float inv1ma = fast_rcp(oma_);                                                                           \
        if (VCR_RCP_NEWTON) inv1ma = fmaf(fmaf(-oma_, inv1ma, 1.f), inv1ma, inv1ma);                              \
        T *= inv1ma;                                                                                             \
        const double inv_d_ = 1.0 / (1.0 - (double)alpha);                                                       \
        if (VCR_DBG_PIX64) { Td *= inv_d_; T = (float)Td; }                                                      \
        const float w = VCR_DBG_PIX64 ? (float)((double)alpha * Td) : alpha * T;                                 \
        float dep = zc, iden = 0.f;                                                                              \
        bool isect = false;                                                                                      \
        if (ISECT) {                                                                                             \
            const float den = fmaf(c2n.y, rx, fmaf(n12.x, ry, n12.y * rz));   /* (explicit: both copies of the macro must round alike) */                                              \
            isect = den > VCR_PLANE_EPS;                                                                         \
            iden = isect ? fast_rcp(den) : 0.f;                                                                  \
            dep = isect ? pl * iden * rz : zc;                                                                   \
        }                                                                                                        \
        f2 fa = pk_fma(c01, g01, f2{g[7], 0.f});                                                                 \
        fa = pk_fma(c2n, g24, fa);                                                                               \
        fa = pk_fma(n12, g56, fa);                                                                               \
        float fg = fmaf(dep, g[3], fa.x + fa.y);                                                                 \
        if (ND == 2) fg += dep * dep * gm2;                                                                      \
        float md = 0.f, idep = 0.f;                                                                              \
        if (ND == 1) {                                                                                           \
            idep = fast_rcp(dep);                                                                                \
            md = -zc_map * VCR_ZNEAR * idep;                                                                     \
            fg += md * gm1 + md * md * gm2;                                                                      \
        }                                                                                                        \
        { const float4 r4_ = R##4; const float sv_[4] = {r4_.x, r4_.y, r4_.z, r4_.w};                            \
_Pragma("unroll")                                                                                                        \
        for (int k = 0; k < S; ++k) fg += sv_[k] * g[8 + k]; }                                                   \
        float dL_dalpha = fmaf(T, fg, -Bsuf * inv1ma);                                                           \
        Bsuf = fmaf(w, fg, Bsuf);                                                                                \
        if (VCR_DBG_PIX64) {                                                                                     \
            dL_dalpha = (float)(Td * (double)fg - Bd * inv_d_);                                                  \
            Bd += (double)alpha * Td * (double)fg;                                                               \
        }                                                                                                        \
        const float pw = ah * dL_dalpha;                                                                         \
        const f2 pp = splat(pw);                                                                                 \
        v[0] = u * pp;                                                                                           \
        v[1] = f2{fabsf(v[0].x), fabsf(v[0].y)};                                                                 \
        const f2 dp = d * pp;                                                                                    \
        v[2] = d * dp;                                                                                           \
        v[3] = f2{d.x * dp.y, pw};                                                                               \
        const f2 ww = splat(w);                                                                                  \
        v[4] = ww * g01;                                                                                         \
        const float wd = w * (ND == 2 ? g[3] + 2.f * dep * gm2                                                   \
                                      : (ND == 1 ? g[3] + (gm1 + 2.f * md * gm2) * zc_map * VCR_ZNEAR * idep * idep : g[3])); \
        v[5] = f2{w * g24.x, isect ? 0.f : wd};                                                                  \
        const float k1 = wd * rz * iden;                                                                         \
        const float k2 = -k1 * pl * iden;                                                                        \
        v[6] = f2{k1, fmaf(k2, rx, w * g24.y)};                                                                  \
        v[7] = pk_fma(splat(k2), ryz, ww * g56);                                                                 \
_Pragma("unroll")                                                                                                        \
        for (int k = 0; k < S; ++k) vs[k] = w * g[8 + k];                                                        \
    }
// Shading + gradient of one survivor whose staged record sits in R0..R3 (see the forward kernel for the staging).
// Branch-free inside: lanes without a hit run with alpha = 0, which makes T, Bsuf and every slot a no-op / zero.
// Slots are RAW sums; preprocess_bwd applies the per-Gaussian constants (GradRec in vcr_common.h).
#define VCR_SHADE_BWD(R, B)                                                                                              \
do {                                                                                                             \
    const float4 r0 = R##0, r1 = R##1, r2 = R##2, r3 = R##3; const int sb_ = (B);                                \
    const uint32_t idx1 = VCR_BWD_IDX1(sb_);                                                                     \
    const f2 gxy = {r0.x, r0.y}, sAC = {r0.z, r0.w};                                                             \
    const f2 d = gxy - fxy;                                                                                      \
    f2 u; float hs;                                                                                              \
    const float e = gauss_exponent(d, sAC, r1.x, r1.y, u, hs);                                                   \
    const float araw = __builtin_amdgcn_exp2f(e);                                                                \
    const bool hit = idx1 <= lastc && hs <= 0.f && araw >= VCR_ALPHA_MIN;                                        \
    const unsigned long long hitm_ = __builtin_amdgcn_ballot_w64(hit);                                           \
    VCR_COUNT_HITS(65, hitm_);                                                                                   \
    if (hitm_ != 0) {                                                                                            \
    VCR_BWD_MATH(R);                                                                                             \
    if (VCR_BWD_SPARSE_HITS > 0 && __popcll(hitm_) <= VCR_BWD_SPARSE_HITS) {                                    \
        /* one or two pixels hit: no wave reduction -- the hitting lanes add their 16 values themselves            */ \
        if (hit) {                                                                                               \
            float* dst_ = reinterpret_cast<float*>(sgrad + gid);                                                 \
_Pragma("unroll")                                                                                                        \
            for (int j = 0; j < 8; ++j) { atomicAdd(dst_ + 2 * j, v[j].x); atomicAdd(dst_ + 2 * j + 1, v[j].y); }  \
        }                                                                                                        \
    } else {                                                                                                     \
    float r4[4];                                                                                                 \
    if (VCR_KO & 4) { r4[0] = v[0].x + v[4].x; r4[1] = v[1].y + v[5].y; r4[2] = v[2].x + v[6].y; r4[3] = v[3].y + v[7].x; } \
    else wave_reduce16(v, r4);                                                                                   \
    if ((lane & 15) < 4) {                                                                                       \
        const int sub = lane & 15;                                                                               \
        const float val = sub == 0 ? r4[0] : (sub == 1 ? r4[1] : (sub == 2 ? r4[2] : r4[3]));                    \
        const int k = 8 * (lane >> 5) + 4 * ((lane >> 4) & 1) + sub;                                             \
        if ((VCR_KO & 2) ? val == 1.2345e-30f : val != 0.f) VCR_GRAD_ATOMIC(gid, k, val);                        \
    }                                                                                                            \
    }                                                                                                            \
    if (S > 0) {     /* semantic gradients: DPP row sums only (no cross-row ds_bpermute round trips); the four row   */ \
        float ts = 0.f;  /* totals of feature k sit in lanes 16 r + k and go out as ONE atomic instruction            */ \
_Pragma("unroll")                                                                                                        \
        for (int k = 0; k < S; ++k) { const float t = row_sum16(vs[k]); ts = (lane & 15) == k ? t : ts; }            \
        if ((lane & 15) < S && ts != 0.f) atomicAdd(sgrad_sem + (size_t)gid * S + (lane & 15), ts);              \
    }                                                                                                            \
    }                                                                                                            \
} while (0)

// ================= backward, row-packed: four 4x4 sub-blocks of the quad walk their OWN survivor lists ======================
// The survivors of the 8x8 culling hit 10 of the 64 pixels on average (profiles/r3_hit_histogram_metric.txt) -- the wave
// shades 64 lanes for them all the same.  Here lane l = 16 r + i owns pixel (4 (r & 1) + (i & 3), 4 (r >> 1) + (i >> 2)) of
// the quad: DPP row r is the 4x4 sub-block r.  The cullers test each list entry against the four sub-blocks' live boxes
// (four ballots), and the shading loop then advances the four rows INDEPENDENTLY -- each row pops its own next survivor
// (per-lane LDS address, a broadcast read within the row), so an iteration carries up to four different Gaussians and the
// loop runs max_r |list_r| times instead of |union_r list_r|.  The 16 slots reduce inside the row only (mirror butterfly on
// DPP, no cross-row traffic) and lane i of every row adds slot i of ITS Gaussian: one atomic instruction per iteration,
// as before.  T / Bsuf are per-pixel state, so rows need not agree on where they are in the list.
__device__ __forceinline__ PixelMap pixel_of_thread_rows(int tile, int gx, int W, int H, int sub, int wv) {
    const int lane = threadIdx.x & 63;
    const int r = lane >> 4, i = lane & 15;
    PixelMap p;
    p.x = (tile % gx) * VCR_TILE + (wv & 1) * 8 + (r & 1) * 4 + (i & 3);
    p.y = (tile / gx) * VCR_TILE + (wv >> 1) * 8 + (r >> 1) * 4 + (i >> 2);
    p.inside = p.x < W && p.y < H && (sub < 0 || r == sub);      // split work items: the wave owns row `sub` only
    p.pix = p.y * W + p.x;
    return p;
}

// bounding box of the live lanes of one row (bit i = pixel (i & 3, i >> 2) of the 4x4 block at (X0, Y0))
__device__ __forceinline__ void live_box16(unsigned live, float X0, float Y0, float& bx0, float& by0, float& bw, float& bh) {
    const int y0 = __builtin_ctz(live) >> 2, y1 = (31 - __builtin_clz(live)) >> 2;
    unsigned c = live | (live >> 8);
    c |= c >> 4;
    const unsigned cols = c & 0xFu;
    const int x0 = __builtin_ctz(cols), x1 = 31 - __builtin_clz(cols);
    bx0 = X0 + (float)x0; by0 = Y0 + (float)y0; bw = (float)(x1 - x0); bh = (float)(y1 - y0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// Sum 16 per-lane values (slot 2j = v[j].x, 2j+1 = v[j].y) over the 16 lanes of each DPP row; lane i of the row returns the
// row total of slot i.  Halving butterfly on involutive lane permutations (row_mirror, row_half_mirror, reversed quad,
// swapped pairs): at each level a lane keeps the half of its live slots that its position selects and receives the
// partner's copy of the same slots.
__device__ __forceinline__ float row_reduce16(const f2 v[8], int lane) {
    // levels 1 and 2 (8 + 4 kept values): the two halves of the row / of each 8-lane half write complementary DPP banks of
    // the same destination, so a kept value costs two v_add_f32_dpp and no selects.  (Hand-scheduled: the s_nop cover the
    // VALU-write -> DPP-read hazard at both ends of the block, which the compiler cannot see into.)
    float a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3;
#define VCR_L1(D, LO, HI) "v_add_f32_dpp " D ", " LO ", " LO " row_mirror row_mask:0xf bank_mask:0x3\n\t" \
                          "v_add_f32_dpp " D ", " HI ", " HI " row_mirror row_mask:0xf bank_mask:0xc\n\t"
#define VCR_L2(D, LO, HI) "v_add_f32_dpp " D ", " LO ", " LO " row_half_mirror row_mask:0xf bank_mask:0x5\n\t" \
                          "v_add_f32_dpp " D ", " HI ", " HI " row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
    asm("s_nop 1\n\t"
        VCR_L1("%0", "%12", "%20") VCR_L1("%1", "%13", "%21") VCR_L1("%2", "%14", "%22") VCR_L1("%3", "%15", "%23")
        VCR_L1("%4", "%16", "%24") VCR_L1("%5", "%17", "%25") VCR_L1("%6", "%18", "%26") VCR_L1("%7", "%19", "%27")
        VCR_L2("%8", "%0", "%4") VCR_L2("%9", "%1", "%5") VCR_L2("%10", "%2", "%6") VCR_L2("%11", "%3", "%7")
        "s_nop 1"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7),
          "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(v[0].x), "v"(v[0].y), "v"(v[1].x), "v"(v[1].y), "v"(v[2].x), "v"(v[2].y), "v"(v[3].x), "v"(v[3].y),
          "v"(v[4].x), "v"(v[4].y), "v"(v[5].x), "v"(v[5].y), "v"(v[6].x), "v"(v[6].y), "v"(v[7].x), "v"(v[7].y));
#undef VCR_L1
#undef VCR_L2
    const bool h3 = lane & 2, h4 = lane & 1;
    const float c0 = (h3 ? b2 : b0) + dpp_get<0x1B>(h3 ? b0 : b2);      // quad_perm [3,2,1,0]
    const float c1 = (h3 ? b3 : b1) + dpp_get<0x1B>(h3 ? b1 : b3);
    return (h4 ? c1 : c0) + dpp_get<0xB1>(h4 ? c0 : c1);                // quad_perm [1,0,3,2]
}

// 1-based list position of the staged survivor in slot SLOT (third plane of its record in `r3`): per-chunk staging -- slot = position in the chunk
#define VCR_BWD_IDX1(SLOT) ((uint32_t)chunk * 64u + (uint32_t)(SLOT) + 1u)
// One work item of the row-packed backward: quad `wv` of `tile` (split items: its 4x4 sub-block `sub`), walked back to front by the
// calling wave; `srec` = the wave's private LDS planes (WREC float4).
template <int S, bool ISECT, int ND, bool QL>
__device__ __forceinline__ void bwd_rows_item(const VcrRasterArgs& a, const GeomRec* __restrict__ rec, const float* __restrict__ semv,
                                              const uint32_t* __restrict__ point_list, const uint2* __restrict__ ranges, int gxc,
                                              const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                              const float* __restrict__ moments, const float* __restrict__ ckpt,
                                              const float* __restrict__ dL_dout, GradRec* __restrict__ sgrad,
                                              float* __restrict__ sgrad_sem, int rows_bias, int rows_pair_cost,
                                              int tile, int sub, int wv, float4* const srec) {
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const PixelMap pm = pixel_of_thread_rows(tile, gx, a.W, a.H, sub, wv);
    const uint2 range = list_range<QL>(ranges, gxc, tile, gx, wv);
    if (range.x == range.y) return;                        // wave-uniform: nothing reaches this quad
    const float* const ck = ckpt + ckpt_base<QL>(range.x, gxc, tile, gx, wv) + (size_t)((pm.y & 7) * 8 + (pm.x & 7));
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63, row = lane >> 4;
    constexpr bool SEM_IN_REC = false;
    constexpr int FC = 0;
    const float X0 = (float)((tile % gx) * VCR_TILE + (wv & 1) * 8);
    const float Y0 = (float)((tile / gx) * VCR_TILE + (wv >> 1) * 8);
    const f2 fxy = {(float)pm.x, (float)pm.y};
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float g[8 + (S > 0 ? S : 0)];
#pragma unroll
    for (int c = 0; c < 8 + S; ++c) g[c] = pm.inside ? dL_dout[c * (size_t)P + pm.pix] : 0.f;
    float gm2 = 0.f;
    if (ND == 2 && pm.inside) { g[3] += dL_dout[(8 + S) * (size_t)P + pm.pix]; gm2 = dL_dout[(9 + S) * (size_t)P + pm.pix]; }
    const float Tf = pm.inside ? final_T[pm.pix] : 1.f;
    float gm1 = 0.f;
    const float zc_map = VCR_ZFAR / (VCR_ZFAR - VCR_ZNEAR);
    if (ND == 1 && pm.inside) {
        const float gd = dL_dout[(8 + S) * (size_t)P + pm.pix];
        const float m1 = moments[pm.pix], m2 = moments[P + pm.pix];
        gm1 = -2.f * m1 * gd; gm2 = (1.f - Tf) * gd; g[7] += m2 * gd;
    }
    const uint32_t lastc = pm.inside ? n_contrib[pm.pix] : 0u;
    const float bgdot = Tf * (a.bg[0] * g[0] + a.bg[1] * g[1] + a.bg[2] * g[2]);
    uint32_t maxc = lastc;
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, o));
    if (maxc == 0) return;                                 // wave-uniform
    float T = Tf, Bsuf = bgdot;
    double Td = (double)Tf, Bd = (double)bgdot;      // (VCR_DBG_PIX64 only; dead otherwise)
    (void)Td; (void)Bd;
    const f2 g01 = {g[0], g[1]}, g24 = {g[2], g[4]}, g56 = {g[5], g[6]}, ryz = {ry, rz};

    int chunk = (int)((maxc - 1) / 64);
    uint32_t id, nid; float4 q0, q1, q2, q3, qs = {0.f, 0.f, 0.f, 0.f}; bool valid, nvalid;
    const uint32_t lim = range.x + maxc;
    VCR_LOAD_ID(range.x + (uint32_t)chunk * 64u + lane, lim, id, valid);
    VCR_GATHER_REC(id, q0, q1, q2, q3);
    VCR_GATHER_SEM(id, qs);
    VCR_LOAD_ID(chunk > 0 ? range.x + (uint32_t)(chunk - 1) * 64u + lane : lim, lim, nid, nvalid);
    for (; chunk >= 0; --chunk) {
        uint32_t nnid; float4 nq0, nq1, nq2, nq3, nqs = {0.f, 0.f, 0.f, 0.f}; bool nnvalid;
        VCR_GATHER_REC(nid, nq0, nq1, nq2, nq3);
        VCR_GATHER_SEM(nid, nqs);
        VCR_LOAD_ID(chunk > 1 ? range.x + (uint32_t)(chunk - 2) * 64u + lane : lim, lim, nnid, nnvalid);
        const float ckT = (VCR_T_ANCHOR && chunk > 0 && pm.inside) ? ck[(size_t)chunk * VCR_CKPT_STRIDE] : 1.f;      // (see the v2 kernel)
        // cull against the live box of each 4x4 sub-block; one survivor mask per row
        const unsigned long long live = __builtin_amdgcn_ballot_w64(lastc > (uint32_t)chunk * 64u);
        unsigned long long mr[4];
        bool keep = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned lr = (unsigned)(live >> (16 * r)) & 0xFFFFu;
            bool k = false;
            if (lr != 0) {                                   // wave-uniform
                float bx0, by0, bw, bh;
                live_box16(lr, X0 + (float)((r & 1) * 4), Y0 + (float)((r >> 1) * 4), bx0, by0, bw, bh);
                k = valid && quad_touch(q0, q1, bx0, by0, bw, bh);
            }
            mr[r] = __builtin_amdgcn_ballot_w64(k);
            keep |= k;
        }
        if (keep) {
            srec[0 * 64 + lane] = make_float4(q0.x, q0.y, -VCR_L2E * q1.x, -VCR_L2E * q1.z);
            srec[1 * 64 + lane] = make_float4(-VCR_L2E * q1.y, __builtin_amdgcn_logf(q0.w), q0.z, q1.w);
            srec[2 * 64 + lane] = make_float4(q2.x, q2.y, q2.z, q3.x);
            srec[3 * 64 + lane] = make_float4(q3.y, q3.z, __uint_as_float(id), 0.f);
            if (S > 0) srec[4 * 64 + lane] = qs;
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long many = mr[0] | mr[1] | mr[2] | mr[3];
        const int iters = max(max(__popcll(mr[0]), __popcll(mr[1])), max(__popcll(mr[2]), __popcll(mr[3])));
        // Per chunk (wave-uniform): the row-packed loop runs `iters` = max_r |list_r| iterations, the whole-quad loop (the v2
        // kernel's: one survivor per iteration for all 64 lanes, 64-lane butterfly -- it does not care how lanes map to
        // pixels) |union_r list_r|.  Large Gaussians sit in all four lists and gain nothing from the packing.
        const int pairs = __popcll(mr[0]) + __popcll(mr[1]) + __popcll(mr[2]) + __popcll(mr[3]);
        if (many && iters * 16 + pairs * rows_pair_cost > __popcll(many) * rows_bias && !(VCR_KO & 1)) {
            unsigned long long m = many;
            float4 A0, A1, A2, A3, A4 = {0.f, 0.f, 0.f, 0.f}, B0, B1, B2, B3, B4 = {0.f, 0.f, 0.f, 0.f};
            int b = 63 - __builtin_clzll(m);
            m &= ~(1ull << b);
            VCR_LDS_FETCH(A, b);
            for (;;) {
                int nb = m ? 63 - __builtin_clzll(m) : 0;
                bool more = m != 0;
                m &= ~(1ull << nb);
                VCR_LDS_FETCH(B, nb);
                VCR_SHADE_BWD(A, b);
                if (!more) break;
                b = nb;
                nb = m ? 63 - __builtin_clzll(m) : 0;
                more = m != 0;
                m &= ~(1ull << nb);
                VCR_LDS_FETCH(A, nb);
                VCR_SHADE_BWD(B, b);
                if (!more) break;
                b = nb;
            }
        } else if (many && !(VCR_KO & 1)) {
            unsigned long long mrow = row == 0 ? mr[0] : (row == 1 ? mr[1] : (row == 2 ? mr[2] : mr[3]));
            // a row whose list is exhausted keeps shading a staged (finite) record with hit = false
            int b = 63 - __builtin_clzll(many);
            bool act;
#define VCR_ROW_NEXT(B, ACT)                                                           \
            do {                                                                        \
                ACT = mrow != 0;                                                        \
                B = ACT ? 63 - __builtin_clzll(mrow) : B;                               \
                mrow = ACT ? mrow & ~(1ull << B) : 0ull;                                \
            } while (0)
#define VCR_SHADE_BWD_ROWS(R, B, ACT)                                                                                     \
            do {                                                                                                           \
                const float4 r0 = R##0, r1 = R##1, r2 = R##2, r3 = R##3;                                                   \
                const uint32_t idx1 = VCR_BWD_IDX1(B);                                                                     \
                const f2 gxy = {r0.x, r0.y}, sAC = {r0.z, r0.w};                                                           \
                const f2 d = gxy - fxy;                                                                                    \
                f2 u; float hs;                                                                                            \
                const float e = gauss_exponent(d, sAC, r1.x, r1.y, u, hs);                                                 \
                const float araw = __builtin_amdgcn_exp2f(e);                                                              \
                const bool hit = (ACT) && idx1 <= lastc && hs <= 0.f && araw >= VCR_ALPHA_MIN;                             \
                const unsigned long long hitm_ = __builtin_amdgcn_ballot_w64(hit);                                         \
                if (hitm_ != 0) {                                                                                          \
                    VCR_BWD_MATH(R);                                                                                       \
                    const float tot = row_reduce16(v, lane);                                                               \
                    if (tot != 0.f) VCR_GRAD_ATOMIC(gid, lane & 15, tot);                                                  \
                    if (S > 0) {                                                                                           \
                        float ts = 0.f;                                                                                    \
_Pragma("unroll")                                                                                                          \
                        for (int k = 0; k < S; ++k) { const float t = row_sum16(vs[k]); ts = (lane & 15) == k ? t : ts; }  \
                        if ((lane & 15) < S && ts != 0.f) atomicAdd(sgrad_sem + (size_t)gid * S + (lane & 15), ts);        \
                    }                                                                                                      \
                }                                                                                                          \
            } while (0)
            float4 A0, A1, A2, A3, A4 = {0.f, 0.f, 0.f, 0.f}, B0, B1, B2, B3, B4 = {0.f, 0.f, 0.f, 0.f};
            int nb = b; bool nact;
            VCR_ROW_NEXT(b, act);
            VCR_LDS_FETCH(A, b);
            for (int it = 0;;) {
                nb = b;
                VCR_ROW_NEXT(nb, nact);
                VCR_LDS_FETCH(B, nb);
                VCR_SHADE_BWD_ROWS(A, b, act);
                if (++it >= iters) break;
                b = nb;
                VCR_ROW_NEXT(b, act);
                VCR_LDS_FETCH(A, b);
                VCR_SHADE_BWD_ROWS(B, nb, nact);
                if (++it >= iters) break;
            }
        }
        if (VCR_T_ANCHOR && chunk > 0) { T = ckT; Td = (double)ckT; }
        id = nid; q0 = nq0; q1 = nq1; q2 = nq2; q3 = nq3; qs = nqs; valid = nvalid; nid = nnid; nvalid = nnvalid;
    }
}

// ---- the same work item with the survivors GROUPED across chunks (round 6) ------------------------------------------------------------
// bwd_rows_item shades what one 64-entry chunk leaves: ~16 survivors, each in ~1.9 of the four row lists, so the row-packed loop runs
// max_r |list_r| ~ 10 iterations for a mean of 7.6 (a quarter of its row-slots idle) and pays its prologue -- masks, the choice between
// the two loops, the first LDS fetch -- per chunk.  Here the survivors of consecutive chunks are compacted, back to front, into the 64
// staging slots (slot 0 = the LAST list entry of the group) and shaded when the next chunk's survivors would not fit: the row lists
// are ~4x longer and their maximum is closer to their mean.  A slot carries its 1-based list position and its four row bits in the
// pad word of plane 3.  Every pixel still meets its contributors in list order, back to front, with the same arithmetic: T, the
// suffix sums and every per-pair value are those of bwd_rows_item; only the order of the fp32 atomics differs.
// Measured (profiles/r6_bwd_group_ab.txt, 4 cameras): metric 396.8 -> 389.2 us, c5 925.9 -> 892.2, dense 634.7 -> 613.8, full-frame
// 751.2 -> 727.1, c2 288.9 -> 283.6.  A second form that tested every entry once per chunk (bounding box of all live pixels) and ran the
// four sub-block tests once per group on the compacted survivors was NOT faster (metric 400.5, per-quad lists 396 against 382): the
// wider boxes let more survivors through than the three saved tests per chunk are worth.
template <int S, bool ISECT, int ND, bool QL>
__device__ __forceinline__ void bwd_group_item(const VcrRasterArgs& a, const GeomRec* __restrict__ rec, const float* __restrict__ semv,
                                              const uint32_t* __restrict__ point_list, const uint2* __restrict__ ranges, int gxc,
                                              const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                              const float* __restrict__ moments, const float* __restrict__ ckpt,
                                              const float* __restrict__ dL_dout, GradRec* __restrict__ sgrad,
                                              float* __restrict__ sgrad_sem, int rows_bias, int rows_pair_cost,
                                              int tile, int sub, int wv, float4* const srec) {
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const PixelMap pm = pixel_of_thread_rows(tile, gx, a.W, a.H, sub, wv);
    const uint2 range = list_range<QL>(ranges, gxc, tile, gx, wv);
    if (range.x == range.y) return;                        // wave-uniform: nothing reaches this quad
    static_assert(!VCR_T_ANCHOR, "the grouped backward has no per-chunk transmittance anchoring");
    (void)ckpt;
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63, row = lane >> 4;
    constexpr bool SEM_IN_REC = false;
    constexpr int FC = 0;
    const float X0 = (float)((tile % gx) * VCR_TILE + (wv & 1) * 8);
    const float Y0 = (float)((tile / gx) * VCR_TILE + (wv >> 1) * 8);
    const f2 fxy = {(float)pm.x, (float)pm.y};
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float g[8 + (S > 0 ? S : 0)];
#pragma unroll
    for (int c = 0; c < 8 + S; ++c) g[c] = pm.inside ? dL_dout[c * (size_t)P + pm.pix] : 0.f;
    float gm2 = 0.f;
    if (ND == 2 && pm.inside) { g[3] += dL_dout[(8 + S) * (size_t)P + pm.pix]; gm2 = dL_dout[(9 + S) * (size_t)P + pm.pix]; }
    const float Tf = pm.inside ? final_T[pm.pix] : 1.f;
    float gm1 = 0.f;
    const float zc_map = VCR_ZFAR / (VCR_ZFAR - VCR_ZNEAR);
    if (ND == 1 && pm.inside) {
        const float gd = dL_dout[(8 + S) * (size_t)P + pm.pix];
        const float m1 = moments[pm.pix], m2 = moments[P + pm.pix];
        gm1 = -2.f * m1 * gd; gm2 = (1.f - Tf) * gd; g[7] += m2 * gd;
    }
    const uint32_t lastc = pm.inside ? n_contrib[pm.pix] : 0u;
    const float bgdot = Tf * (a.bg[0] * g[0] + a.bg[1] * g[1] + a.bg[2] * g[2]);
    uint32_t maxc = lastc;
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, o));
    if (maxc == 0) return;                                 // wave-uniform
    float T = Tf, Bsuf = bgdot;
    double Td = (double)Tf, Bd = (double)bgdot;      // (VCR_DBG_PIX64 only; dead otherwise)
    (void)Td; (void)Bd;
    const f2 g01 = {g[0], g[1]}, g24 = {g[2], g[4]}, g56 = {g[5], g[6]}, ryz = {ry, rz};

    // 1-based list position of the staged survivor being shaded (28 bits of the pad word of `r3`; the row bits sit above them)
#undef VCR_BWD_IDX1
#define VCR_BWD_IDX1(SLOT) (__float_as_uint(r3.w) & 0x0FFFFFFFu)
#undef VCR_ROW_NEXT
#define VCR_ROW_NEXT(B, ACT)                                                           \
            do {                                                                        \
                ACT = mrow != 0;                                                        \
                B = ACT ? __builtin_ctzll(mrow) : B;                                    \
                mrow = ACT ? mrow & (mrow - 1ull) : 0ull;                               \
            } while (0)
    // shade the `fill` staged survivors (slot order = back to front) and empty the group
#define VCR_BWD_FLUSH()                                                                                              \
    do {                                                                                                             \
        __builtin_amdgcn_wave_barrier();                                                                             \
        const uint32_t rb_ = lane < fill ? __float_as_uint(srec[3 * 64 + lane].w) >> 28 : 0u;                        \
        unsigned long long mr[4];                                                                                    \
_Pragma("unroll")                                                                                                    \
        for (int r = 0; r < 4; ++r) mr[r] = __builtin_amdgcn_ballot_w64((rb_ >> r) & 1u);                            \
        const unsigned long long many = mr[0] | mr[1] | mr[2] | mr[3];                                               \
        const int iters = max(max(__popcll(mr[0]), __popcll(mr[1])), max(__popcll(mr[2]), __popcll(mr[3])));         \
        const int pairs = __popcll(mr[0]) + __popcll(mr[1]) + __popcll(mr[2]) + __popcll(mr[3]);                     \
        if (many && iters * 16 + pairs * rows_pair_cost > __popcll(many) * rows_bias) {                              \
            unsigned long long m = many;                                                                             \
            float4 A0, A1, A2, A3, A4 = {0.f, 0.f, 0.f, 0.f}, B0, B1, B2, B3, B4 = {0.f, 0.f, 0.f, 0.f};             \
            int b = __builtin_ctzll(m);                                                                              \
            m &= m - 1ull;                                                                                           \
            VCR_LDS_FETCH(A, b);                                                                                     \
            for (;;) {                                                                                               \
                int nb = m ? __builtin_ctzll(m) : 0;                                                                 \
                bool more = m != 0;                                                                                  \
                m &= m - 1ull;                                                                                       \
                VCR_LDS_FETCH(B, nb);                                                                                \
                VCR_SHADE_BWD(A, b);                                                                                 \
                if (!more) break;                                                                                    \
                b = nb;                                                                                              \
                nb = m ? __builtin_ctzll(m) : 0;                                                                     \
                more = m != 0;                                                                                       \
                m &= m - 1ull;                                                                                       \
                VCR_LDS_FETCH(A, nb);                                                                                \
                VCR_SHADE_BWD(B, b);                                                                                 \
                if (!more) break;                                                                                    \
                b = nb;                                                                                              \
            }                                                                                                        \
        } else if (many) {                                                                                           \
            unsigned long long mrow = row == 0 ? mr[0] : (row == 1 ? mr[1] : (row == 2 ? mr[2] : mr[3]));            \
            int b = __builtin_ctzll(many);                                                                           \
            bool act;                                                                                                \
            float4 A0, A1, A2, A3, A4 = {0.f, 0.f, 0.f, 0.f}, B0, B1, B2, B3, B4 = {0.f, 0.f, 0.f, 0.f};             \
            int nb = b; bool nact;                                                                                   \
            VCR_ROW_NEXT(b, act);                                                                                    \
            VCR_LDS_FETCH(A, b);                                                                                     \
            for (int it = 0;;) {                                                                                     \
                nb = b;                                                                                              \
                VCR_ROW_NEXT(nb, nact);                                                                              \
                VCR_LDS_FETCH(B, nb);                                                                                \
                VCR_SHADE_BWD_ROWS(A, b, act);                                                                       \
                if (++it >= iters) break;                                                                            \
                b = nb;                                                                                              \
                VCR_ROW_NEXT(b, act);                                                                                \
                VCR_LDS_FETCH(A, b);                                                                                 \
                VCR_SHADE_BWD_ROWS(B, nb, nact);                                                                     \
                if (++it >= iters) break;                                                                            \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                             \
        fill = 0;                                                                                                    \
    } while (0)
    int fill = 0;                                          // staged survivors of the current group (wave-uniform)
    int chunk = (int)((maxc - 1) / 64);
    uint32_t id, nid; float4 q0, q1, q2, q3, qs = {0.f, 0.f, 0.f, 0.f}; bool valid, nvalid;
    const uint32_t lim = range.x + maxc;
    VCR_LOAD_ID(range.x + (uint32_t)chunk * 64u + lane, lim, id, valid);
    VCR_GATHER_REC(id, q0, q1, q2, q3);
    VCR_GATHER_SEM(id, qs);
    VCR_LOAD_ID(chunk > 0 ? range.x + (uint32_t)(chunk - 1) * 64u + lane : lim, lim, nid, nvalid);
    for (; chunk >= 0; --chunk) {
        uint32_t nnid; float4 nq0, nq1, nq2, nq3, nqs = {0.f, 0.f, 0.f, 0.f}; bool nnvalid;
        VCR_GATHER_REC(nid, nq0, nq1, nq2, nq3);
        VCR_GATHER_SEM(nid, nqs);
        VCR_LOAD_ID(chunk > 1 ? range.x + (uint32_t)(chunk - 2) * 64u + lane : lim, lim, nnid, nnvalid);
        // cull against the live box of each 4x4 sub-block (as bwd_rows_item); the four row bits travel with the staged record
        const unsigned long long live = __builtin_amdgcn_ballot_w64(lastc > (uint32_t)chunk * 64u);
        uint32_t rbits = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned lr = (unsigned)(live >> (16 * r)) & 0xFFFFu;
            if (lr != 0) {                                   // wave-uniform
                float bx0, by0, bw, bh;
                live_box16(lr, X0 + (float)((r & 1) * 4), Y0 + (float)((r >> 1) * 4), bx0, by0, bw, bh);
                if (valid && quad_touch(q0, q1, bx0, by0, bw, bh)) rbits |= 1u << r;
            }
        }
        const unsigned long long km = __builtin_amdgcn_ballot_w64(rbits != 0);
        const int cnt = __popcll(km);
        if (fill + cnt > 64) VCR_BWD_FLUSH();                // (wave-uniform)
        if (rbits != 0) {
            // back to front: the chunk's highest lane is its last entry and takes the lowest free slot
            const int slot = fill + (lane == 63 ? 0 : __popcll(km >> (lane + 1)));
            srec[0 * 64 + slot] = make_float4(q0.x, q0.y, -VCR_L2E * q1.x, -VCR_L2E * q1.z);
            srec[1 * 64 + slot] = make_float4(-VCR_L2E * q1.y, __builtin_amdgcn_logf(q0.w), q0.z, q1.w);
            srec[2 * 64 + slot] = make_float4(q2.x, q2.y, q2.z, q3.x);
            srec[3 * 64 + slot] = make_float4(q3.y, q3.z, __uint_as_float(id), __uint_as_float((rbits << 28) | ((uint32_t)chunk * 64u + (uint32_t)lane + 1u)));
            if (S > 0) srec[4 * 64 + slot] = qs;
        }
        fill += cnt;
        id = nid; q0 = nq0; q1 = nq1; q2 = nq2; q3 = nq3; qs = nqs; valid = nvalid; nid = nnid; nvalid = nnvalid;
    }
    if (fill > 0) VCR_BWD_FLUSH();
#undef VCR_BWD_FLUSH
#undef VCR_BWD_IDX1
}

#define VCR_BWD_WREC (S > 0 ? 320 : 256)
template <int S, bool ISECT, int ND, bool QL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VCR_ROWS_WAVES))) composite_bwd_rows_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                               const float* __restrict__ semv,
                                                               const uint32_t* __restrict__ point_list,
                                                               const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ meta,
                                                               int num_tiles, int gxc, const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ moments, const float* __restrict__ ckpt,
                                                               const float* __restrict__ dL_dout, GradRec* __restrict__ sgrad,
                                                               float* __restrict__ sgrad_sem, int rows_bias, int rows_pair_cost, int det_sel) {
#ifdef VCR_DETERMINISTIC_BWD
    // test-only build: the launcher issues one launch per (workgroup, wave) and every other wave leaves at once, so the fp32
    // atomics of the whole backward happen in ONE fixed order (tile order, quad 0..3, list back to front)
    if ((int)(blockIdx.x * 4 + (threadIdx.x >> 6)) != det_sel) return;
#else
    (void)det_sel;
#endif
    int sub;
    const int tile = work_item(tile_order, meta, num_tiles, sub);
    if (tile < 0) return;
    const int wv = threadIdx.x >> 6;
    __shared__ float4 s_rec_all[4 * VCR_BWD_WREC];
    bwd_rows_item<S, ISECT, ND, QL>(a, rec, semv, point_list, ranges, gxc, final_T, n_contrib, moments, ckpt, dL_dout, sgrad, sgrad_sem,
                                    rows_bias, rows_pair_cost, tile, sub, wv, s_rec_all + wv * VCR_BWD_WREC);
}

template <int S, bool ISECT, int ND, bool QL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VCR_ROWS_WAVES))) composite_bwd_group_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                               const float* __restrict__ semv,
                                                               const uint32_t* __restrict__ point_list,
                                                               const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ meta,
                                                               int num_tiles, int gxc, const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ moments, const float* __restrict__ ckpt,
                                                               const float* __restrict__ dL_dout, GradRec* __restrict__ sgrad,
                                                               float* __restrict__ sgrad_sem, int rows_bias, int rows_pair_cost, int det_sel) {
#ifdef VCR_DETERMINISTIC_BWD
    if ((int)(blockIdx.x * 4 + (threadIdx.x >> 6)) != det_sel) return;
#else
    (void)det_sel;
#endif
    int sub;
    const int tile = work_item(tile_order, meta, num_tiles, sub);
    if (tile < 0) return;
    const int wv = threadIdx.x >> 6;
    __shared__ float4 s_rec_all[4 * VCR_BWD_WREC];
    bwd_group_item<S, ISECT, ND, QL>(a, rec, semv, point_list, ranges, gxc, final_T, n_contrib, moments, ckpt, dL_dout, sgrad, sgrad_sem,
                                     rows_bias, rows_pair_cost, tile, sub, wv, s_rec_all + wv * VCR_BWD_WREC);
}

template <int S, bool ISECT, int ND>
int launch_fwd_fc(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o, int tiles,
                  hipStream_t st, bool two_phase) {
    const int gxc = a.quad_lists ? 2 * ((a.W + VCR_TILE - 1) / VCR_TILE) : 0;
#define VCR_FWD_Q(FC, NDD, Q)                                                                                     \
    hipLaunchKernelGGL((composite_fwd_v2_kernel<S, ISECT, FC, NDD, Q>), dim3(tiles + 3 * VCR_SPLIT_MAX), dim3(256), 0, st, a, g.rec, g.sem, \
                       b.point_list, b.ranges, b.tile_order, b.meta, tiles, gxc, im.final_T, im.n_contrib, im.moments, o.out, o.count, o.score, im.t_ckpt)
#define VCR_FWD(FC, NDD) do { if (gxc) { VCR_FWD_Q(FC, NDD, true); } else { VCR_FWD_Q(FC, NDD, false); } } while (0)
    if constexpr (VCR_FWD_TP != 0 && S <= 2) {
        if (a.f_count == 0 && two_phase) {   // the training / evaluation render of small footprints (the count modes keep the v2 loop)
            if (gxc)
                hipLaunchKernelGGL((composite_fwd_tp_kernel<S, ISECT, ND, true>), dim3(tiles + 3 * VCR_SPLIT_MAX), dim3(256), 0, st, a, g.rec,
                                   b.point_list, b.ranges, b.tile_order, b.meta, tiles, gxc, im.final_T, im.n_contrib, im.moments, o.out);
            else
                hipLaunchKernelGGL((composite_fwd_tp_kernel<S, ISECT, ND, false>), dim3(tiles + 3 * VCR_SPLIT_MAX), dim3(256), 0, st, a, g.rec,
                                   b.point_list, b.ranges, b.tile_order, b.meta, tiles, gxc, im.final_T, im.n_contrib, im.moments, o.out);
            VCR_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    switch (a.f_count) {
        case 0: VCR_FWD(0, ND); break;
        case 1: case 2: VCR_FWD(1, 0); break;
        case 4: VCR_FWD(4, 0); break;
        default: VCR_FWD(3, 0); break;
    }
#undef VCR_FWD
#undef VCR_FWD_Q
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool ISECT, int ND>
int launch_fwd_s(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o, int tiles,
                 hipStream_t st, bool two_phase) {
    switch (a.S) {
        case 0: return launch_fwd_fc<0, ISECT, ND>(a, g, b, im, o, tiles, st, two_phase);
        case 1: return launch_fwd_fc<1, ISECT, ND>(a, g, b, im, o, tiles, st, two_phase);
        case 2: return launch_fwd_fc<2, ISECT, ND>(a, g, b, im, o, tiles, st, two_phase);
        case 3: return launch_fwd_fc<3, ISECT, ND>(a, g, b, im, o, tiles, st, two_phase);
        default: return launch_fwd_fc<4, ISECT, ND>(a, g, b, im, o, tiles, st, two_phase);
    }
}

template <bool ISECT, int ND>
int launch_bwd_s(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, const float* dL_dout, GradRec* sgrad,
                 float* sgrad_sem, int tiles, hipStream_t st) {
    const int gxc = a.quad_lists ? 2 * ((a.W + VCR_TILE - 1) / VCR_TILE) : 0;
    // per chunk: row-packed loop iff 16 * max_r |list_r| + pair * sum_r |list_r| <= bias * |union_r list_r| (the second term
    // prices the atomics: an iteration with all four rows live issues 64 atomic lanes instead of 16).  bias 0 = never,
    // (64, 0) = always; (20, 2) from the sweep in profiles/r3_bwd_rows_ab.txt
    // (re-swept for the grouped form in round 6, profiles/r6_rows_sweep.txt: (20, 2) again -- (28, 2) gains 1 % at the metric scene and
    //  loses 36 % on the full-frame variant, (16, 2) loses 13 %)
    constexpr int rows_bias = 20, rows_pair_cost = 2;
#ifdef VCR_DETERMINISTIC_BWD
    const int det_first = 0, det_last = 4 * (tiles + 3 * VCR_SPLIT_MAX);      // one launch per (workgroup, wave), in order
#else
    const int det_first = -1, det_last = 0;
#endif
#define VCR_BWD_Q(SS, Q)                                                                                         \
    for (int det = det_first; det < det_last; ++det)                                                             \
        hipLaunchKernelGGL((composite_bwd_rows_kernel<SS, ISECT, ND, Q>), dim3(tiles + 3 * VCR_SPLIT_MAX), dim3(256), 0, st, a, g.rec, g.sem, \
                           b.point_list, b.ranges, b.tile_order, b.meta, tiles, gxc, im.final_T, im.n_contrib, im.moments, im.t_ckpt, dL_dout, sgrad, sgrad_sem, rows_bias, rows_pair_cost, det);
#define VCR_BWD_GQ(SS, Q)                                                                                        \
    for (int det = det_first; det < det_last; ++det)                                                             \
        hipLaunchKernelGGL((composite_bwd_group_kernel<SS, ISECT, ND, Q>), dim3(tiles + 3 * VCR_SPLIT_MAX), dim3(256), 0, st, a, g.rec, g.sem, \
                           b.point_list, b.ranges, b.tile_order, b.meta, tiles, gxc, im.final_T, im.n_contrib, im.moments, im.t_ckpt, dL_dout, sgrad, sgrad_sem, rows_bias, rows_pair_cost, det);
    constexpr bool grouped = !VCR_T_ANCHOR;                 // (anchored experiment builds keep the per-chunk form: checkpoints are per chunk)
#define VCR_BWD(SS) do { if (grouped) { if (gxc) { VCR_BWD_GQ(SS, true) } else { VCR_BWD_GQ(SS, false) } }           \
                         else if (gxc) { VCR_BWD_Q(SS, true) } else { VCR_BWD_Q(SS, false) } } while (0)
    switch (a.S) {
        case 0: VCR_BWD(0); break;
        case 1: VCR_BWD(1); break;
        case 2: VCR_BWD(2); break;
        case 3: VCR_BWD(3); break;
        default: VCR_BWD(4); break;
    }
#undef VCR_BWD
#undef VCR_BWD_GQ
#undef VCR_BWD_Q
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

#ifdef VCR_DBG_ACC64
namespace {
__global__ void acc64_to_sgrad_kernel(int N, const double* __restrict__ acc, float* __restrict__ sgrad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)N * 16) sgrad[i] = (float)acc[i];
}
double* g_acc64_host = nullptr;
}
int vcr_dbg_acc64_begin(int N, hipStream_t st) {
    VCR_HIP_CHECK(hipStreamSynchronize(st));
    if (g_acc64_host) { VCR_HIP_CHECK(hipFree(g_acc64_host)); g_acc64_host = nullptr; }
    VCR_HIP_CHECK(hipMalloc((void**)&g_acc64_host, sizeof(double) * 16 * (size_t)N));
    VCR_HIP_CHECK(hipMemsetAsync(g_acc64_host, 0, sizeof(double) * 16 * (size_t)N, st));
    VCR_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_acc64), &g_acc64_host, sizeof(double*), 0, hipMemcpyHostToDevice, st));
    return 0;
}
int vcr_dbg_acc64_end(int N, GradRec* sgrad, hipStream_t st) {
    hipLaunchKernelGGL(acc64_to_sgrad_kernel, dim3((unsigned)(((size_t)N * 16 + 255) / 256)), dim3(256), 0, st, N, g_acc64_host,
                       reinterpret_cast<float*>(sgrad));
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif

extern "C" int vcr_debug_hit_histogram(uint32_t out[130], int reset) {
#ifdef VCR_TPSTATS
    VCR_HIP_CHECK(hipDeviceSynchronize());
    unsigned long long st[16];
    VCR_HIP_CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tpstats), sizeof(st)));
    for (int k = 0; k < 130; ++k) out[k] = 0;
    for (int k = 0; k < 16; ++k) { out[2 * k] = (uint32_t)st[k]; out[2 * k + 1] = (uint32_t)(st[k] >> 32); }
    if (reset) {
        unsigned long long z[16] = {0};
        VCR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tpstats), z, sizeof(z)));
    }
    return 0;
#elif defined(VCR_HITHIST)
    VCR_HIP_CHECK(hipDeviceSynchronize());
    VCR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hithist), sizeof(unsigned int) * 130));
    if (reset) {
        unsigned int z[130] = {0};
        VCR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_hithist), z, sizeof(z)));
    }
    return 0;
#else
    (void)out; (void)reset;
    vcr_set_error("vcr_debug_hit_histogram: this library was built without -DVCR_HITHIST");
    return 1;
#endif
}

// Which form of the forward shades this frame: the two-phase kernel wins where footprints are small (metric scene, R / V = 2.8
// tiles per visible Gaussian: 183 against 212 us; c5: R / V = 1.9) and loses where a survivor covers a large part of its quad
// (dense variant, R / V = 10.5: 289 against 274 us; full-frame variant: 338 against 292) -- profiles/r6_fwd_variants.txt.  Both
// kernels write bit-identical images and image state, so the choice is invisible to the backward.
bool vcr_forward_two_phase(int64_t tile_instances, int64_t visible) {
    return VCR_FWD_TP != 0 && !VCR_T_ANCHOR && tile_instances <= (int64_t)VCR_TP_MAX_TILES_PER_GAUSSIAN * visible;
}

int vcr_launch_composite_forward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o,
                                 hipStream_t st, bool two_phase) {
    const int tiles = ((a.W + VCR_TILE - 1) / VCR_TILE) * ((a.H + VCR_TILE - 1) / VCR_TILE);
    const bool isect = a.dirs != nullptr && a.normals_precomp != nullptr;
    if (a.num_dist == 2)
        return isect ? launch_fwd_s<true, 2>(a, g, b, im, o, tiles, st, two_phase) : launch_fwd_s<false, 2>(a, g, b, im, o, tiles, st, two_phase);
    if (a.num_dist == 1)
        return isect ? launch_fwd_s<true, 1>(a, g, b, im, o, tiles, st, two_phase) : launch_fwd_s<false, 1>(a, g, b, im, o, tiles, st, two_phase);
    return isect ? launch_fwd_s<true, 0>(a, g, b, im, o, tiles, st, two_phase) : launch_fwd_s<false, 0>(a, g, b, im, o, tiles, st, two_phase);
}

int vcr_launch_composite_backward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, const float* dL_dout,
                                  GradRec* sgrad, float* sgrad_sem, hipStream_t st) {
    const int tiles = ((a.W + VCR_TILE - 1) / VCR_TILE) * ((a.H + VCR_TILE - 1) / VCR_TILE);
    const bool isect = a.dirs != nullptr && a.normals_precomp != nullptr;
    if (a.num_dist == 2)
        return isect ? launch_bwd_s<true, 2>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st)
                     : launch_bwd_s<false, 2>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st);
    if (a.num_dist == 1)
        return isect ? launch_bwd_s<true, 1>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st)
                     : launch_bwd_s<false, 1>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st);
    return isect ? launch_bwd_s<true, 0>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st)
                 : launch_bwd_s<false, 0>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st);
}

