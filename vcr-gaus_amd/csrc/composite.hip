// K6 / K7: per-tile front-to-back alpha compositing and its adjoint for gfx950 (wave64).
//
// Contract: SURVEY.md Appendix A.3 step 9-10 / A.4, restated in oracle/raster_torch.py::composite_tile.
// Output channel layout consumed by gaussian_renderer/__init__.py:122-123,150,155-161:
//   colour3 | depth1 | camera-space normal3 | alpha1 | semantics S.
//
// Geometry: one 256-thread workgroup per 16x16 tile; each 64-lane wave owns one 8x8 pixel quad so that
// the early-out (T < 1e-4) and the "nobody in this wave is touched" skip are wave-uniform.  The tile's
// depth-ordered Gaussian list is staged through LDS in 256-record batches (one 64-byte GeomRec gather per
// lane); the per-pixel loop then reads each record as an LDS broadcast.
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

// Reduce 16 per-lane values over the wave with a halving butterfly: v_permlane32_swap / v_permlane16_swap
// exchange half of the live values per step (8+4 swaps), then 4 row rotations finish the remaining 4.
// On return lane l holds, in out[0..3], the wave totals of v[8*(l>>5) + 4*((l>>4)&1) + 0..3].
__device__ __forceinline__ void wave_reduce16(const float v[16], float out[4]) {
    float u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[8 + k]), false, false);
        u[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[k]), __float_as_uint(u[4 + k]), false, false);
        float x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        x = dpp_add<0x128>(x);
        x = dpp_add<0x124>(x);
        x = dpp_add<0x122>(x);
        x = dpp_add<0x121>(x);
        out[k] = x;
    }
}

struct PixelMap {
    int x, y, pix;
    bool inside;
};

__device__ __forceinline__ PixelMap pixel_of_thread(int tile, int gx, int W, int H) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    PixelMap p;
    p.x = (tile % gx) * VCR_TILE + (wv & 1) * 8 + (lane & 7);
    p.y = (tile / gx) * VCR_TILE + (wv >> 1) * 8 + (lane >> 3);
    p.inside = p.x < W && p.y < H;
    p.pix = p.y * W + p.x;
    return p;
}

template <int S, bool ISECT, int FC>
__global__ void __launch_bounds__(256) composite_fwd_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                            const float* __restrict__ semv,
                                                            const uint32_t* __restrict__ point_list,
                                                            const uint2* __restrict__ ranges, float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                                            int32_t* __restrict__ count, float* __restrict__ score) {
    __shared__ float4 s_q[4][256];
    __shared__ uint32_t s_id[256];
    __shared__ float s_sem[(S > 0 ? S : 1) * 256];
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const int tile = blockIdx.x;
    const PixelMap pm = pixel_of_thread(tile, gx, a.W, a.H);
    const uint2 range = ranges[tile];
    const int P = a.H * a.W;
    const float fx = (float)pm.x, fy = (float)pm.y;
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float T = 1.f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, A = 0.f;
    float SM[S > 0 ? S : 1];
#pragma unroll
    for (int k = 0; k < S; ++k) SM[k] = 0.f;
    uint32_t contributor = 0, last = 0;
    bool done = !pm.inside;

    for (uint32_t base = range.x; base < range.y; base += 256) {
        if (__syncthreads_and(done)) break;
        const uint32_t n = min(256u, range.y - base);
        if (threadIdx.x < n) {
            const uint32_t id = point_list[base + threadIdx.x];
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            s_q[0][threadIdx.x] = src[0]; s_q[1][threadIdx.x] = src[1];
            s_q[2][threadIdx.x] = src[2]; s_q[3][threadIdx.x] = src[3];
            s_id[threadIdx.x] = id;
#pragma unroll
            for (int k = 0; k < S; ++k) s_sem[k * 256 + threadIdx.x] = semv[(size_t)id * S + k];
        }
        __syncthreads();
        for (uint32_t j = 0; j < n; ++j) {
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            const float4 q0 = s_q[0][j], q1 = s_q[1][j];
            contributor++;
            const float dx = q0.x - fx, dy = q0.y - fy;
            const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
            const float alpha = fminf(VCR_ALPHA_MAX, q0.w * __expf(power));
            bool hit = !done && power <= 0.f && alpha >= VCR_ALPHA_MIN;
            const float test_T = T * (1.f - alpha);
            if (hit && test_T < VCR_T_EPS) { done = true; hit = false; }
            if (FC != 0) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m) {
                    const float ws = wave_sum(hit ? alpha * T : 0.f);
                    if ((threadIdx.x & 63) == 0) {
                        atomicAdd(count + s_id[j], (int)__popcll(m));
                        if (FC != 3) atomicAdd(score + s_id[j], ws);
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
            if (hit) {
                const float4 q2 = s_q[2][j], q3 = s_q[3][j];
                const float w = alpha * T;
                float dep = q0.z;
                if (ISECT) {
                    const float den = q3.x * rx + q3.y * ry + q3.z * rz;
                    if (den > VCR_PLANE_EPS) dep = q1.w / den * rz;
                }
                C0 += w * q2.x; C1 += w * q2.y; C2 += w * q2.z;
                D += w * dep;
                N0 += w * q3.x; N1 += w * q3.y; N2 += w * q3.z;
                A += w;
#pragma unroll
                for (int k = 0; k < S; ++k) SM[k] += w * s_sem[k * 256 + j];
                T = test_T;
                last = contributor;
            }
        }
    }
    if (pm.inside) {
        final_T[pm.pix] = T;
        n_contrib[pm.pix] = last;
        if (FC != 3) {
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
        }
    }
}

template <int S, bool ISECT>
__global__ void __launch_bounds__(256) composite_bwd_kernel(VcrRasterArgs a, const GeomRec* __restrict__ rec,
                                                            const float* __restrict__ semv,
                                                            const uint32_t* __restrict__ point_list,
                                                            const uint2* __restrict__ ranges,
                                                            const float* __restrict__ final_T,
                                                            const uint32_t* __restrict__ n_contrib,
                                                            const float* __restrict__ dL_dout, GradRec* __restrict__ sgrad,
                                                            float* __restrict__ sgrad_sem) {
    __shared__ float4 s_q[4][256];
    __shared__ uint32_t s_id[256];
    __shared__ float s_sem[(S > 0 ? S : 1) * 256];
    __shared__ float s_acc[VCR_GRAD_FLOATS + (S > 0 ? S : 0)][256];
    __shared__ uint32_t s_max;
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE;
    const int tile = blockIdx.x;
    const PixelMap pm = pixel_of_thread(tile, gx, a.W, a.H);
    const uint2 range = ranges[tile];
    const int P = a.H * a.W;
    const int lane = threadIdx.x & 63;
    const float fx = (float)pm.x, fy = (float)pm.y;
    float rx = 0.f, ry = 0.f, rz = 1.f;
    if (ISECT && pm.inside) { rx = a.dirs[pm.pix]; ry = a.dirs[P + pm.pix]; rz = a.dirs[2 * P + pm.pix]; }

    float g[8 + (S > 0 ? S : 0)];
#pragma unroll
    for (int c = 0; c < 8 + S; ++c) g[c] = pm.inside ? dL_dout[c * (size_t)P + pm.pix] : 0.f;
    const float Tf = pm.inside ? final_T[pm.pix] : 1.f;
    const uint32_t lastc = pm.inside ? n_contrib[pm.pix] : 0u;
    const float bgdot = Tf * (a.bg[0] * g[0] + a.bg[1] * g[1] + a.bg[2] * g[2]);

    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    atomicMax(&s_max, lastc);
    __syncthreads();
    const uint32_t maxc = s_max;              // deepest contributor index (1-based) in this tile
    float T = Tf;
    float Asuf = 0.f;                         // sum_{j behind i} w_j (f_j . g)

    const int nbatch = (int)((maxc + 255) / 256);
    for (int b = nbatch - 1; b >= 0; --b) {
        const uint32_t base = range.x + (uint32_t)b * 256u;
        const uint32_t n = min(256u, min(range.y - base, maxc - (uint32_t)b * 256u));
        __syncthreads();
        if (threadIdx.x < n) {
            const uint32_t id = point_list[base + threadIdx.x];
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            s_q[0][threadIdx.x] = src[0]; s_q[1][threadIdx.x] = src[1];
            s_q[2][threadIdx.x] = src[2]; s_q[3][threadIdx.x] = src[3];
            s_id[threadIdx.x] = id;
#pragma unroll
            for (int k = 0; k < S; ++k) s_sem[k * 256 + threadIdx.x] = semv[(size_t)id * S + k];
        }
#pragma unroll
        for (int k = 0; k < VCR_GRAD_FLOATS + S; ++k) s_acc[k][threadIdx.x] = 0.f;
        __syncthreads();
        for (int j = (int)n - 1; j >= 0; --j) {
            const uint32_t idx1 = (uint32_t)b * 256u + (uint32_t)j + 1u;
            const float4 q0 = s_q[0][j], q1 = s_q[1][j];
            const float dx = q0.x - fx, dy = q0.y - fy;
            const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
            const float G = __expf(power);
            const float araw = q0.w * G;
            const float alpha = fminf(VCR_ALPHA_MAX, araw);
            const bool hit = idx1 <= lastc && power <= 0.f && alpha >= VCR_ALPHA_MIN;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = 0.f;
            float vs[S > 0 ? S : 1];
#pragma unroll
            for (int k = 0; k < S; ++k) vs[k] = 0.f;
            if (hit) {
                const float4 q2 = s_q[2][j], q3 = s_q[3][j];
                const float inv1ma = 1.f / (1.f - alpha);
                T *= inv1ma;                                   // transmittance in front of this Gaussian
                const float w = alpha * T;
                float dep = q0.z;
                float den = 1.f;
                bool isect = false;
                if (ISECT) {
                    den = q3.x * rx + q3.y * ry + q3.z * rz;
                    isect = den > VCR_PLANE_EPS;
                    if (isect) dep = q1.w / den * rz;
                }
                float fg = q2.x * g[0] + q2.y * g[1] + q2.z * g[2] + dep * g[3] + q3.x * g[4] + q3.y * g[5] +
                           q3.z * g[6] + g[7];
#pragma unroll
                for (int k = 0; k < S; ++k) fg += s_sem[k * 256 + j] * g[8 + k];
                const float dL_dalpha = T * fg - (Asuf + bgdot) * inv1ma;
                Asuf += w * fg;
                // alpha = o * G (gradient ignores the 0.99 clamp, as the public rasterizer does)
                const float dL_dpow = araw * dL_dalpha;
                const float gdx = -(q1.x * dx + q1.y * dy) * dL_dpow;
                const float gdy = -(q1.z * dy + q1.y * dx) * dL_dpow;
                v[0] = gdx; v[1] = gdy; v[2] = fabsf(gdx); v[3] = fabsf(gdy);
                v[4] = -0.5f * dx * dx * dL_dpow; v[5] = -dx * dy * dL_dpow; v[6] = -0.5f * dy * dy * dL_dpow;
                v[7] = G * dL_dalpha;
                v[8] = w * g[0]; v[9] = w * g[1]; v[10] = w * g[2];
                const float wd = w * g[3];
                v[13] = w * g[4]; v[14] = w * g[5]; v[15] = w * g[6];
                if (ISECT && isect) {
                    const float iden = 1.f / den;
                    const float k1 = wd * rz * iden;               // d dep / d plane
                    v[12] = k1;
                    const float k2 = -k1 * q1.w * iden;            // d dep / d den * wd
                    v[13] += k2 * rx; v[14] += k2 * ry; v[15] += k2 * rz;
                } else {
                    v[11] = wd;
                }
#pragma unroll
                for (int k = 0; k < S; ++k) vs[k] = w * g[8 + k];
            }
            float r4[4];
            wave_reduce16(v, r4);
            if ((lane & 15) < 4) {
                const int sub = lane & 15;
                const float val = sub == 0 ? r4[0] : (sub == 1 ? r4[1] : (sub == 2 ? r4[2] : r4[3]));
                const int k = 8 * (lane >> 5) + 4 * ((lane >> 4) & 1) + sub;
                atomicAdd(&s_acc[k][j], val);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float t = wave_sum(vs[k]);
                if (lane == 0) atomicAdd(&s_acc[VCR_GRAD_FLOATS + k][j], t);
            }
        }
        __syncthreads();
        if (threadIdx.x < n) {
            const uint32_t id = s_id[threadIdx.x];
            float* dst = reinterpret_cast<float*>(sgrad + id);
#pragma unroll
            for (int k = 0; k < VCR_GRAD_FLOATS; ++k) {
                const float val = s_acc[k][threadIdx.x];
                if (val != 0.f) atomicAdd(dst + k, val);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float val = s_acc[VCR_GRAD_FLOATS + k][threadIdx.x];
                if (val != 0.f) atomicAdd(sgrad_sem + (size_t)id * S + k, val);
            }
        }
    }
}

template <int S, bool ISECT>
int launch_fwd_fc(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o, int tiles,
                  hipStream_t st) {
#define VCR_FWD(FC)                                                                                               \
    hipLaunchKernelGGL((composite_fwd_kernel<S, ISECT, FC>), dim3(tiles), dim3(256), 0, st, a, g.rec, g.sem,       \
                       b.point_list, b.ranges, im.final_T, im.n_contrib, o.out, o.count, o.score)
    switch (a.f_count) {
        case 0: VCR_FWD(0); break;
        case 1: case 2: VCR_FWD(1); break;
        default: VCR_FWD(3); break;
    }
#undef VCR_FWD
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool ISECT>
int launch_fwd_s(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o, int tiles,
                 hipStream_t st) {
    switch (a.S) {
        case 0: return launch_fwd_fc<0, ISECT>(a, g, b, im, o, tiles, st);
        case 1: return launch_fwd_fc<1, ISECT>(a, g, b, im, o, tiles, st);
        case 2: return launch_fwd_fc<2, ISECT>(a, g, b, im, o, tiles, st);
        case 3: return launch_fwd_fc<3, ISECT>(a, g, b, im, o, tiles, st);
        default: return launch_fwd_fc<4, ISECT>(a, g, b, im, o, tiles, st);
    }
}

template <bool ISECT>
int launch_bwd_s(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, const float* dL_dout, GradRec* sgrad,
                 float* sgrad_sem, int tiles, hipStream_t st) {
#define VCR_BWD(SS)                                                                                              \
    hipLaunchKernelGGL((composite_bwd_kernel<SS, ISECT>), dim3(tiles), dim3(256), 0, st, a, g.rec, g.sem,         \
                       b.point_list, b.ranges, im.final_T, im.n_contrib, dL_dout, sgrad, sgrad_sem)
    switch (a.S) {
        case 0: VCR_BWD(0); break;
        case 1: VCR_BWD(1); break;
        case 2: VCR_BWD(2); break;
        case 3: VCR_BWD(3); break;
        default: VCR_BWD(4); break;
    }
#undef VCR_BWD
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int vcr_launch_composite_forward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o,
                                 hipStream_t st) {
    const int tiles = ((a.W + VCR_TILE - 1) / VCR_TILE) * ((a.H + VCR_TILE - 1) / VCR_TILE);
    const bool isect = a.dirs != nullptr && a.normals_precomp != nullptr;
    return isect ? launch_fwd_s<true>(a, g, b, im, o, tiles, st) : launch_fwd_s<false>(a, g, b, im, o, tiles, st);
}

int vcr_launch_composite_backward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, const float* dL_dout,
                                  GradRec* sgrad, float* sgrad_sem, hipStream_t st) {
    const int tiles = ((a.W + VCR_TILE - 1) / VCR_TILE) * ((a.H + VCR_TILE - 1) / VCR_TILE);
    const bool isect = a.dirs != nullptr && a.normals_precomp != nullptr;
    return isect ? launch_bwd_s<true>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st)
                 : launch_bwd_s<false>(a, g, b, im, dL_dout, sgrad, sgrad_sem, tiles, st);
}
