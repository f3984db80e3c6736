// Per-Gaussian parameter kernels around the rasterizer call (HBM-streaming, one lane per Gaussian):
//   fused activations + shortest-axis normal + camera orientation
//     (scene/gaussian_model.py:125-192, tools/general_utils.py:98-119, gaussian_renderer/__init__.py:95-101),
//   fused multi-tensor Adam (scene/gaussian_model.py:232-262: torch.optim.Adam(lr=0, eps=1e-15), per-group lr),
//   densification statistics (scene/gaussian_model.py:669-671, trainer.py:345).
#include "vcr_common.h"
#include "model_math.h"
#include <string.h>
#include <math.h>

namespace {

__global__ void __launch_bounds__(256) activate_fwd_kernel(int N, const float* __restrict__ scaling_raw,
                                                           const float* __restrict__ rotation_raw,
                                                           const float* __restrict__ opacity_raw,
                                                           const float* __restrict__ xyz, const float* __restrict__ campos,
                                                           const float* __restrict__ Rw2c, float* __restrict__ scales,
                                                           float* __restrict__ rots, float* __restrict__ opac,
                                                           float* __restrict__ normals, uint8_t* __restrict__ aux) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t i3 = 3 * (size_t)i;
    const float l[3] = {scaling_raw[i3], scaling_raw[i3 + 1], scaling_raw[i3 + 2]};
    float p[3] = {0.f, 0.f, 0.f};
    if (normals) { p[0] = xyz[i3]; p[1] = xyz[i3 + 1]; p[2] = xyz[i3 + 2]; }
    const ActOut a = activate_one(l, reinterpret_cast<const float4*>(rotation_raw)[i], opacity_raw[i], p, campos, Rw2c, normals != nullptr);
    scales[i3] = a.s[0]; scales[i3 + 1] = a.s[1]; scales[i3 + 2] = a.s[2];
    reinterpret_cast<float4*>(rots)[i] = a.q;
    opac[i] = a.o;
    if (!normals) return;
    normals[i3] = a.n[0]; normals[i3 + 1] = a.n[1]; normals[i3 + 2] = a.n[2];
    aux[i] = a.aux;
}

__global__ void __launch_bounds__(256) activate_bwd_kernel(int N, const float* __restrict__ scaling_raw,
                                                           const float* __restrict__ rotation_raw,
                                                           const float* __restrict__ opacity_raw,
                                                           const float* __restrict__ Rw2c, const uint8_t* __restrict__ aux,
                                                           const float* __restrict__ d_scales, const float* __restrict__ d_rots,
                                                           const float* __restrict__ d_opac, const float* __restrict__ d_normals,
                                                           const float* __restrict__ d_scaling_extra, float* __restrict__ d_scaling_raw, float* __restrict__ d_rotation_raw,
                                                           float* __restrict__ d_opacity_raw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t i3 = 3 * (size_t)i;
    const float l[3] = {scaling_raw[i3], scaling_raw[i3 + 1], scaling_raw[i3 + 2]};
    float ds[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f}, ex[3] = {0.f, 0.f, 0.f};
    if (d_scales) { ds[0] = d_scales[i3]; ds[1] = d_scales[i3 + 1]; ds[2] = d_scales[i3 + 2]; }
    if (d_normals) { dn[0] = d_normals[i3]; dn[1] = d_normals[i3 + 1]; dn[2] = d_normals[i3 + 2]; }
    if (d_scaling_extra) { ex[0] = d_scaling_extra[i3]; ex[1] = d_scaling_extra[i3 + 1]; ex[2] = d_scaling_extra[i3 + 2]; }
    float gs[3], go;
    float4 gq;
    activate_bwd_one(l, reinterpret_cast<const float4*>(rotation_raw)[i], opacity_raw[i], Rw2c, d_normals ? aux[i] : (uint8_t)0,
                     d_scales != nullptr, ds, d_rots != nullptr, d_rots ? reinterpret_cast<const float4*>(d_rots)[i] : make_float4(0.f, 0.f, 0.f, 0.f),
                     d_opac != nullptr, d_opac ? d_opac[i] : 0.f, d_normals != nullptr, dn, ex, gs, gq, go);
    d_scaling_raw[i3] = gs[0]; d_scaling_raw[i3 + 1] = gs[1]; d_scaling_raw[i3 + 2] = gs[2];
    d_opacity_raw[i] = go;
    reinterpret_cast<float4*>(d_rotation_raw)[i] = gq;
}

// ---- the static tail of a training iteration in ONE pass over the Gaussians (single GPU, no surgery this iteration) ------
// activation backward (+ the l1_scale gradient) -> densification statistics -> Adam on xyz / scaling / rotation / opacity ->
// activation for the NEXT iteration's camera.  Replaces activate_bwd + scale_reg_bwd + densify_stats + adam + activate_fwd:
// the raw-parameter gradients are never written, the parameters are read once.  Same per-Gaussian functions as the
// stand-alone kernels above.
__global__ void __launch_bounds__(256) geometry_step_kernel(VcrGeometryStep a, GeomBias gb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const TailGrads none = {};
    geometry_step_one<false>(a, gb, i, none);
}

// ---- fused multi-tensor Adam ---------------------------------------------------------------------
#define VCR_ADAM_MAX 8
struct AdamPack {
    float* p[VCR_ADAM_MAX]; const float* g[VCR_ADAM_MAX]; float* m[VCR_ADAM_MAX]; float* v[VCR_ADAM_MAX];
    long long start[VCR_ADAM_MAX + 1];       // prefix of float4-group counts
    float lr[VCR_ADAM_MAX];
    int tail[VCR_ADAM_MAX];                  // numel % 4
    int n;
};

// Each thread owns 4 consecutive elements (16-byte loads/stores); `start` counts float4 groups per tensor
// (tensor sizes are padded up to a multiple of 4 by the host wrapper through a scalar tail launch).
__global__ void __launch_bounds__(256) adam_kernel(AdamPack pk, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale) {
    const long long total = pk.start[pk.n];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int t = 0;
#pragma unroll
        for (int k = 1; k < VCR_ADAM_MAX; ++k) if (k < pk.n && i >= pk.start[k]) t = k;
        const long long j = i - pk.start[t];
        const float4 g4 = reinterpret_cast<const float4*>(pk.g[t])[j];
        float4 m4 = reinterpret_cast<float4*>(pk.m[t])[j];
        float4 v4 = reinterpret_cast<float4*>(pk.v[t])[j];
        float4 p4 = reinterpret_cast<float4*>(pk.p[t])[j];
        const float step = pk.lr[t] / bc1;
        const float gg[4] = {g4.x * gscale, g4.y * gscale, g4.z * gscale, g4.w * gscale};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) adam_one(pp[c], mm[c], vv[c], gg[c], b1, b2, eps, step, bc2_sqrt);
        reinterpret_cast<float4*>(pk.m[t])[j] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(pk.v[t])[j] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        reinterpret_cast<float4*>(pk.p[t])[j] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    }
}

// scalar tails (numel % 4 elements per tensor)
__global__ void adam_tail_kernel(AdamPack pk, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    const int t = blockIdx.x;
    const long long n4 = pk.start[t + 1] - pk.start[t];
    const int tail = (int)pk.tail[t];
    if ((int)threadIdx.x >= tail) return;
    const long long j = n4 * 4 + threadIdx.x;
    const float g = pk.g[t][j] * gscale;
    const float m = b1 * pk.m[t][j] + (1.f - b1) * g;
    const float v = b2 * pk.v[t][j] + (1.f - b2) * g * g;
    pk.m[t][j] = m; pk.v[t][j] = v;
    pk.p[t][j] -= (pk.lr[t] / bc1) * (m / (sqrtf(v) / bc2_sqrt + eps));
}

// xyz_gradient_accum[vis] += ||grad[:, :2]||, denom[vis] += 1, max_radii2D[vis] = max(., radii)
__global__ void __launch_bounds__(256) densify_stats_kernel(int N, const float* __restrict__ grad2d,
                                                            const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                            float* __restrict__ denom, float* __restrict__ max_radii) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = grad2d[3 * (size_t)i], gy = grad2d[3 * (size_t)i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
    max_radii[i] = fmaxf(max_radii[i], (float)r);
}

// Mean squared distance to the 3 nearest neighbours (simple-knn's distCUDA2, scene/gaussian_model.py:211), exact
// brute force: 256 queries per block, candidates streamed through LDS in 256-point tiles.  One-time initialisation.
__global__ void __launch_bounds__(256) knn3_kernel(int N, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float s_p[256][3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (i < N) { px = pts[3 * (size_t)i]; py = pts[3 * (size_t)i + 1]; pz = pts[3 * (size_t)i + 2]; }
    float b0 = 3.4e38f, b1 = 3.4e38f, b2 = 3.4e38f;
    for (int base = 0; base < N; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < N) { s_p[threadIdx.x][0] = pts[3 * (size_t)j]; s_p[threadIdx.x][1] = pts[3 * (size_t)j + 1]; s_p[threadIdx.x][2] = pts[3 * (size_t)j + 2]; }
        __syncthreads();
        const int n = min(256, N - base);
        for (int k = 0; k < n; ++k) {
            const float dx = s_p[k][0] - px, dy = s_p[k][1] - py, dz = s_p[k][2] - pz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (base + k == i) continue;
            if (d < b2) {
                if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
                else b2 = d;
            }
        }
    }
    if (i < N) {
        const int have = N - 1 < 3 ? N - 1 : 3;
        float s = 0.f;
        if (have > 0) s += b0;
        if (have > 1) s += b1;
        if (have > 2) s += b2;
        out[i] = have > 0 ? s / 3.f : 0.f;            // simple-knn divides by 3 regardless
    }
}

}  // namespace

// ---- exact 3-NN through a uniform grid (N > 2048): O(N) expected instead of the O(N^2) brute force above ------------------
namespace {
struct KnnGrid { float lo[3]; float h, inv_h, slack; int dim[3]; };

__device__ __forceinline__ unsigned int float_order(float f) {        // monotone map float -> uint (for atomicMin / atomicMax)
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float float_unorder(unsigned int u) {
    const unsigned int v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(v);
#else
    memcpy(&f, &v, 4);
#endif
    return f;
}

// lohi[0..2] = min, lohi[3..5] = max (order-mapped uints; initialised to 0xFFFFFFFF / 0)
__global__ void __launch_bounds__(256) knn_bbox_kernel(int N, const float* __restrict__ pts, unsigned int* __restrict__ lohi) {
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float v = pts[3 * (size_t)i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(lohi + k, float_order(lo[k])); atomicMax(lohi + 3 + k, float_order(hi[k])); }
    }
}

__device__ __forceinline__ int knn_cell(float v, float lo, float inv_h, int dim) {
    const int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ void __launch_bounds__(256) knn_key_kernel(int N, const float* __restrict__ pts, KnnGrid g, uint32_t* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int cx = knn_cell(pts[3 * (size_t)i], g.lo[0], g.inv_h, g.dim[0]), cy = knn_cell(pts[3 * (size_t)i + 1], g.lo[1], g.inv_h, g.dim[1]),
              cz = knn_cell(pts[3 * (size_t)i + 2], g.lo[2], g.inv_h, g.dim[2]);
    keys[i] = (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
}

// positions in cell order (float4: x, y, z, original index as bits) + [begin, end) of every cell
__global__ void __launch_bounds__(256) knn_gather_kernel(int N, const float* __restrict__ pts, const uint32_t* __restrict__ keys_sorted,
                                                         const uint32_t* __restrict__ ids_sorted, float4* __restrict__ spts,
                                                         uint2* __restrict__ ranges) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const uint32_t id = ids_sorted[j], k = keys_sorted[j];
    spts[j] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
    if (j == 0) ranges[k].x = 0;
    else {
        const uint32_t kp = keys_sorted[j - 1];
        if (kp != k) { ranges[kp].y = (uint32_t)j; ranges[k].x = (uint32_t)j; }
    }
    if (j == N - 1) ranges[k].y = (uint32_t)N;
}

// One lane per point (in cell order): grow a box of cells around the point's cell shell by shell; after every shell the
// search stops if the third-best squared distance is not larger than the squared distance to the nearest face of the box
// (faces on the boundary of the grid do not count: nothing lies beyond them).  Exact.
__global__ void __launch_bounds__(256) knn_search_kernel(int N, KnnGrid g, const float4* __restrict__ spts, const uint2* __restrict__ ranges,
                                                         float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float4 p = spts[j];
    const int cx = knn_cell(p.x, g.lo[0], g.inv_h, g.dim[0]), cy = knn_cell(p.y, g.lo[1], g.inv_h, g.dim[1]),
              cz = knn_cell(p.z, g.lo[2], g.inv_h, g.dim[2]);
    float b0 = 3.4e38f, b1 = 3.4e38f, b2 = 3.4e38f;
    auto visit = [&](int x, int y, int z) {
        const uint2 rg = ranges[(size_t)(z * g.dim[1] + y) * g.dim[0] + x];
        for (uint32_t k = rg.x; k < rg.y; ++k) {
            if ((int)k == j) continue;
            const float4 q = spts[k];
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < b2) {
                if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
                else b2 = d;
            }
        }
    };
    const int rmax = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.dim[0] - 1), y0 = max(cy - r, 0), y1 = min(cy + r, g.dim[1] - 1),
                  z0 = max(cz - r, 0), z1 = min(cz + r, g.dim[2] - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                if (abs(z - cz) == r || abs(y - cy) == r) {                  // a row on the shell: all of it
                    for (int x = x0; x <= x1; ++x) visit(x, y, z);
                } else {                                                     // an interior row: only its two ends are new
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (cx + r < g.dim[0]) visit(cx + r, y, z);
                }
            }
        // distance from the point to the nearest face of the box that is not a face of the whole grid (nothing lies
        // beyond those); g.slack absorbs the rounding of the face positions and of the cell assignment
        float face = 3.4e38f;
        if (cx - r > 0) face = fminf(face, p.x - (g.lo[0] + (float)(cx - r) * g.h));
        if (cx + r < g.dim[0] - 1) face = fminf(face, g.lo[0] + (float)(cx + r + 1) * g.h - p.x);
        if (cy - r > 0) face = fminf(face, p.y - (g.lo[1] + (float)(cy - r) * g.h));
        if (cy + r < g.dim[1] - 1) face = fminf(face, g.lo[1] + (float)(cy + r + 1) * g.h - p.y);
        if (cz - r > 0) face = fminf(face, p.z - (g.lo[2] + (float)(cz - r) * g.h));
        if (cz + r < g.dim[2] - 1) face = fminf(face, g.lo[2] + (float)(cz + r + 1) * g.h - p.z);
        if (face >= 3.0e38f) break;                                          // the box covers the whole grid
        face -= g.slack;
        if (face > 0.f && b2 <= face * face) break;
    }
    out[__float_as_uint(p.w)] = (b0 + b1 + b2) / 3.f;
}
}  // namespace

extern "C" int vcr_knn3_mean_dist2(int N, const float* points, float* out, void* stream) {
    if (N <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 2048) {                                         // small clouds (and N - 1 < 3 neighbours): brute force
        hipLaunchKernelGGL(knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, points, out);
        VCR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // one-time initialisation: host round trips and hipMalloc are fine here
    unsigned int* d_lohi = nullptr;
    VCR_HIP_CHECK(hipMalloc((void**)&d_lohi, 6 * sizeof(unsigned int)));
    unsigned int init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, got[6];
    hipError_t e = hipMemcpyAsync(d_lohi, init, sizeof(init), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { hipLaunchKernelGGL(knn_bbox_kernel, dim3(512), dim3(256), 0, st, N, points, d_lohi); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(got, d_lohi, sizeof(got), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_lohi);
    if (e != hipSuccess) { vcr_set_error("vcr_knn3_mean_dist2: bounding box failed: %s", hipGetErrorString(e)); return 1; }
    KnnGrid g;
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) { g.lo[k] = float_unorder(got[k]); ext = fmaxf(ext, float_unorder(got[3 + k]) - g.lo[k]); }
    if (!(ext > 0.f) || !(ext < 3.0e38f)) {                  // all points coincide (or non-finite input): every distance is 0 / brute force
        hipLaunchKernelGGL(knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, points, out);
        VCR_HIP_CHECK(hipGetLastError());
        return 0;
    }
    int gdim = (int)cbrt((double)N / 2.0);                   // ~2 points per cell if the cloud filled its box
    gdim = gdim < 1 ? 1 : (gdim > 256 ? 256 : gdim);
    g.h = ext / (float)gdim * 1.0001f;
    g.inv_h = 1.f / g.h;
    g.slack = 4e-7f * (fmaxf(fmaxf(fabsf(g.lo[0]), fabsf(g.lo[1])), fabsf(g.lo[2])) + ext) + 1e-4f * g.h;
    size_t ncell = 1;
    for (int k = 0; k < 3; ++k) {
        const int d = (int)floorf((float_unorder(got[3 + k]) - g.lo[k]) * g.inv_h) + 1;
        g.dim[k] = d < 1 ? 1 : (d > gdim ? gdim : d);
        ncell *= (size_t)g.dim[k];
    }
    int bits = 1;
    while (((size_t)1 << bits) < ncell) ++bits;
    const size_t nb = vcr_align(sizeof(uint32_t) * (size_t)N), pb = vcr_align(sizeof(uint2) * (size_t)N);
    const size_t total = 3 * nb + 2 * pb + vcr_align(sizeof(float4) * (size_t)N) + vcr_align(sizeof(uint2) * ncell) +
                         vcr_align(sizeof(uint32_t) * VCR_SORT_TOTALS_WORDS) + vcr_sort_scratch_bytes(N);
    char* buf = nullptr;
    VCR_HIP_CHECK(hipMalloc((void**)&buf, total));
    char* c = buf;
    uint32_t* keys = (uint32_t*)c; c += nb;
    uint32_t* keys_s = (uint32_t*)c; c += nb;
    uint32_t* ids_s = (uint32_t*)c; c += nb;
    uint2* pa = (uint2*)c; c += pb;
    uint2* pbuf = (uint2*)c; c += pb;
    float4* spts = (float4*)c; c += vcr_align(sizeof(float4) * (size_t)N);
    uint2* ranges = (uint2*)c; c += vcr_align(sizeof(uint2) * ncell);
    uint32_t* totals = (uint32_t*)c; c += vcr_align(sizeof(uint32_t) * VCR_SORT_TOTALS_WORDS);
    uint32_t* hist = (uint32_t*)c;
    const int blocks = (N + 255) / 256;
    int rc = 0;
    e = hipMemsetAsync(ranges, 0, sizeof(uint2) * ncell, st);
    if (e == hipSuccess) { hipLaunchKernelGGL(knn_key_kernel, dim3(blocks), dim3(256), 0, st, N, points, g, keys); e = hipGetLastError(); }
    if (e == hipSuccess) rc = vcr_sort_pairs(N, keys, nullptr, nullptr, pa, pbuf, keys_s, ids_s, 0, bits, hist, totals, st, nullptr);
    if (e == hipSuccess && !rc) {
        hipLaunchKernelGGL(knn_gather_kernel, dim3(blocks), dim3(256), 0, st, N, points, keys_s, ids_s, spts, ranges);
        hipLaunchKernelGGL(knn_search_kernel, dim3(blocks), dim3(256), 0, st, N, g, spts, ranges, out);
        e = hipGetLastError();
    }
    const hipError_t es = hipStreamSynchronize(st);          // the scratch is freed below
    (void)hipFree(buf);
    if (rc) return 1;
    if (e != hipSuccess || es != hipSuccess) { vcr_set_error("vcr_knn3_mean_dist2: %s", hipGetErrorString(e != hipSuccess ? e : es)); return 1; }
    return 0;
}

extern "C" int vcr_activate_forward(int N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                    const float* xyz, const float* campos, const float* R_w2c, float* scales, float* rots,
                                    float* opac, float* normals_cam, uint8_t* aux, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(activate_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_raw,
                       rotation_raw, opacity_raw, xyz, campos, R_w2c, scales, rots, opac, normals_cam, aux);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_activate_backward(int N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                     const float* R_w2c, const uint8_t* aux, const float* d_scales, const float* d_rots,
                                     const float* d_opac, const float* d_normals, const float* d_scaling_extra,
                                     float* d_scaling_raw, float* d_rotation_raw, float* d_opacity_raw, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_raw,
                       rotation_raw, opacity_raw, R_w2c, aux, d_scales, d_rots, d_opac, d_normals, d_scaling_extra,
                       d_scaling_raw, d_rotation_raw, d_opacity_raw);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, const float* lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, void* stream) {
    if (ntensors <= 0) return 0;
    if (ntensors > VCR_ADAM_MAX) { vcr_set_error("vcr_adam_step: at most %d tensors per call", VCR_ADAM_MAX); return 1; }
    AdamPack pk;
    bool any_tail = false;
    pk.n = ntensors;
    pk.start[0] = 0;
    for (int k = 0; k < ntensors; ++k) {
        pk.p[k] = params[k]; pk.g[k] = grads[k]; pk.m[k] = exp_avg[k]; pk.v[k] = exp_avg_sq[k];
        pk.lr[k] = lr[k];
        pk.start[k + 1] = pk.start[k] + numel[k] / 4;
        pk.tail[k] = (int)(numel[k] % 4);
        any_tail |= pk.tail[k] != 0;
        if ((((uintptr_t)params[k]) | ((uintptr_t)grads[k]) | ((uintptr_t)exp_avg[k]) | ((uintptr_t)exp_avg_sq[k])) & 15) {
            vcr_set_error("vcr_adam_step: tensor %d is not 16-byte aligned", k);
            return 1;
        }
    }
    for (int k = ntensors; k < VCR_ADAM_MAX; ++k) { pk.p[k] = nullptr; pk.g[k] = nullptr; pk.m[k] = nullptr; pk.v[k] = nullptr; pk.lr[k] = 0.f; pk.tail[k] = 0; pk.start[k + 1] = pk.start[ntensors]; }
    const long long total = pk.start[ntensors];
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    if (total > 0) {
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk, beta1, beta2, eps, (float)bc1,
                           (float)sqrt(bc2), grad_scale);
    }
    if (any_tail)
        hipLaunchKernelGGL(adam_tail_kernel, dim3(ntensors), dim3(64), 0, (hipStream_t)stream, pk, beta1, beta2, eps,
                           (float)bc1, (float)sqrt(bc2), grad_scale);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_geometry_step(const VcrGeometryStep* args, void* stream) {
    if (!args) { vcr_set_error("vcr_geometry_step: args is NULL"); return 1; }
    const VcrGeometryStep& a = *args;
    if (a.N <= 0) return 0;
    if (!a.xyz || !a.scaling || !a.rotation || !a.opacity || !a.m_scaling || !a.v_scaling || !a.m_rotation || !a.v_rotation ||
        !a.m_opacity || !a.v_opacity || (a.d_means3D && (!a.m_xyz || !a.v_xyz || a.step_xyz < 1)) || a.step_scaling < 1 ||
        a.step_rotation < 1 || a.step_opacity < 1 ||
        (a.d_normals && (!a.aux || !a.Rw2c)) || (a.scale_reg_sums && (!a.scale_reg_gout || !a.trans || !a.scale)) ||
        (a.grad2d && (!a.radii || !a.accum || !a.denom || !a.max_radii)) ||
        (a.next_scales && (!a.next_rots || !a.next_opac || (a.next_normals && (!a.next_campos || !a.next_Rw2c || !a.next_aux))))) {
        vcr_set_error("vcr_geometry_step: inconsistent arguments"); return 1;
    }
    if (!(a.grad_scale > 0.f)) { vcr_set_error("vcr_geometry_step: grad_scale must be > 0 (1 on one GPU, 1 / world after a sum all-reduce)"); return 1; }
    if ((((uintptr_t)a.rotation) | ((uintptr_t)a.m_rotation) | ((uintptr_t)a.v_rotation) | ((uintptr_t)a.d_rots) | ((uintptr_t)a.next_rots)) & 15) {
        vcr_set_error("vcr_geometry_step: quaternion arrays must be 16-byte aligned"); return 1;
    }
    GeomBias gb;
    vcr_geometry_bias(a, gb);
    hipLaunchKernelGGL(geometry_step_kernel, dim3((a.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, gb);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_densify_stats(int N, const float* grad2d, const int32_t* radii, float* accum, float* denom,
                                 float* max_radii, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, grad2d, radii, accum,
                       denom, max_radii);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------- densify / prune row surgery (scene/gaussian_model.py:425-531) -------------------------------------------
// The reference re-packs every parameter tensor and both Adam moments with boolean-mask indexing + torch.cat, ~40 kernels
// and as many allocations per operation.  Here ONE launch moves the selected rows of ALL arrays of the model at once:
//   mode 0 (compact): out = in[mask]                                        (prune_points, `:456-475`)
//   mode 1 (append) : out = cat(in, in[mask] x copies)  (zero_new: zeros)   (densification_postfix, `:477-531`)
// vcr_rows_plan counts the selected rows per 256-row block and scans the counts (offsets[nblk] = total, read by the host to
// size the outputs -- the same synchronisation the reference's `mask.sum()` / boolean indexing implies).
namespace {
constexpr int ROWS_PER_BLOCK = 256;

__global__ void __launch_bounds__(256) rows_count_kernel(int N, const uint8_t* __restrict__ mask, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s[4];
    const int i = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    const bool k = i < N && mask[i];
    const unsigned long long b = __builtin_amdgcn_ballot_w64(k);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of nblk counts by ONE workgroup (nblk <= a few 10^4): offsets[0..nblk), offsets[nblk] = total
__global__ void __launch_bounds__(1024) rows_scan_kernel(int nblk, uint32_t* __restrict__ counts_offsets) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + t;
        const uint32_t v = i < nblk ? counts_offsets[i] : 0u;
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t pre = carry;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (i < nblk) counts_offsets[i] = pre + inc - v;
        __syncthreads();
        if (t == 1023) carry = pre + inc;
        __syncthreads();
    }
    if (t == 0) counts_offsets[nblk] = carry;
}

__global__ void __launch_bounds__(256) rows_move_kernel(int N, const uint8_t* __restrict__ mask,
                                                        const uint32_t* __restrict__ offsets, int nblk, VcrRowArrays arr,
                                                        int mode, int copies) {
    __shared__ uint32_t s_rank[ROWS_PER_BLOCK];          // destination rank of the row inside the selection, or ~0u
    __shared__ uint32_t s_w[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row0 = blockIdx.x * ROWS_PER_BLOCK;
    const int i = row0 + t;
    const bool k = i < N && mask[i];
    const unsigned long long b = __builtin_amdgcn_ballot_w64(k);
    if (lane == 0) s_w[w] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t pre = offsets[blockIdx.x];
    for (int q = 0; q < w; ++q) pre += s_w[q];
    s_rank[t] = k ? pre + (uint32_t)__popcll(b & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t M = offsets[nblk];
    const int rows = min(ROWS_PER_BLOCK, N - row0);
    for (int a = 0; a < arr.n; ++a) {
        const VcrRowArray A = arr.a[a];
        const int wd = A.width;
        const float* __restrict__ src = A.in + (size_t)row0 * wd;
        for (int e = t; e < rows * wd; e += 256) {
            const int r = e / wd, j = e - r * wd;
            const uint32_t rk = s_rank[r];
            const float v = src[e];
            if (mode == 1) A.out[(size_t)row0 * wd + e] = v;                       // the first N rows are kept as they are
            if (rk != 0xFFFFFFFFu) {
                if (mode == 0) A.out[(size_t)rk * wd + j] = v;
                else
                    for (int c = 0; c < copies; ++c)
                        A.out[((size_t)N + (size_t)c * M + rk) * wd + j] = A.zero_new ? 0.f : v;
            }
        }
    }
}
}  // namespace

extern "C" size_t vcr_rows_plan_bytes(int N) { return sizeof(uint32_t) * ((size_t)(N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK + 1); }

extern "C" int vcr_rows_plan(int N, const uint8_t* mask, uint32_t* offsets, void* stream) {
    if (N < 0 || (N > 0 && (!mask || !offsets))) { vcr_set_error("vcr_rows_plan: bad arguments"); return 1; }
    const int nblk = (N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipStream_t st = (hipStream_t)stream;
    if (nblk > 0) hipLaunchKernelGGL(rows_count_kernel, dim3(nblk), dim3(256), 0, st, N, mask, offsets);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, st, nblk, offsets);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_rows_move(int N, const uint8_t* mask, const uint32_t* offsets, const VcrRowArrays* arrays, int mode,
                             int copies, void* stream) {
    if (!arrays || arrays->n < 0 || arrays->n > VCR_MAX_ROW_ARRAYS || mode < 0 || mode > 1 || copies < 1) {
        vcr_set_error("vcr_rows_move: bad arguments"); return 1;
    }
    if (N <= 0 || arrays->n == 0) return 0;
    const int nblk = (N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(rows_move_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, N, mask, offsets, nblk, *arrays, mode, copies);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
