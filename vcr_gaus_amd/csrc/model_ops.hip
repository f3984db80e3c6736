// Per-Gaussian parameter kernels around the rasterizer call (HBM-streaming, one lane per Gaussian):
//   fused activations + shortest-axis normal + camera orientation
//     (scene/gaussian_model.py:125-192, tools/general_utils.py:98-119, gaussian_renderer/__init__.py:95-101),
//   fused multi-tensor Adam (scene/gaussian_model.py:232-262: torch.optim.Adam(lr=0, eps=1e-15), per-group lr),
//   densification statistics (scene/gaussian_model.py:669-671, trainer.py:345).
#include "vcr_common.h"

namespace {

__device__ __forceinline__ void quat_R(float r, float x, float y, float z, float R[9]) {
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// aux: bits0-1 = shortest axis, bit2 = flipped
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
    const float l0 = scaling_raw[i3], l1 = scaling_raw[i3 + 1], l2 = scaling_raw[i3 + 2];
    const float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
    scales[i3] = s0; scales[i3 + 1] = s1; scales[i3 + 2] = s2;
    const float4 qr = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float inv = 1.f / fmaxf(sqrtf(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w), 1e-12f);
    const float4 q = make_float4(qr.x * inv, qr.y * inv, qr.z * inv, qr.w * inv);
    reinterpret_cast<float4*>(rots)[i] = q;
    opac[i] = 1.f / (1.f + expf(-opacity_raw[i]));
    if (!normals) return;
    int axis = 0;                              // torch.argmin: first minimum
    float sm = s0;
    if (s1 < sm) { sm = s1; axis = 1; }
    if (s2 < sm) { sm = s2; axis = 2; }
    float R[9];
    quat_R(q.x, q.y, q.z, q.w, R);
    float n0 = R[axis], n1 = R[3 + axis], n2 = R[6 + axis];
    const float vx = xyz[i3] - campos[0], vy = xyz[i3 + 1] - campos[1], vz = xyz[i3 + 2] - campos[2];
    const bool keep = (vx * n0 + vy * n1 + vz * n2) > 0.f;
    if (!keep) { n0 = -n0; n1 = -n1; n2 = -n2; }
    normals[i3] = Rw2c[0] * n0 + Rw2c[1] * n1 + Rw2c[2] * n2;
    normals[i3 + 1] = Rw2c[3] * n0 + Rw2c[4] * n1 + Rw2c[5] * n2;
    normals[i3 + 2] = Rw2c[6] * n0 + Rw2c[7] * n1 + Rw2c[8] * n2;
    aux[i] = (uint8_t)(axis | (keep ? 0 : 4));
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
#pragma unroll
    for (int k = 0; k < 3; ++k)
        d_scaling_raw[i3 + k] = (d_scales ? d_scales[i3 + k] * expf(scaling_raw[i3 + k]) : 0.f) + (d_scaling_extra ? d_scaling_extra[i3 + k] : 0.f);
    const float o = 1.f / (1.f + expf(-opacity_raw[i]));
    d_opacity_raw[i] = d_opac ? d_opac[i] * o * (1.f - o) : 0.f;
    const float4 qr = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float inv = 1.f / fmaxf(sqrtf(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w), 1e-12f);
    const float r = qr.x * inv, x = qr.y * inv, y = qr.z * inv, z = qr.w * inv;
    float g[4] = {0.f, 0.f, 0.f, 0.f};                 // gradient w.r.t. the unit quaternion
    if (d_rots) { const float4 d = reinterpret_cast<const float4*>(d_rots)[i]; g[0] = d.x; g[1] = d.y; g[2] = d.z; g[3] = d.w; }
    if (d_normals) {
        const int axis = aux[i] & 3;
        const float sgn = (aux[i] & 4) ? -1.f : 1.f;
        const float c0 = d_normals[i3], c1 = d_normals[i3 + 1], c2 = d_normals[i3 + 2];
        // n_cam = Rw2c * (sgn * R[:,axis])
        const float w0 = sgn * (Rw2c[0] * c0 + Rw2c[3] * c1 + Rw2c[6] * c2);
        const float w1 = sgn * (Rw2c[1] * c0 + Rw2c[4] * c1 + Rw2c[7] * c2);
        const float w2 = sgn * (Rw2c[2] * c0 + Rw2c[5] * c1 + Rw2c[8] * c2);
        float dR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        dR[axis] = w0; dR[3 + axis] = w1; dR[6 + axis] = w2;
        float h[4];
        h[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        h[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
        h[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
        h[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        // build_rotation re-normalises its (already unit) input: project onto the tangent space
        const float hd = h[0] * r + h[1] * x + h[2] * y + h[3] * z;
        g[0] += h[0] - r * hd; g[1] += h[1] - x * hd; g[2] += h[2] - y * hd; g[3] += h[3] - z * hd;
    }
    // q = raw/|raw|
    const float gd = g[0] * r + g[1] * x + g[2] * y + g[3] * z;
    reinterpret_cast<float4*>(d_rotation_raw)[i] =
        make_float4((g[0] - r * gd) * inv, (g[1] - x * gd) * inv, (g[2] - y * gd) * inv, (g[3] - z * gd) * inv);
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
        for (int c = 0; c < 4; ++c) {
            mm[c] = b1 * mm[c] + (1.f - b1) * gg[c];
            vv[c] = b2 * vv[c] + (1.f - b2) * gg[c] * gg[c];
            pp[c] -= step * (mm[c] / (sqrtf(vv[c]) / bc2_sqrt + eps));
        }
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

extern "C" int vcr_knn3_mean_dist2(int N, const float* points, float* out, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, points, out);
    VCR_HIP_CHECK(hipGetLastError());
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
