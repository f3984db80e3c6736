// Image-space kernels of the D-Normal training step (all HBM-streaming, one pass each):
//   depth -> normal (tools/normal_utils.py:24-41, tools/graphics_utils.py:111-131) and its adjoint,
//   rendered-normal normalisation (gaussian_renderer/__init__.py:133-134),
//   confidence-weighted normal loss (tools/loss_utils.py:122-143, trainer.py:261-293),
//   fused L1 + SSIM (tools/loss_utils.py:36,49-92) with saved partials for a single-pass backward.
#include "vcr_common.h"

namespace {

// ---------------- block reduction helper ------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float x) {
    x = dpp_add<0x128>(x); x = dpp_add<0x124>(x); x = dpp_add<0x122>(x); x = dpp_add<0x121>(x);
    x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
    return x;
}
// Block-level sum, then ONE fp64 atomic per block into one of VCR_NSLOT slots (spread over L2 channels so the
// atomics do not serialise on a single address); finalize_sums_kernel folds the slots in a fixed order.
#define VCR_NSLOT 256
template <int K>
__device__ __forceinline__ void block_accumulate(double* __restrict__ slots, const float (&v)[K]) {
    __shared__ float s_part[K][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) s_part[k][wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const float t = s_part[threadIdx.x][0] + s_part[threadIdx.x][1] + s_part[threadIdx.x][2] + s_part[threadIdx.x][3];
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (t != 0.f) atomicAdd(slots + (size_t)(b % VCR_NSLOT) * K + threadIdx.x, (double)t);
    }
}

// sums layout: [K results][VCR_NSLOT x K slots]
// mode 0: result[k] = sums[k] * scale (means);  mode 1: result[0] = count > 0 ? (sums[0]+sums[1])/sums[2] : 0
__global__ void finalize_sums_kernel(int K, double* __restrict__ sums, int mode, double scale, float* __restrict__ result) {
    __shared__ double s_t[8];
    // 64 lanes: lane l folds slots l, l+64, l+128, l+192 (fixed order), then a fixed butterfly
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
        for (int s = threadIdx.x; s < VCR_NSLOT; s += 64) t += sums[K + (size_t)s * K + k];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (threadIdx.x == 0) {
            sums[k] = t;
            s_t[k] = t;
            if (mode == 0 && result) result[k] = (float)(t * scale);
        }
    }
    __syncthreads();
    if (mode == 1 && threadIdx.x == 0 && result) result[0] = s_t[2] > 0.0 ? (float)((s_t[0] + s_t[1]) / s_t[2]) : 0.f;
}

// ---------------- depth -> normal ----------------------------------------------------------------
struct Intr { float fx, fy, cx, cy; };

__device__ __forceinline__ void backproject(const float* __restrict__ depth, int W, int x, int y, Intr k, float X[3]) {
    const float z = depth[y * W + x];
    X[0] = ((float)x + 0.5f - k.cx) * z / k.fx;
    X[1] = ((float)y + 0.5f - k.cy) * z / k.fy;
    X[2] = z;
}

// torch.gradient, spacing 1: central inside, one-sided at the borders.
__device__ __forceinline__ void grad_cols(const float* depth, int W, int H, int x, int y, Intr k, float d[3]) {
    float a[3], b[3];
    const int x0 = x > 0 ? x - 1 : x, x1 = x < W - 1 ? x + 1 : x;
    backproject(depth, W, x0, y, k, a);
    backproject(depth, W, x1, y, k, b);
    const float sc = (x0 == x - 1 && x1 == x + 1) ? 0.5f : 1.f;
    d[0] = (b[0] - a[0]) * sc; d[1] = (b[1] - a[1]) * sc; d[2] = (b[2] - a[2]) * sc;
}
__device__ __forceinline__ void grad_rows(const float* depth, int W, int H, int x, int y, Intr k, float d[3]) {
    float a[3], b[3];
    const int y0 = y > 0 ? y - 1 : y, y1 = y < H - 1 ? y + 1 : y;
    backproject(depth, W, x, y0, k, a);
    backproject(depth, W, x, y1, k, b);
    const float sc = (y0 == y - 1 && y1 == y + 1) ? 0.5f : 1.f;
    d[0] = (b[0] - a[0]) * sc; d[1] = (b[1] - a[1]) * sc; d[2] = (b[2] - a[2]) * sc;
}

__global__ void __launch_bounds__(256) depth_normal_fwd_kernel(int H, int W, Intr k, const float* __restrict__ depth,
                                                               float* __restrict__ normal) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float dx[3], dy[3];
    grad_cols(depth, W, H, x, y, k, dx);
    grad_rows(depth, W, H, x, y, k, dy);
    const float c0 = dx[1] * dy[2] - dx[2] * dy[1], c1 = dx[2] * dy[0] - dx[0] * dy[2], c2 = dx[0] * dy[1] - dx[1] * dy[0];
    const float inv = 1.f / fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);
    float* o = normal + 3 * ((size_t)y * W + x);
    o[0] = c0 * inv; o[1] = c1 * inv; o[2] = c2 * inv;
}

// pass 1: dL/dn -> dL/d(dx), dL/d(dy) per pixel ([P,6] scratch)
__global__ void __launch_bounds__(256) depth_normal_bwd1_kernel(int H, int W, Intr k, const float* __restrict__ depth,
                                                                const float* __restrict__ dnormal,
                                                                float* __restrict__ dgrad) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float dx[3], dy[3];
    grad_cols(depth, W, H, x, y, k, dx);
    grad_rows(depth, W, H, x, y, k, dy);
    const float c[3] = {dx[1] * dy[2] - dx[2] * dy[1], dx[2] * dy[0] - dx[0] * dy[2], dx[0] * dy[1] - dx[1] * dy[0]};
    const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float* g = dnormal + 3 * ((size_t)y * W + x);
    float dc[3];
    if (len > 1e-12f) {
        const float inv = 1.f / len;
        const float n[3] = {c[0] * inv, c[1] * inv, c[2] * inv};
        const float dot = n[0] * g[0] + n[1] * g[1] + n[2] * g[2];
        dc[0] = (g[0] - n[0] * dot) * inv; dc[1] = (g[1] - n[1] * dot) * inv; dc[2] = (g[2] - n[2] * dot) * inv;
    } else {
        dc[0] = g[0] * 1e12f; dc[1] = g[1] * 1e12f; dc[2] = g[2] * 1e12f;
    }
    // c = dx x dy  ->  d(dx) = dy x dc,  d(dy) = dc x dx
    float* o = dgrad + 6 * ((size_t)y * W + x);
    o[0] = dy[1] * dc[2] - dy[2] * dc[1]; o[1] = dy[2] * dc[0] - dy[0] * dc[2]; o[2] = dy[0] * dc[1] - dy[1] * dc[0];
    o[3] = dc[1] * dx[2] - dc[2] * dx[1]; o[4] = dc[2] * dx[0] - dc[0] * dx[2]; o[5] = dc[0] * dx[1] - dc[1] * dx[0];
}

// coefficient of f[i] inside torch.gradient evaluated at index q (q in {i-1, i, i+1}), length n
__device__ __forceinline__ float grad_coef(int q, int i, int n) {
    if (q < 0 || q >= n) return 0.f;
    if (q == 0) return i == 1 ? 1.f : (i == 0 ? -1.f : 0.f);
    if (q == n - 1) return i == n - 1 ? 1.f : (i == n - 2 ? -1.f : 0.f);
    return i == q + 1 ? 0.5f : (i == q - 1 ? -0.5f : 0.f);
}

// pass 2: gather the stencil adjoint and fold through X = z * k(u,v)
__global__ void __launch_bounds__(256) depth_normal_bwd2_kernel(int H, int W, Intr k, const float* __restrict__ dgrad,
                                                                float* __restrict__ ddepth) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float dX[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int o = -1; o <= 1; ++o) {
        const float cc = grad_coef(x + o, x, W);
        if (cc != 0.f) {
            const float* g = dgrad + 6 * ((size_t)y * W + (x + o));
            dX[0] += cc * g[0]; dX[1] += cc * g[1]; dX[2] += cc * g[2];
        }
        const float cr = grad_coef(y + o, y, H);
        if (cr != 0.f) {
            const float* g = dgrad + 6 * ((size_t)(y + o) * W + x) + 3;
            dX[0] += cr * g[0]; dX[1] += cr * g[1]; dX[2] += cr * g[2];
        }
    }
    ddepth[(size_t)y * W + x] = dX[0] * ((float)x + 0.5f - k.cx) / k.fx + dX[1] * ((float)y + 0.5f - k.cy) / k.fy + dX[2];
}

// ---------------- planar [3,H,W] -> normalised [H,W,3] ---------------------------------------------
__global__ void __launch_bounds__(256) normalize_chw_fwd_kernel(int P, const float* __restrict__ in,
                                                                float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float a = in[i], b = in[P + i], c = in[2 * (size_t)P + i];
    const float inv = 1.f / fmaxf(sqrtf(a * a + b * b + c * c), 1e-12f);
    out[3 * (size_t)i] = a * inv; out[3 * (size_t)i + 1] = b * inv; out[3 * (size_t)i + 2] = c * inv;
}
__global__ void __launch_bounds__(256) normalize_chw_bwd_kernel(int P, const float* __restrict__ in,
                                                                const float* __restrict__ dout, float* __restrict__ din) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float a = in[i], b = in[P + i], c = in[2 * (size_t)P + i];
    const float len = sqrtf(a * a + b * b + c * c);
    const float g0 = dout[3 * (size_t)i], g1 = dout[3 * (size_t)i + 1], g2 = dout[3 * (size_t)i + 2];
    float d0, d1, d2;
    if (len > 1e-12f) {
        const float inv = 1.f / len;
        const float n0 = a * inv, n1 = b * inv, n2 = c * inv, dot = n0 * g0 + n1 * g1 + n2 * g2;
        d0 = (g0 - n0 * dot) * inv; d1 = (g1 - n1 * dot) * inv; d2 = (g2 - n2 * dot) * inv;
    } else { d0 = g0 * 1e12f; d1 = g1 * 1e12f; d2 = g2 * 1e12f; }
    din[i] = d0; din[P + i] = d1; din[2 * (size_t)P + i] = d2;
}

// ---------------- weighted / masked normal loss ----------------------------------------------------
// sums[0] = sum w*|p-g|_1, sums[1] = sum w*(1 - p.g), sums[2] = #selected pixels
__global__ void __launch_bounds__(256) normal_loss_fwd_kernel(int P, const float* __restrict__ pred,
                                                              const float* __restrict__ gt, const float* __restrict__ wsrc,
                                                              float exp_t, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ depth, float depth_max,
                                                              double* __restrict__ sums) {
    float s0 = 0.f, s1 = 0.f, cnt = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        if (mask && !mask[i]) continue;
        if (depth && !(depth[i] < depth_max)) continue;
        const float p0 = pred[3 * (size_t)i], p1 = pred[3 * (size_t)i + 1], p2 = pred[3 * (size_t)i + 2];
        const float g0 = gt[3 * (size_t)i], g1 = gt[3 * (size_t)i + 1], g2 = gt[3 * (size_t)i + 2];
        float w = 1.f;
        if (wsrc && exp_t > 0.f) {
            const float c = wsrc[3 * (size_t)i] * g0 + wsrc[3 * (size_t)i + 1] * g1 + wsrc[3 * (size_t)i + 2] * g2;
            w = __expf((c - 1.f) / exp_t);
        }
        s0 += w * (fabsf(p0 - g0) + fabsf(p1 - g1) + fabsf(p2 - g2));
        s1 += w * (1.f - (p0 * g0 + p1 * g1 + p2 * g2));
        cnt += 1.f;
    }
    const float v[3] = {s0, s1, cnt};
    block_accumulate<3>(sums + 3, v);
}

// dpred = scale * w * (sign(p-g) - g); optionally dgt = scale * w * (-sign(p-g) - p) (consistency loss, both sides live)
__global__ void __launch_bounds__(256) normal_loss_bwd_kernel(int P, const float* __restrict__ pred,
                                                              const float* __restrict__ gt, const float* __restrict__ wsrc,
                                                              float exp_t, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ depth, float depth_max,
                                                              const double* __restrict__ sums, const float* __restrict__ gout,
                                                              float* __restrict__ dpred, float* __restrict__ dgt, int acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float d[3] = {0.f, 0.f, 0.f}, e[3] = {0.f, 0.f, 0.f};
    if ((!mask || mask[i]) && (!depth || depth[i] < depth_max)) {
        const float scale = gout[0] / (float)sums[2];
        const float p[3] = {pred[3 * (size_t)i], pred[3 * (size_t)i + 1], pred[3 * (size_t)i + 2]};
        const float g[3] = {gt[3 * (size_t)i], gt[3 * (size_t)i + 1], gt[3 * (size_t)i + 2]};
        float w = 1.f;
        if (wsrc && exp_t > 0.f) {
            const float c = wsrc[3 * (size_t)i] * g[0] + wsrc[3 * (size_t)i + 1] * g[1] + wsrc[3 * (size_t)i + 2] * g[2];
            w = __expf((c - 1.f) / exp_t);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float df = p[k] - g[k];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            d[k] = scale * w * (sg - g[k]);
            e[k] = scale * w * (-sg - p[k]);
        }
    }
    if (acc & 1) { d[0] += dpred[3 * (size_t)i]; d[1] += dpred[3 * (size_t)i + 1]; d[2] += dpred[3 * (size_t)i + 2]; }
    dpred[3 * (size_t)i] = d[0]; dpred[3 * (size_t)i + 1] = d[1]; dpred[3 * (size_t)i + 2] = d[2];
    if (dgt) {
        if (acc & 2) { e[0] += dgt[3 * (size_t)i]; e[1] += dgt[3 * (size_t)i + 1]; e[2] += dgt[3 * (size_t)i + 2]; }
        dgt[3 * (size_t)i] = e[0]; dgt[3 * (size_t)i + 1] = e[1]; dgt[3 * (size_t)i + 2] = e[2];
    }
}

// ---------------- all three normal losses in one pass each way ---------------------------------------
// mono_normal L3 = normal_loss(n, gt); depth_normal L4 = normal_loss(est, gt, weight exp((n.gt - 1)/exp_t), mask, depth
// threshold); consistent_normal L5 = normal_loss(est, n), with n = normalize(rendered normal), est = depth -> normal
// (trainer.py:261-293).  One forward kernel replaces normalize + depth_to_normal + up to three loss kernels; one backward
// kernel replaces the loss backwards, the normalisation backward and stage 1 of the depth-to-normal adjoint.  n and est
// are recomputed in the backward instead of being stored (the depth stencil is L2-resident).
// sums9: [9 totals][VCR_NSLOT x 9]: triple (sum w|p-g|_1, sum w(1-p.g), count) per loss.
struct PixNormals { float n[3], nlen, raw[3], e[3], elen, dx[3], dy[3], c[3]; };

__device__ __forceinline__ void pixel_normals(int H, int W, Intr k, const float* __restrict__ depth,
                                              const float* __restrict__ nrm, int x, int y, PixNormals& q) {
    const size_t P = (size_t)H * W, i = (size_t)y * W + x;
    q.raw[0] = nrm[i]; q.raw[1] = nrm[P + i]; q.raw[2] = nrm[2 * P + i];
    q.nlen = sqrtf(q.raw[0] * q.raw[0] + q.raw[1] * q.raw[1] + q.raw[2] * q.raw[2]);
    const float ni = 1.f / fmaxf(q.nlen, 1e-12f);
    q.n[0] = q.raw[0] * ni; q.n[1] = q.raw[1] * ni; q.n[2] = q.raw[2] * ni;
    grad_cols(depth, W, H, x, y, k, q.dx);
    grad_rows(depth, W, H, x, y, k, q.dy);
    q.c[0] = q.dx[1] * q.dy[2] - q.dx[2] * q.dy[1]; q.c[1] = q.dx[2] * q.dy[0] - q.dx[0] * q.dy[2];
    q.c[2] = q.dx[0] * q.dy[1] - q.dx[1] * q.dy[0];
    q.elen = sqrtf(q.c[0] * q.c[0] + q.c[1] * q.c[1] + q.c[2] * q.c[2]);
    const float ei = 1.f / fmaxf(q.elen, 1e-12f);
    q.e[0] = q.c[0] * ei; q.e[1] = q.c[1] * ei; q.e[2] = q.c[2] * ei;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) normal_losses_fwd_kernel(int H, int W, Intr k, const float* __restrict__ depth,
                                                                const float* __restrict__ nrm, const float* __restrict__ gt,
                                                                const uint8_t* __restrict__ mask, float depth_max, float exp_t,
                                                                int active, double* __restrict__ sums9) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (x < W && y < H) {
        PixNormals q;
        pixel_normals(H, W, k, depth, nrm, x, y, q);
        const size_t i = (size_t)y * W + x;
        float g[3] = {0.f, 0.f, 0.f};
        if (active & 3) { g[0] = gt[3 * i]; g[1] = gt[3 * i + 1]; g[2] = gt[3 * i + 2]; }
        if (active & 1) {
            v[0] = fabsf(q.n[0] - g[0]) + fabsf(q.n[1] - g[1]) + fabsf(q.n[2] - g[2]);
            v[1] = 1.f - (q.n[0] * g[0] + q.n[1] * g[1] + q.n[2] * g[2]);
            v[2] = 1.f;
        }
        if ((active & 2) && (!mask || mask[i]) && !(depth_max > 0.f && !(depth[i] < depth_max))) {
            float w = 1.f;
            if (exp_t > 0.f) w = __expf((q.n[0] * g[0] + q.n[1] * g[1] + q.n[2] * g[2] - 1.f) / exp_t);
            v[3] = w * (fabsf(q.e[0] - g[0]) + fabsf(q.e[1] - g[1]) + fabsf(q.e[2] - g[2]));
            v[4] = w * (1.f - (q.e[0] * g[0] + q.e[1] * g[1] + q.e[2] * g[2]));
            v[5] = 1.f;
        }
        if (active & 4) {
            v[6] = fabsf(q.e[0] - q.n[0]) + fabsf(q.e[1] - q.n[1]) + fabsf(q.e[2] - q.n[2]);
            v[7] = 1.f - (q.e[0] * q.n[0] + q.e[1] * q.n[1] + q.e[2] * q.n[2]);
            v[8] = 1.f;
        }
    }
    block_accumulate<9>(sums9 + 9, v);
}

// totals + the three losses (0 when a loss has no selected pixel / is inactive).  9 waves, one per sum, lane l folds slots
// l, l + 64, ... in the fixed order of finalize_losses_kernel (bit-identical results); the one-wave form took 22 us.
__global__ void __launch_bounds__(576) finalize_normal_losses_kernel(double* __restrict__ sums9, float* __restrict__ res3) {
    __shared__ double s_t[9];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    double t = 0.0;
    for (int s = lane; s < VCR_NSLOT; s += 64) t += sums9[9 + (size_t)s * 9 + k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) { s_t[k] = t; sums9[k] = t; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int l = threadIdx.x;
        res3[l] = s_t[3 * l + 2] > 0.0 ? (float)((s_t[3 * l] + s_t[3 * l + 1]) / s_t[3 * l + 2]) : 0.f;
    }
}

// seeds3: dL/dL3, dL/dL4, dL/dL5 (device).  Writes d(out) for the three normal planes and the [P,6] stage-1 scratch of the
// depth-to-normal adjoint (consumed by depth_normal_bwd2_kernel).
__global__ void __launch_bounds__(256) normal_losses_bwd_kernel(int H, int W, Intr k, const float* __restrict__ depth,
                                                                const float* __restrict__ nrm, const float* __restrict__ gt,
                                                                const uint8_t* __restrict__ mask, float depth_max, float exp_t,
                                                                int active, const double* __restrict__ sums9,
                                                                const float* __restrict__ seeds3, float* __restrict__ dnrm,
                                                                float* __restrict__ dgrad) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    PixNormals q;
    pixel_normals(H, W, k, depth, nrm, x, y, q);
    const size_t P = (size_t)H * W, i = (size_t)y * W + x;
    float g[3] = {0.f, 0.f, 0.f};
    if (active & 3) { g[0] = gt[3 * i]; g[1] = gt[3 * i + 1]; g[2] = gt[3 * i + 2]; }
    float dn[3] = {0.f, 0.f, 0.f}, de[3] = {0.f, 0.f, 0.f};
    if ((active & 1) && sums9[2] > 0.0) {
        const float sc = seeds3[0] / (float)sums9[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] += sc * (sgn(q.n[j] - g[j]) - g[j]);
    }
    if ((active & 2) && sums9[5] > 0.0 && (!mask || mask[i]) && !(depth_max > 0.f && !(depth[i] < depth_max))) {
        float w = 1.f;
        if (exp_t > 0.f) w = __expf((q.n[0] * g[0] + q.n[1] * g[1] + q.n[2] * g[2] - 1.f) / exp_t);
        const float sc = seeds3[1] / (float)sums9[5] * w;
#pragma unroll
        for (int j = 0; j < 3; ++j) de[j] += sc * (sgn(q.e[j] - g[j]) - g[j]);
    }
    if ((active & 4) && sums9[8] > 0.0) {
        const float sc = seeds3[2] / (float)sums9[8];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float sg = sgn(q.e[j] - q.n[j]);
            de[j] += sc * (sg - q.n[j]);
            dn[j] += sc * (-sg - q.e[j]);
        }
    }
    // through n = raw / |raw|
    float d0, d1, d2;
    if (q.nlen > 1e-12f) {
        const float inv = 1.f / q.nlen, dot = q.n[0] * dn[0] + q.n[1] * dn[1] + q.n[2] * dn[2];
        d0 = (dn[0] - q.n[0] * dot) * inv; d1 = (dn[1] - q.n[1] * dot) * inv; d2 = (dn[2] - q.n[2] * dot) * inv;
    } else { d0 = dn[0] * 1e12f; d1 = dn[1] * 1e12f; d2 = dn[2] * 1e12f; }
    dnrm[i] = d0; dnrm[P + i] = d1; dnrm[2 * P + i] = d2;
    dnrm[3 * P + i] = 0.f;                    // the alpha plane behind the normal planes carries no loss
    // through est = c / |c|, c = dx x dy   (stage 1 of the depth-to-normal adjoint)
    float dc[3];
    if (q.elen > 1e-12f) {
        const float inv = 1.f / q.elen, dot = q.e[0] * de[0] + q.e[1] * de[1] + q.e[2] * de[2];
        dc[0] = (de[0] - q.e[0] * dot) * inv; dc[1] = (de[1] - q.e[1] * dot) * inv; dc[2] = (de[2] - q.e[2] * dot) * inv;
    } else { dc[0] = de[0] * 1e12f; dc[1] = de[1] * 1e12f; dc[2] = de[2] * 1e12f; }
    float* o = dgrad + 6 * i;
    o[0] = q.dy[1] * dc[2] - q.dy[2] * dc[1]; o[1] = q.dy[2] * dc[0] - q.dy[0] * dc[2]; o[2] = q.dy[0] * dc[1] - q.dy[1] * dc[0];
    o[3] = dc[1] * q.dx[2] - dc[2] * q.dx[1]; o[4] = dc[2] * q.dx[0] - dc[0] * q.dx[2]; o[5] = dc[0] * q.dx[1] - dc[1] * q.dx[0];
}

// ---------------- fused L1 + SSIM ------------------------------------------------------------------
#define SSIM_R 5
#define SSIM_TX 32
// Rows per workgroup (round 5, profiles/r5_ssim_tiles.txt + kernel traces): 32 rows make the halo 1.72x instead of 2.13x on the
// loads and 1.31x instead of 1.63x on the horizontal pass -- the BACKWARD gains (64 -> 58 us at 1080p), the FORWARD, whose five
// row-sum planes then take 42 KB of LDS (3 instead of 6 workgroups per CU), loses (64 -> 75 us): 16 rows forward, 32 backward.
#define SSIM_TY_FWD 16
#define SSIM_TY_BWD 32
#define SSIM_HW (SSIM_TX + 2 * SSIM_R)        // 42: tile + halo, x
#define SSIM_HS (SSIM_HW + 2)                 // 44: halo row stride
struct GaussWin { float w[11]; };

// One workgroup = 32 x TY output pixels of one channel.  The (32+10) x (TY+10) halo of both images is staged in LDS and the 11x11
// window is evaluated separably with REGISTER sliding windows: every lane produces 4 adjacent outputs in the horizontal pass (14
// loaded values feed 4 x 11 taps) and TY / 8 in the vertical pass, ~3x fewer LDS reads per output pixel than the one-output-per-
// lane form (LDS-bound: 84 us at 1080p).  The five window sums travel in natural pairs -- (a, b), (a^2, b^2) and a b alone --
// on packed fp32 (v_pk_fma_f32: 3 instructions per tap instead of 5; the kernel is VALU-bound).  Zero padding as
// F.conv2d(padding=5).
// sums[0] += sum |a-b| ; sums[1] += sum ssim_map.  If `part` != null stores d ssim / d{mu1, sigma1^2, sigma12}.
typedef float lf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lf2 lpk_fma(lf2 a, lf2 b, lf2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int TY>
__global__ void __launch_bounds__(256) l1_ssim_fwd_kernel(int H, int W, GaussWin gw, const float* __restrict__ img1,
                                                          const float* __restrict__ img2, double* __restrict__ sums,
                                                          float* __restrict__ part) {
    constexpr int HH = TY + 2 * SSIM_R, VO = TY / 8;      // halo rows; output rows per lane in the vertical pass
    __shared__ lf2 s_ab[HH][SSIM_HS];                 // (a, b)
    __shared__ lf2 s_hm[HH][SSIM_TX + 1];             // row sums of (a, b)
    __shared__ lf2 s_hq[HH][SSIM_TX + 1];             // row sums of (a^2, b^2)
    __shared__ float s_hx[HH][SSIM_TX + 1];           // row sums of a b
    const int c = blockIdx.z, x0 = blockIdx.x * SSIM_TX, y0 = blockIdx.y * TY;
    const size_t P = (size_t)H * W;
    const float* A = img1 + c * P;
    const float* B = img2 + c * P;
    for (int t = threadIdx.x; t < SSIM_HW * HH; t += 256) {
        const int ly = t / SSIM_HW, lx = t % SSIM_HW, gx = x0 + lx - SSIM_R, gy = y0 + ly - SSIM_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        s_ab[ly][lx] = in ? lf2{A[(size_t)gy * W + gx], B[(size_t)gy * W + gx]} : lf2{0.f, 0.f};
    }
    __syncthreads();
    // horizontal pass: item = (halo row, group of 4 output columns)
    for (int it = threadIdx.x; it < HH * (SSIM_TX / 4); it += 256) {
        const int ly = it / (SSIM_TX / 4), lx0 = (it % (SSIM_TX / 4)) * 4;
        lf2 m[4], q[4];
        float x[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) { m[o] = lf2{0.f, 0.f}; q[o] = lf2{0.f, 0.f}; x[o] = 0.f; }
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            const lf2 ab = s_ab[ly][lx0 + k];
            const lf2 sq = ab * ab;
            const float cr = ab.x * ab.y;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (k - o >= 0 && k - o < 11) {
                    const float w = gw.w[k - o];
                    m[o] = lpk_fma(lf2{w, w}, ab, m[o]); q[o] = lpk_fma(lf2{w, w}, sq, q[o]); x[o] += w * cr;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) { s_hm[ly][lx0 + o] = m[o]; s_hq[ly][lx0 + o] = q[o]; s_hx[ly][lx0 + o] = x[o]; }
    }
    __syncthreads();
    // vertical pass: lane = (column, group of VO output rows)
    const int lx = threadIdx.x & 31, ly0 = (threadIdx.x >> 5) * VO;
    lf2 rm[VO], rq[VO];
    float rx[VO];
    {
        lf2 cm[10 + VO], cq[10 + VO];
        float cx[10 + VO];
#pragma unroll
        for (int k = 0; k < 10 + VO; ++k) { cm[k] = s_hm[ly0 + k][lx]; cq[k] = s_hq[ly0 + k][lx]; cx[k] = s_hx[ly0 + k][lx]; }
#pragma unroll
        for (int o = 0; o < VO; ++o) {
            lf2 tm = {0.f, 0.f}, tq = {0.f, 0.f};
            float tx = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = gw.w[k];
                tm = lpk_fma(lf2{w, w}, cm[o + k], tm); tq = lpk_fma(lf2{w, w}, cq[o + k], tq); tx += w * cx[o + k];
            }
            rm[o] = tm; rq[o] = tq; rx[o] = tx;
        }
    }
    float l1 = 0.f, sv_sum = 0.f;
    const int gx = x0 + lx;
#pragma unroll
    for (int o = 0; o < VO; ++o) {
        const int gy = y0 + ly0 + o;
        if (gx < W && gy < H) {
            const float m1 = rm[o].x, m2 = rm[o].y, q11 = rq[o].x, q22 = rq[o].y, q12 = rx[o];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            const float s11 = q11 - m11, s22 = q22 - m22, s12 = q12 - m12;
            const float An = 2.f * m12 + C1, Bn = 2.f * s12 + C2, Cd = m11 + m22 + C1, Dd = s11 + s22 + C2;
            const float iCD = 1.f / (Cd * Dd);
            const float sv = An * Bn * iCD;
            sv_sum += sv;
            const lf2 ctr = s_ab[ly0 + o + SSIM_R][lx + SSIM_R];
            l1 += fabsf(ctr.x - ctr.y);
            if (part) {
                // ssim = A B / (C D) with sigma terms expanded through mu1: total derivative w.r.t. mu1 at fixed
                // raw moments q11,q12:  s11 = q11 - mu1^2, s12 = q12 - mu1 mu2
                const float dS_dm1 = (2.f * m2 * Bn + An * (-2.f * m2)) * iCD - sv * (2.f * m1 * Dd + Cd * (-2.f * m1)) / (Cd * Dd);
                const float dS_dq11 = -sv / Dd;               // through sigma1^2 in D
                const float dS_dq12 = 2.f * An * iCD;         // through sigma12 in B
                const size_t oo = c * P + (size_t)gy * W + gx;
                part[oo] = dS_dm1; part[3 * P + oo] = dS_dq11; part[6 * P + oo] = dS_dq12;
            }
        }
    }
    const float v[2] = {l1, sv_sum};
    block_accumulate<2>(sums + 2, v);
}

// dimg1(p) = gl1 * sign(a-b) + gss * sum_q w(q-p) [ dm1(q) + 2 a(p) dq11(q) + b(p) dq12(q) ]      (same tiling; the three
// planes as one packed pair + one scalar)
template <int TY>
__global__ void __launch_bounds__(256) l1_ssim_bwd_kernel(int H, int W, GaussWin gw, const float* __restrict__ img1,
                                                          const float* __restrict__ img2, const float* __restrict__ part,
                                                          const float* __restrict__ g_l1, const float* __restrict__ g_ssim,
                                                          float wl1, float wss, float* __restrict__ dimg1) {
    constexpr int HH = TY + 2 * SSIM_R, VO = TY / 8;      // halo rows; output rows per lane in the vertical pass
    __shared__ lf2 s_p01[HH][SSIM_HS];
    __shared__ float s_p2[HH][SSIM_HS];
    __shared__ lf2 s_h01[HH][SSIM_TX + 1];
    __shared__ float s_h2[HH][SSIM_TX + 1];
    const int c = blockIdx.z, x0 = blockIdx.x * SSIM_TX, y0 = blockIdx.y * TY;
    const size_t P = (size_t)H * W;
    for (int t = threadIdx.x; t < SSIM_HW * HH; t += 256) {
        const int ly = t / SSIM_HW, lx = t % SSIM_HW, gx = x0 + lx - SSIM_R, gy = y0 + ly - SSIM_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t o = c * P + (size_t)gy * W + gx;
        s_p01[ly][lx] = in ? lf2{part[o], part[3 * P + o]} : lf2{0.f, 0.f};
        s_p2[ly][lx] = in ? part[6 * P + o] : 0.f;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < HH * (SSIM_TX / 4); it += 256) {
        const int ly = it / (SSIM_TX / 4), lx0 = (it % (SSIM_TX / 4)) * 4;
        lf2 p01[14];
        float p2[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) { p01[k] = s_p01[ly][lx0 + k]; p2[k] = s_p2[ly][lx0 + k]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            lf2 t01 = {0.f, 0.f};
            float t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) { const float w = gw.w[k]; t01 = lpk_fma(lf2{w, w}, p01[o + k], t01); t2 += w * p2[o + k]; }
            s_h01[ly][lx0 + o] = t01; s_h2[ly][lx0 + o] = t2;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly0 = (threadIdx.x >> 5) * VO;
    lf2 r01[VO];
    float r2[VO];
    {
        lf2 c01[10 + VO];
        float c2[10 + VO];
#pragma unroll
        for (int k = 0; k < 10 + VO; ++k) { c01[k] = s_h01[ly0 + k][lx]; c2[k] = s_h2[ly0 + k][lx]; }
#pragma unroll
        for (int o = 0; o < VO; ++o) {
            lf2 t01 = {0.f, 0.f};
            float t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) { const float w = gw.w[k]; t01 = lpk_fma(lf2{w, w}, c01[o + k], t01); t2 += w * c2[o + k]; }
            r01[o] = t01; r2[o] = t2;
        }
    }
    const int gx = x0 + lx;
    const float n = 1.f / (3.f * (float)P);
    const float kl1 = wl1 * g_l1[0] * n, kss = wss * g_ssim[0] * n;
#pragma unroll
    for (int o = 0; o < VO; ++o) {
        const int gy = y0 + ly0 + o;
        if (gx < W && gy < H) {
            const size_t oo = c * P + (size_t)gy * W + gx;
            const float a = img1[oo], b = img2[oo];
            const float df = a - b;
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            dimg1[oo] = kl1 * sg + kss * (r01[o].x + 2.f * a * r01[o].y + b * r2[o]);
        }
    }
}

}  // namespace

static Intr make_intr(const float* k4) { return Intr{k4[0], k4[1], k4[2], k4[3]}; }

extern "C" int vcr_depth_to_normal_forward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                           float* normal, void* stream) {
    const float k4[4] = {fx, fy, cx, cy};
    hipLaunchKernelGGL(depth_normal_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, H, W,
                       make_intr(k4), depth, normal);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_depth_to_normal_backward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                            const float* dnormal, float* scratch6, float* ddepth, void* stream) {
    const float k4[4] = {fx, fy, cx, cy};
    const dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(depth_normal_bwd1_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_intr(k4), depth, dnormal,
                       scratch6);
    hipLaunchKernelGGL(depth_normal_bwd2_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_intr(k4), scratch6,
                       ddepth);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_normalize_chw_forward(int P, const float* in_chw, float* out_hwc, void* stream) {
    hipLaunchKernelGGL(normalize_chw_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, in_chw, out_hwc);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_normalize_chw_backward(int P, const float* in_chw, const float* dout_hwc, float* din_chw, void* stream) {
    hipLaunchKernelGGL(normalize_chw_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, in_chw, dout_hwc,
                       din_chw);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_normal_loss_forward(int P, const float* pred, const float* gt, const float* wsrc, float exp_t,
                                       const uint8_t* mask, const float* depth, float depth_max, double* sums3,
                                       float* loss, int sums_prezeroed, void* stream) {
    if (!(sums_prezeroed & 1)) VCR_HIP_CHECK(hipMemsetAsync(sums3, 0, 3 * (1 + VCR_NSLOT) * sizeof(double), (hipStream_t)stream));
    const int blocks = min((P + 255) / 256, 2048);
    hipLaunchKernelGGL(normal_loss_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, pred, gt, wsrc, exp_t, mask,
                       depth, depth_max, sums3);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, 3, sums3, 1, 1.0, loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_normal_loss_backward(int P, const float* pred, const float* gt, const float* wsrc, float exp_t,
                                        const uint8_t* mask, const float* depth, float depth_max, const double* sums3,
                                        const float* gout, float* dpred, float* dgt, int accumulate, void* stream) {
    hipLaunchKernelGGL(normal_loss_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, pred, gt, wsrc,
                       exp_t, mask, depth, depth_max, sums3, gout, dpred, dgt, accumulate);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

static GaussWin make_window() {
    GaussWin g;
    double s = 0.0, v[11];
    for (int i = 0; i < 11; ++i) { v[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
    for (int i = 0; i < 11; ++i) g.w[i] = (float)(v[i] / s);
    return g;
}

extern "C" int vcr_normal_losses_forward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                         const float* normal_planes, const float* gt, const uint8_t* mask, float depth_max,
                                         float exp_t, int active, double* sums9, float* res3, int sums_prezeroed, void* stream) {
    if ((active & 3) && !gt) { vcr_set_error("vcr_normal_losses_forward: gt is NULL"); return 1; }
    if (!(sums_prezeroed & 1)) VCR_HIP_CHECK(hipMemsetAsync(sums9, 0, 9 * (1 + VCR_NSLOT) * sizeof(double), (hipStream_t)stream));
    const float k4[4] = {fx, fy, cx, cy};
    hipLaunchKernelGGL(normal_losses_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, H, W,
                       make_intr(k4), depth, normal_planes, gt, mask, depth_max, exp_t, active, sums9);
    if (!(sums_prezeroed & 2))
        hipLaunchKernelGGL(finalize_normal_losses_kernel, dim3(1), dim3(576), 0, (hipStream_t)stream, sums9, res3);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_normal_losses_backward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                          const float* normal_planes, const float* gt, const uint8_t* mask, float depth_max,
                                          float exp_t, int active, const double* sums9, const float* seeds3, float* scratch6,
                                          float* d_depth, float* d_normal_planes, void* stream) {
    if ((active & 3) && !gt) { vcr_set_error("vcr_normal_losses_backward: gt is NULL"); return 1; }
    const float k4[4] = {fx, fy, cx, cy};
    const dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(normal_losses_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_intr(k4), depth,
                       normal_planes, gt, mask, depth_max, exp_t, active, sums9, seeds3, d_normal_planes, scratch6);
    hipLaunchKernelGGL(depth_normal_bwd2_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_intr(k4), scratch6, d_depth);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_l1_ssim_forward(int H, int W, const float* img1, const float* img2, double* sums2, float* means2,
                                   float* partials9, int sums_prezeroed, void* stream) {
    if (!(sums_prezeroed & 1)) VCR_HIP_CHECK(hipMemsetAsync(sums2, 0, 2 * (1 + VCR_NSLOT) * sizeof(double), (hipStream_t)stream));
    const dim3 grid((W + SSIM_TX - 1) / SSIM_TX, (H + SSIM_TY_FWD - 1) / SSIM_TY_FWD, 3);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel<SSIM_TY_FWD>, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_window(), img1, img2, sums2,
                       partials9);
    if (!(sums_prezeroed & 2))
        hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, 2, sums2, 0,
                           1.0 / (3.0 * (double)H * (double)W), means2);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_l1_ssim_backward(int H, int W, const float* img1, const float* img2, const float* partials9,
                                    const float* g_l1, const float* g_ssim, float* dimg1, void* stream) {
    const dim3 grid((W + SSIM_TX - 1) / SSIM_TX, (H + SSIM_TY_BWD - 1) / SSIM_TY_BWD, 3);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel<SSIM_TY_BWD>, grid, dim3(256), 0, (hipStream_t)stream, H, W, make_window(), img1, img2, partials9,
                       g_l1, g_ssim, 1.f, 1.f, dimg1);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------- l1_scale regulariser (trainer.py:243-245, tools/math_utils.py:50-74) ----------------------------
// mean over Gaussians inside the normalised bounding box of min_axis(exp(_scaling)).
namespace {
__global__ void __launch_bounds__(256) scale_reg_fwd_kernel(int N, const float* __restrict__ scaling_raw,
                                                            const float* __restrict__ xyz, const float* __restrict__ trans,
                                                            const float* __restrict__ scale, double* __restrict__ sums) {
    float s0 = 0.f, cnt = 0.f;
    const float t0 = trans[0], t1 = trans[1], t2 = trans[2], i0 = 1.f / scale[0], i1 = 1.f / scale[1], i2 = 1.f / scale[2];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const size_t i3 = 3 * (size_t)i;
        const bool in = fabsf((xyz[i3] - t0) * i0) < 1.f && fabsf((xyz[i3 + 1] - t1) * i1) < 1.f &&
                        fabsf((xyz[i3 + 2] - t2) * i2) < 1.f;
        if (!in) continue;
        s0 += __expf(fminf(scaling_raw[i3], fminf(scaling_raw[i3 + 1], scaling_raw[i3 + 2])));
        cnt += 1.f;
    }
    const float v[3] = {s0, 0.f, cnt};
    block_accumulate<3>(sums + 3, v);
}
__global__ void __launch_bounds__(256) scale_reg_bwd_kernel(int N, const float* __restrict__ scaling_raw,
                                                            const float* __restrict__ xyz, const float* __restrict__ trans,
                                                            const float* __restrict__ scale, const double* __restrict__ sums,
                                                            const float* __restrict__ gout, float* __restrict__ dscaling) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t i3 = 3 * (size_t)i;
    float d[3] = {0.f, 0.f, 0.f};
    const bool in = fabsf((xyz[i3] - trans[0]) / scale[0]) < 1.f && fabsf((xyz[i3 + 1] - trans[1]) / scale[1]) < 1.f &&
                    fabsf((xyz[i3 + 2] - trans[2]) / scale[2]) < 1.f;
    if (in) {
        const float a = scaling_raw[i3], b = scaling_raw[i3 + 1], c = scaling_raw[i3 + 2];
        const int k = (a <= b && a <= c) ? 0 : (b <= c ? 1 : 2);           // first minimum, like torch.min
        d[k] = gout[0] / (float)sums[2] * __expf(fminf(a, fminf(b, c)));
    }
    dscaling[i3] = d[0]; dscaling[i3 + 1] = d[1]; dscaling[i3 + 2] = d[2];
}
}  // namespace

extern "C" int vcr_scale_reg_forward(int N, const float* scaling_raw, const float* xyz, const float* trans, const float* scale,
                                     double* sums3, float* loss, int sums_prezeroed, void* stream) {
    if (!(sums_prezeroed & 1)) VCR_HIP_CHECK(hipMemsetAsync(sums3, 0, 3 * (1 + VCR_NSLOT) * sizeof(double), (hipStream_t)stream));
    if (N > 0)
        hipLaunchKernelGGL(scale_reg_fwd_kernel, dim3(min((N + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, N,
                           scaling_raw, xyz, trans, scale, sums3);
    if (!(sums_prezeroed & 2)) hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, 3, sums3, 1, 1.0, loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_scale_reg_backward(int N, const float* scaling_raw, const float* xyz, const float* trans, const float* scale,
                                      const double* sums3, const float* gout, float* dscaling, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(scale_reg_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_raw, xyz,
                       trans, scale, sums3, gout, dscaling);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void weighted_total_kernel(int K, const float* __restrict__ res, const float* __restrict__ w, int sub_index,
                                      float* __restrict__ total) {
    float t = sub_index >= 0 ? -w[sub_index] : 0.f;
    for (int k = 0; k < K; ++k) t += res[k] * w[k];
    *total = t;
}

// (The folded slots are zeroed again, so a caller may keep ONE reduction buffer alive across steps instead of zero-filling
// a fresh one per call.)
// One launch for what the loss node needs at the end of its forward: fold the slots of the L1+SSIM sums (2), the
// scale-regulariser sums (3, optional) and the normal-loss sums (9, optional) in the fixed order of the individual
// finalize kernels, write the six loss values and their weighted total.
__global__ void __launch_bounds__(256) finalize_losses_kernel(double* __restrict__ sums2, double inv_count,
                                                              double* __restrict__ sums3, double* __restrict__ sums9,
                                                              float* __restrict__ res6, const float* __restrict__ w,
                                                              int sub_index, float* __restrict__ total) {
    __shared__ double s_t[14];
    // 14 sums (2 + 3 + 9), each folded by ONE wave in the fixed order of the individual finalize kernels (lane l takes
    // slots l, l+64, l+128, l+192, then a fixed butterfly); the four waves of the block take them round-robin
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int e = wv; e < 14; e += 4) {
        double* base = e < 2 ? sums2 : (e < 5 ? sums3 : sums9);
        const int K = e < 2 ? 2 : (e < 5 ? 3 : 9), k = e < 2 ? e : (e < 5 ? e - 2 : e - 5);
        double t = 0.0;
        if (base) {
            for (int s = lane; s < VCR_NSLOT; s += 64) { t += base[K + (size_t)s * K + k]; base[K + (size_t)s * K + k] = 0.0; }
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) base[k] = t;
        }
        if (lane == 0) s_t[e] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r[6];
        r[0] = (float)(s_t[0] * inv_count); r[1] = (float)(s_t[1] * inv_count);
        r[2] = s_t[4] > 0.0 ? (float)((s_t[2] + s_t[3]) / s_t[4]) : 0.f;
        for (int l = 0; l < 3; ++l) r[3 + l] = s_t[5 + 3 * l + 2] > 0.0 ? (float)((s_t[5 + 3 * l] + s_t[5 + 3 * l + 1]) / s_t[5 + 3 * l + 2]) : 0.f;
        float t = sub_index >= 0 ? -w[sub_index] : 0.f;
        for (int k = 0; k < 6; ++k) { res6[k] = r[k]; t += r[k] * w[k]; }
        *total = t;
    }
}

extern "C" int vcr_finalize_losses(int H, int W, double* sums2, double* sums3, double* sums9, float* res6, const float* w,
                                   int sub_index, float* total, void* stream) {
    if (!sums2 || !res6 || !w || !total) { vcr_set_error("vcr_finalize_losses: bad arguments"); return 1; }
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums2, 1.0 / (3.0 * (double)H * (double)W),
                       sums3, sums9, res6, w, sub_index, total);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_weighted_total(int K, const float* res, const float* w, int sub_index, float* total, void* stream) {
    if (K <= 0 || !res || !w || !total) { vcr_set_error("vcr_weighted_total: bad arguments"); return 1; }
    hipLaunchKernelGGL(weighted_total_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, K, res, w, sub_index, total);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_sums_elems(int k) { return k * (1 + VCR_NSLOT); }

// ---------------- small regularisers of `Trainer._compute_loss` (trainer.py:247-249,282-300) -----------------------------
// edge-aware weighted mean of the distortion / depth-variance map (tools/normal_utils.py:57-66), the normal-curvature
// loss (tools/loss_utils.py:287-300 + l1_loss(curv, 0)) and the opacity entropy (tools/loss_utils.py:30-33), one
// streaming kernel each way.  sums layout as above ([K results][VCR_NSLOT x K slots]).
namespace {

__device__ __forceinline__ float edge_weight(const float* __restrict__ img, int H, int W, int x, int y) {
    if (x == 0 || y == 0 || x == W - 1 || y == H - 1) return 0.f;                 // zero padding of the interior map
    const size_t P = (size_t)H * W;
    float gl = 0.f, gr = 0.f, gt = 0.f, gb = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = img + c * P + (size_t)y * W + x;
        const float v = p[0];
        gl += fabsf(v - p[-1]); gr += fabsf(v - p[1]); gt += fabsf(v - p[-W]); gb += fabsf(v - p[W]);
    }
    return __expf(-fmaxf(fmaxf(gl, gr), fmaxf(gt, gb)) * (1.f / 3.f));
}

__global__ void __launch_bounds__(256) edge_aware_fwd_kernel(int H, int W, const float* __restrict__ img,
                                                             const float* __restrict__ map, double* __restrict__ sums) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    float v[1] = {0.f};
    if (x < W && y < H) v[0] = map[(size_t)y * W + x] * edge_weight(img, H, W, x, y);
    block_accumulate<1>(sums + 1, v);
}

__global__ void __launch_bounds__(256) edge_aware_bwd_kernel(int H, int W, const float* __restrict__ img,
                                                             const float* __restrict__ gout, float* __restrict__ dmap) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < W && y < H) dmap[(size_t)y * W + x] = gout[0] / ((float)H * (float)W) * edge_weight(img, H, W, x, y);
}

// curvature vector of pixel (x, y): sum over the 4 replicate-padded neighbours nb of (n[nb] - n[c] m[c]) m[nb]
__device__ __forceinline__ void curv_vec(const float* __restrict__ n, const uint8_t* __restrict__ m, int H, int W, int x, int y,
                                         float cv[3], float& msum) {
    const int xs[4] = {x, x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
    const int ys[4] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1, y};
    const size_t c = (size_t)y * W + x;
    const float mc = m[c] ? 1.f : 0.f;
    const float c0 = n[3 * c] * mc, c1 = n[3 * c + 1] * mc, c2 = n[3 * c + 2] * mc;
    cv[0] = cv[1] = cv[2] = 0.f; msum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t q = (size_t)ys[k] * W + xs[k];
        if (m[q]) { cv[0] += n[3 * q] - c0; cv[1] += n[3 * q + 1] - c1; cv[2] += n[3 * q + 2] - c2; msum += 1.f; }
    }
    cv[0] *= mc; cv[1] *= mc; cv[2] *= mc;
}

__global__ void __launch_bounds__(256) curv_fwd_kernel(int H, int W, const float* __restrict__ n, const uint8_t* __restrict__ m,
                                                       double* __restrict__ sums) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    float v[1] = {0.f};
    if (x < W && y < H) {
        float cv[3], ms;
        curv_vec(n, m, H, W, x, y, cv, ms);
        v[0] = fabsf(cv[0]) + fabsf(cv[1]) + fabsf(cv[2]);
    }
    block_accumulate<1>(sums + 1, v);
}


// gather form of the adjoint (deterministic): pixel p receives +S[c] m[p] from every pixel c that has p as a
// (replicate-padded) neighbour, and -S[p] m[p] * (number of masked neighbours of p) from itself; S = sign(curv) m g / (H W).
__global__ void __launch_bounds__(256) curv_bwd_kernel(int H, int W, const float* __restrict__ n, const uint8_t* __restrict__ m,
                                                       const float* __restrict__ gout, float* __restrict__ dn) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    float g[3] = {0.f, 0.f, 0.f};
    if (m[p]) {
        float cv[3], ms;
        curv_vec(n, m, H, W, x, y, cv, ms);
        const float s0 = sgn(cv[0]), s1 = sgn(cv[1]), s2 = sgn(cv[2]);
        g[0] = -s0 * ms; g[1] = -s1 * ms; g[2] = -s2 * ms;
        // pixels c whose up / left / bottom / right neighbour is p (p itself at the matching border)
        const int cx[4] = {x, x + 1, x, x - 1}, cy[4] = {y + 1, y, y - 1, y};
        const bool self[4] = {y == 0, x == 0, y == H - 1, x == W - 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (cx[k] >= 0 && cx[k] < W && cy[k] >= 0 && cy[k] < H) {
                float c2[3], m2;
                curv_vec(n, m, H, W, cx[k], cy[k], c2, m2);
                g[0] += sgn(c2[0]); g[1] += sgn(c2[1]); g[2] += sgn(c2[2]);
            }
            if (self[k]) { g[0] += s0; g[1] += s1; g[2] += s2; }
        }
    }
    const float sc = gout[0] / ((float)H * (float)W);
    dn[3 * p] = g[0] * sc; dn[3 * p + 1] = g[1] * sc; dn[3 * p + 2] = g[2] * sc;
}

__device__ __forceinline__ bool inside_box(const float* __restrict__ xyz, const float* __restrict__ trans,
                                           const float* __restrict__ scale, size_t i3) {
    return fabsf((xyz[i3] - trans[0]) / scale[0]) < 1.f && fabsf((xyz[i3 + 1] - trans[1]) / scale[1]) < 1.f &&
           fabsf((xyz[i3 + 2] - trans[2]) / scale[2]) < 1.f;
}

__global__ void __launch_bounds__(256) entropy_fwd_kernel(int N, const float* __restrict__ opacity_raw,
                                                          const float* __restrict__ xyz, const float* __restrict__ trans,
                                                          const float* __restrict__ scale, double* __restrict__ sums) {
    float s = 0.f, cnt = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        if (xyz && !inside_box(xyz, trans, scale, 3 * (size_t)i)) continue;
        const float p = 1.f / (1.f + __expf(-opacity_raw[i]));
        s += -p * __logf(p + 1e-6f) - (1.f - p) * __logf(1.f - p + 1e-6f);
        cnt += 1.f;
    }
    const float v[3] = {s, 0.f, cnt};
    block_accumulate<3>(sums + 3, v);
}

__global__ void __launch_bounds__(256) entropy_bwd_kernel(int N, const float* __restrict__ opacity_raw,
                                                          const float* __restrict__ xyz, const float* __restrict__ trans,
                                                          const float* __restrict__ scale, const double* __restrict__ sums,
                                                          const float* __restrict__ gout, float* __restrict__ dopacity_raw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float d = 0.f;
    if (!xyz || inside_box(xyz, trans, scale, 3 * (size_t)i)) {
        const float p = 1.f / (1.f + __expf(-opacity_raw[i])), q = 1.f - p;
        const float dH = -__logf(p + 1e-6f) - p / (p + 1e-6f) + __logf(q + 1e-6f) + q / (q + 1e-6f);
        d = gout[0] / (float)sums[2] * dH * p * q;
    }
    dopacity_raw[i] = d;
}

}  // namespace

extern "C" int vcr_edge_aware_forward(int H, int W, const float* gt_image, const float* map, double* sums1, float* loss,
                                      void* stream) {
    if (H <= 0 || W <= 0 || !gt_image || !map || !sums1 || !loss) { vcr_set_error("vcr_edge_aware_forward: bad arguments"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    VCR_HIP_CHECK(hipMemsetAsync(sums1, 0, (1 + VCR_NSLOT) * sizeof(double), st));
    hipLaunchKernelGGL(edge_aware_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, H, W, gt_image, map, sums1);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, st, 1, sums1, 0, 1.0 / ((double)H * (double)W), loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_edge_aware_backward(int H, int W, const float* gt_image, const float* gout, float* dmap, void* stream) {
    if (H <= 0 || W <= 0 || !gt_image || !gout || !dmap) { vcr_set_error("vcr_edge_aware_backward: bad arguments"); return 1; }
    hipLaunchKernelGGL(edge_aware_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, H, W, gt_image,
                       gout, dmap);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_curv_forward(int H, int W, const float* normal_hwc, const uint8_t* mask, double* sums1, float* loss,
                                void* stream) {
    if (H <= 0 || W <= 0 || !normal_hwc || !mask || !sums1 || !loss) { vcr_set_error("vcr_curv_forward: bad arguments"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    VCR_HIP_CHECK(hipMemsetAsync(sums1, 0, (1 + VCR_NSLOT) * sizeof(double), st));
    hipLaunchKernelGGL(curv_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, H, W, normal_hwc, mask, sums1);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, st, 1, sums1, 0, 1.0 / ((double)H * (double)W), loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_curv_backward(int H, int W, const float* normal_hwc, const uint8_t* mask, const float* gout, float* dnormal,
                                 void* stream) {
    if (H <= 0 || W <= 0 || !normal_hwc || !mask || !gout || !dnormal) { vcr_set_error("vcr_curv_backward: bad arguments"); return 1; }
    hipLaunchKernelGGL(curv_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, H, W, normal_hwc, mask,
                       gout, dnormal);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// xyz == NULL: no bounding-box mask (every Gaussian counts)
extern "C" int vcr_entropy_forward(int N, const float* opacity_raw, const float* xyz, const float* trans, const float* scale,
                                   double* sums3, float* loss, void* stream) {
    if (N < 0 || !opacity_raw || !sums3 || !loss || (xyz && (!trans || !scale))) { vcr_set_error("vcr_entropy_forward: bad arguments"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    VCR_HIP_CHECK(hipMemsetAsync(sums3, 0, 3 * (1 + VCR_NSLOT) * sizeof(double), st));
    if (N > 0)
        hipLaunchKernelGGL(entropy_fwd_kernel, dim3(min((N + 255) / 256, 2048)), dim3(256), 0, st, N, opacity_raw, xyz, trans, scale, sums3);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, st, 3, sums3, 1, 1.0, loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_entropy_backward(int N, const float* opacity_raw, const float* xyz, const float* trans, const float* scale,
                                    const double* sums3, const float* gout, float* dopacity_raw, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(entropy_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, opacity_raw, xyz, trans, scale,
                       sums3, gout, dopacity_raw);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------- depth -> TSDF input (tools/graphics_utils.py:134-141, tools/depth2mesh.py:37-52) --------------------------
// One pass over the rendered depth: zero it where alpha < alpha_thres, where the (optional) ground-truth alpha mask is
// < 0.5, and where the back-projected WORLD point lies outside the normalised bounding box; optionally emit the camera- /
// world-space points (depth2point) of the depth as given.  c2w: row-major 4x4 camera-to-world matrix.
namespace {
struct Mat34 { float m[12]; };

__global__ void __launch_bounds__(256) tsdf_input_kernel(int H, int W, Intr k, Mat34 c2w, const float* __restrict__ trans,
                                                         const float* __restrict__ scale, const float* __restrict__ depth_in,
                                                         const float* __restrict__ alpha, float alpha_thres,
                                                         const float* __restrict__ gt_alpha, float* __restrict__ depth_out,
                                                         float* __restrict__ xyz_cam, float* __restrict__ xyz_world) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const float z0 = depth_in[p];
    const float ux = ((float)x + 0.5f - k.cx) / k.fx, uy = ((float)y + 0.5f - k.cy) / k.fy;
    if (xyz_cam || xyz_world) {
        const float X = ux * z0, Y = uy * z0;
        if (xyz_cam) { xyz_cam[3 * p] = X; xyz_cam[3 * p + 1] = Y; xyz_cam[3 * p + 2] = z0; }
        if (xyz_world) {
#pragma unroll
            for (int r = 0; r < 3; ++r) xyz_world[3 * p + r] = c2w.m[4 * r] * X + c2w.m[4 * r + 1] * Y + c2w.m[4 * r + 2] * z0 + c2w.m[4 * r + 3];
        }
    }
    if (!depth_out) return;
    float z = z0;
    if (gt_alpha && gt_alpha[p] < 0.5f) z = 0.f;
    if (alpha && alpha[p] < alpha_thres) z = 0.f;
    if (trans) {                                            // the box test sees the point of the ALREADY masked depth
        const float X = ux * z, Y = uy * z;
        bool in = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w = c2w.m[4 * r] * X + c2w.m[4 * r + 1] * Y + c2w.m[4 * r + 2] * z + c2w.m[4 * r + 3];
            in = in && fabsf((w - trans[r]) / scale[r]) < 1.f;
        }
        if (!in) z = 0.f;
    }
    depth_out[p] = z;
}
}  // namespace

extern "C" int vcr_tsdf_depth_input(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_rowmajor16,
                                    const float* trans, const float* scale, const float* depth, const float* alpha,
                                    float alpha_thres, const float* gt_alpha, float* depth_out, float* xyz_cam,
                                    float* xyz_world, void* stream) {
    if (H <= 0 || W <= 0 || !c2w_rowmajor16 || !depth || (trans && !scale)) { vcr_set_error("vcr_tsdf_depth_input: bad arguments"); return 1; }
    Mat34 m;
    for (int i = 0; i < 12; ++i) m.m[i] = c2w_rowmajor16[i];
    hipLaunchKernelGGL(tsdf_input_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, H, W,
                       Intr{fx, fy, cx, cy}, m, trans, scale, depth, alpha, alpha_thres, gt_alpha, depth_out, xyz_cam, xyz_world);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------- semantic loss (gaussian_renderer/__init__.py:146-148, trainer.py:304-307) ------------------------------------
// The reference pushes the rendered semantic feature planes through a 1x1 Conv2d classifier and takes
// F.cross_entropy(logits, labels) / log(num_cls).  Both are per-pixel arithmetic on S <= 4 features and K <= 8 classes, so
// classifier + log-softmax + NLL run in one streaming kernel each way (the backward also yields the classifier's gradient).
// sums layout: [1 result][VCR_NSLOT slots] for the forward; the backward accumulates K*(S+1) fp32 sums by block reduction +
// atomics into dW [K,S] and db [K] (zeroed here).
namespace {
constexpr int SEM_MAX_S = 4, SEM_MAX_K = 8;

struct SemCls { float W[SEM_MAX_K * SEM_MAX_S]; float b[SEM_MAX_K]; };

__device__ __forceinline__ void sem_logits(const SemCls& c, int S, int K, const float* __restrict__ sem, size_t P, size_t p,
                                           float f[SEM_MAX_S], float l[SEM_MAX_K], float& lse) {
    for (int s = 0; s < S; ++s) f[s] = sem[s * P + p];
    float m = -3.4e38f;
    for (int k = 0; k < K; ++k) {
        float v = c.b[k];
        for (int s = 0; s < S; ++s) v += c.W[k * S + s] * f[s];
        l[k] = v; m = fmaxf(m, v);
    }
    float z = 0.f;
    for (int k = 0; k < K; ++k) z += __expf(l[k] - m);
    lse = m + __logf(z);
}

__device__ __forceinline__ void load_cls(SemCls& c, int S, int K, const float* __restrict__ W, const float* __restrict__ b) {
    for (int i = 0; i < K * S; ++i) c.W[i] = W[i];           // wave-uniform loads of <= 40 numbers
    for (int i = 0; i < K; ++i) c.b[i] = b[i];
}

__global__ void __launch_bounds__(256) sem_ce_fwd_kernel(size_t P, int S, int K, const float* __restrict__ Wd,
                                                         const float* __restrict__ bd, const float* __restrict__ sem,
                                                         const long long* __restrict__ labels, double* __restrict__ sums) {
    SemCls c;
    load_cls(c, S, K, Wd, bd);
    float v[1] = {0.f};
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        float f[SEM_MAX_S], l[SEM_MAX_K], lse;
        sem_logits(c, S, K, sem, P, p, f, l, lse);
        const int t = (int)labels[p];
        v[0] += lse - (t >= 0 && t < K ? l[t] : lse);
    }
    block_accumulate<1>(sums + 1, v);
}

__global__ void __launch_bounds__(256) sem_ce_bwd_kernel(size_t P, int S, int K, const float* __restrict__ Wd,
                                                         const float* __restrict__ bd, const float* __restrict__ sem,
                                                         const long long* __restrict__ labels, const float* __restrict__ gout,
                                                         float scale, float* __restrict__ dsem, float* __restrict__ dW,
                                                         float* __restrict__ db) {
    __shared__ float s_red[4][SEM_MAX_K * (SEM_MAX_S + 1)];
    SemCls c;
    load_cls(c, S, K, Wd, bd);
    float acc[SEM_MAX_K * (SEM_MAX_S + 1)];
    const int NV = K * (S + 1);
    for (int i = 0; i < SEM_MAX_K * (SEM_MAX_S + 1); ++i) acc[i] = 0.f;
    const float g = gout[0] * scale;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        float f[SEM_MAX_S], l[SEM_MAX_K], lse;
        sem_logits(c, S, K, sem, P, p, f, l, lse);
        const int t = (int)labels[p];
        float ds[SEM_MAX_S] = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < K) {
            for (int k = 0; k < K; ++k) {
                const float dl = g * (__expf(l[k] - lse) - (k == t ? 1.f : 0.f));
                for (int s = 0; s < S; ++s) { ds[s] += dl * c.W[k * S + s]; acc[k * (S + 1) + s] += dl * f[s]; }
                acc[k * (S + 1) + S] += dl;
            }
        }
        for (int s = 0; s < S; ++s) dsem[s * P + p] = ds[s];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = 0; i < NV; ++i) {
        const float t = wave_sum(acc[i]);
        if (lane == 0) s_red[wv][i] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        const int i = threadIdx.x, k = i / (S + 1), s = i - k * (S + 1);
        const float t = s_red[0][i] + s_red[1][i] + s_red[2][i] + s_red[3][i];
        if (t != 0.f) atomicAdd(s < S ? dW + k * S + s : db + k, t);
    }
}
}  // namespace

// sem: [S,P] feature planes (rows of the rasterizer output), W [K,S] and b [K]: the classifier's parameters (device),
// labels: device int64 [P].  loss (device float[1]) = mean_p CE(p) / log(K).
extern "C" int vcr_semantic_ce_forward(long long P, int S, int K, const float* sem, const float* W, const float* b,
                                       const long long* labels, double* sums1, float* loss, void* stream) {
    if (P <= 0 || S < 1 || S > SEM_MAX_S || K < 2 || K > SEM_MAX_K || !sem || !W || !b || !labels || !sums1 || !loss) {
        vcr_set_error("vcr_semantic_ce_forward: bad arguments (S <= %d, 2 <= K <= %d)", SEM_MAX_S, SEM_MAX_K); return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    VCR_HIP_CHECK(hipMemsetAsync(sums1, 0, (1 + VCR_NSLOT) * sizeof(double), st));
    const int blocks = (int)((P + 255) / 256 < 4096 ? (P + 255) / 256 : 4096);
    hipLaunchKernelGGL(sem_ce_fwd_kernel, dim3(blocks), dim3(256), 0, st, (size_t)P, S, K, W, b, sem, labels, sums1);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(64), 0, st, 1, sums1, 0, 1.0 / ((double)P * log((double)K)), loss);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// dsem [S,P], dW [K,S], db [K] (device, fp32; dW / db are zeroed here)
extern "C" int vcr_semantic_ce_backward(long long P, int S, int K, const float* sem, const float* W, const float* b,
                                        const long long* labels, const float* gout, float* dsem, float* dW, float* db,
                                        void* stream) {
    if (P <= 0 || S < 1 || S > SEM_MAX_S || K < 2 || K > SEM_MAX_K || !sem || !W || !b || !labels || !gout || !dsem || !dW || !db) {
        vcr_set_error("vcr_semantic_ce_backward: bad arguments"); return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    VCR_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * K * S, st));
    VCR_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * K, st));
    const int blocks = (int)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
    hipLaunchKernelGGL(sem_ce_bwd_kernel, dim3(blocks), dim3(256), 0, st, (size_t)P, S, K, W, b, sem, labels, gout,
                       (float)(1.0 / ((double)P * log((double)K))), dsem, dW, db);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
