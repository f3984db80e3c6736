// Per-Gaussian device functions shared by the parameter kernels (model_ops.hip) and the fused backward tail of the
// rasterizer (preprocess.hip): fused activation and its adjoint, one Adam update, and the static tail of an iteration for ONE
// Gaussian.  Each translation unit gets its own copy (no relocatable device code).
#pragma once
#include "vcr_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void quat_R(float r, float x, float y, float z, float R[9]) {
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// ---- per-Gaussian pieces, shared by the stand-alone kernels and the fused geometry step ---------------------------------
struct ActOut { float s[3]; float4 q; float o; float n[3]; uint8_t aux; };

// exp / normalize / sigmoid + shortest-axis normal, flipped to face away from the camera and rotated into camera space
// (scene/gaussian_model.py:125-192, gaussian_renderer/__init__.py:95-101).  aux: bits0-1 = shortest axis, bit2 = flipped
__device__ __forceinline__ ActOut activate_one(const float l[3], float4 qr, float oraw, const float p[3], const float* __restrict__ campos,
                                               const float* __restrict__ Rw2c, bool want_normal) {
    ActOut a;
    a.s[0] = expf(l[0]); a.s[1] = expf(l[1]); a.s[2] = expf(l[2]);
    const float inv = 1.f / fmaxf(sqrtf(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w), 1e-12f);
    a.q = make_float4(qr.x * inv, qr.y * inv, qr.z * inv, qr.w * inv);
    a.o = 1.f / (1.f + expf(-oraw));
    a.n[0] = a.n[1] = a.n[2] = 0.f; a.aux = 0;
    if (!want_normal) return a;
    int axis = 0;                              // torch.argmin: first minimum
    float sm = a.s[0];
    if (a.s[1] < sm) { sm = a.s[1]; axis = 1; }
    if (a.s[2] < sm) { sm = a.s[2]; axis = 2; }
    float R[9];
    quat_R(a.q.x, a.q.y, a.q.z, a.q.w, R);
    float n0 = R[axis], n1 = R[3 + axis], n2 = R[6 + axis];
    const float vx = p[0] - campos[0], vy = p[1] - campos[1], vz = p[2] - campos[2];
    const bool keep = (vx * n0 + vy * n1 + vz * n2) > 0.f;
    if (!keep) { n0 = -n0; n1 = -n1; n2 = -n2; }
    a.n[0] = Rw2c[0] * n0 + Rw2c[1] * n1 + Rw2c[2] * n2;
    a.n[1] = Rw2c[3] * n0 + Rw2c[4] * n1 + Rw2c[5] * n2;
    a.n[2] = Rw2c[6] * n0 + Rw2c[7] * n1 + Rw2c[8] * n2;
    a.aux = (uint8_t)(axis | (keep ? 0 : 4));
    return a;
}

// adjoint of activate_one: gradients w.r.t. the raw scaling (gs), rotation (gq), opacity (go).  `has_*`: which upstream
// gradients exist; ds / dq / dop / dn: upstream gradients; extra: second gradient path into the raw scaling (l1_scale).
// Arithmetic in fp64 (round 5, like the projection backward in front of it: the quaternion adjoints project out the component
// along q twice -- differences of nearly equal products -- and one lane does this once per Gaussian beside ~350 B of traffic).
__device__ __forceinline__ void activate_bwd_one(const float l[3], float4 qr, float oraw, const float* __restrict__ Rw2c, uint8_t aux,
                                                 bool has_s, const float ds[3], bool has_q, float4 dq, bool has_o, float dop,
                                                 bool has_n, const float dn[3], const float extra[3], float gs[3], float4& gq,
                                                 float& go, bool n_world = false) {
    typedef double D;
#pragma unroll
    for (int k = 0; k < 3; ++k) gs[k] = (has_s ? ds[k] * expf(l[k]) : 0.f) + extra[k];
    const float o = 1.f / (1.f + expf(-oraw));
    go = has_o ? dop * o * (1.f - o) : 0.f;
    const D qx = qr.x, qy = qr.y, qz = qr.z, qw = qr.w;
    const D nrm = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    const D inv = 1.0 / (nrm > 1e-12 ? nrm : 1e-12);
    const D r = qx * inv, x = qy * inv, y = qz * inv, z = qw * inv;
    D g[4] = {0, 0, 0, 0};                             // gradient w.r.t. the unit quaternion
    if (has_q) { g[0] = dq.x; g[1] = dq.y; g[2] = dq.z; g[3] = dq.w; }
    if (has_n) {
        const int axis = aux & 3;
        const D sgn = (aux & 4) ? -1.0 : 1.0;
        const D c0 = dn[0], c1 = dn[1], c2 = dn[2];
        // n_cam = Rw2c * (sgn * R[:,axis]); n_world: `dn` already is the gradient w.r.t. R[:,axis] (summed over the ranks' views)
        const D w0 = n_world ? c0 : sgn * ((D)Rw2c[0] * c0 + (D)Rw2c[3] * c1 + (D)Rw2c[6] * c2);
        const D w1 = n_world ? c1 : sgn * ((D)Rw2c[1] * c0 + (D)Rw2c[4] * c1 + (D)Rw2c[7] * c2);
        const D w2 = n_world ? c2 : sgn * ((D)Rw2c[2] * c0 + (D)Rw2c[5] * c1 + (D)Rw2c[8] * c2);
        D dR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        dR[axis] = w0; dR[3 + axis] = w1; dR[6 + axis] = w2;
        D h[4];
        h[0] = 2.0 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        h[1] = 2.0 * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0 * x * dR[8]);
        h[2] = 2.0 * (-2.0 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0 * y * dR[8]);
        h[3] = 2.0 * (-2.0 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        // build_rotation re-normalises its (already unit) input: project onto the tangent space
        const D hd = h[0] * r + h[1] * x + h[2] * y + h[3] * z;
        g[0] += h[0] - r * hd; g[1] += h[1] - x * hd; g[2] += h[2] - y * hd; g[3] += h[3] - z * hd;
    }
    // q = raw/|raw|
    const D gd = g[0] * r + g[1] * x + g[2] * y + g[3] * z;
    gq = make_float4((float)((g[0] - r * gd) * inv), (float)((g[1] - x * gd) * inv), (float)((g[2] - y * gd) * inv),
                     (float)((g[3] - z * gd) * inv));
}

// one Adam update (torch.optim.Adam, eps outside the bias-corrected root): step = lr / bc1
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, float b1, float b2, float eps, float step,
                                         float bc2_sqrt) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step * (m / (sqrtf(v) / bc2_sqrt + eps));
}

struct GeomBias { float st[4], bc2[4]; };       // per group (xyz, scaling, rotation, opacity): lr / (1 - b1^t), sqrt(1 - b2^t)

// host: the bias-corrected step sizes of the four geometry groups (same arithmetic as vcr_adam_step)
inline void vcr_geometry_bias(const VcrGeometryStep& a, GeomBias& gb) {
    const int steps[4] = {a.step_xyz > 0 ? a.step_xyz : 1, a.step_scaling, a.step_rotation, a.step_opacity};
    const float lrs[4] = {a.lr_xyz, a.lr_scaling, a.lr_rotation, a.lr_opacity};
    for (int k = 0; k < 4; ++k) {
        const double bc1 = 1.0 - pow((double)a.beta1, steps[k]), bc2 = 1.0 - pow((double)a.beta2, steps[k]);
        gb.st[k] = lrs[k] / (float)bc1;
        gb.bc2[k] = (float)sqrt(bc2);
    }
}

// upstream gradients of ONE Gaussian held in registers (REGS form of geometry_step_one: they come straight out of the
// projection backward instead of out of the arrays VcrGeometryStep names)
struct TailGrads {
    float dp[3];            // dL/d mean (world)
    float ds[3];            // dL/d activated scales
    float dn[3];            // dL/d camera-space normal
    float dm2[2];           // dL/d screen position used by the densification statistics
    float4 dq;              // dL/d unit quaternion
    float dop;              // dL/d activated opacity
    int radius;             // screen radius of this render (> 0: visible)
    bool has_n;             // normals were an input of the render
};

// The static tail for Gaussian i: activation backward (+ the l1_scale gradient) -> densification statistics -> Adam on
// xyz / scaling / rotation / opacity -> activation for the NEXT iteration's camera.  REGS = false: upstream gradients from
// a.d_* / a.grad2d / a.radii (NULL = absent); REGS = true: from `t` (all present; statistics when a.accum != NULL).
template <bool REGS>
__device__ __forceinline__ void geometry_step_one(const VcrGeometryStep& a, const GeomBias& gb, int i, const TailGrads& t) {
    const float st_xyz = gb.st[0], st_scaling = gb.st[1], st_rotation = gb.st[2], st_opacity = gb.st[3];
    const size_t i3 = 3 * (size_t)i;
    float l[3] = {a.scaling[i3], a.scaling[i3 + 1], a.scaling[i3 + 2]};
    float4 qr = reinterpret_cast<float4*>(a.rotation)[i];
    float oraw = a.opacity[i];
    float p[3] = {a.xyz[i3], a.xyz[i3 + 1], a.xyz[i3 + 2]};
    // ---- gradients w.r.t. the raw parameters
    float ds[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f}, ex[3] = {0.f, 0.f, 0.f};
    const bool has_s = REGS || a.d_scales != nullptr, has_q = REGS || a.d_rots != nullptr, has_o = REGS || a.d_opac != nullptr;
    const bool has_n = REGS ? t.has_n : a.d_normals != nullptr;
    const bool has_p = REGS || a.d_means3D != nullptr;
    if (REGS) {
        ds[0] = t.ds[0]; ds[1] = t.ds[1]; ds[2] = t.ds[2];
        if (has_n) { dn[0] = t.dn[0]; dn[1] = t.dn[1]; dn[2] = t.dn[2]; }
    } else {
        if (a.d_scales) { ds[0] = a.d_scales[i3]; ds[1] = a.d_scales[i3 + 1]; ds[2] = a.d_scales[i3 + 2]; }
        if (a.d_normals) { dn[0] = a.d_normals[i3]; dn[1] = a.d_normals[i3 + 1]; dn[2] = a.d_normals[i3 + 2]; }
    }
    if (a.scale_reg_sums) {          // l1_scale (trainer.py:243-245): d/d raw of mean over the box of min_axis exp(raw)
        const bool in = fabsf((p[0] - a.trans[0]) / a.scale[0]) < 1.f && fabsf((p[1] - a.trans[1]) / a.scale[1]) < 1.f &&
                        fabsf((p[2] - a.trans[2]) / a.scale[2]) < 1.f;
        if (in) {
            const int k = (l[0] <= l[1] && l[0] <= l[2]) ? 0 : (l[1] <= l[2] ? 1 : 2);           // first minimum, like torch.min
            ex[k] = a.scale_reg_gout[0] / (float)a.scale_reg_sums[2] * __expf(fminf(l[0], fminf(l[1], l[2])));
        }
    }
    float gs[3], go;
    float4 gq;
    float4 dq_up = REGS ? t.dq : (a.d_rots ? reinterpret_cast<const float4*>(a.d_rots)[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    float dop_up = REGS ? t.dop : (a.d_opac ? a.d_opac[i] : 0.f);
    // data parallel: the upstream gradients are sums over the ranks' views -- their mean is what Adam sees; the l1_scale term
    // (`ex`, a function of the replicated parameters only) is added once, unscaled
    const float gsc = a.grad_scale;
    ds[0] *= gsc; ds[1] *= gsc; ds[2] *= gsc; dn[0] *= gsc; dn[1] *= gsc; dn[2] *= gsc;
    dq_up.x *= gsc; dq_up.y *= gsc; dq_up.z *= gsc; dq_up.w *= gsc; dop_up *= gsc;
    activate_bwd_one(l, qr, oraw, a.Rw2c, has_n ? a.aux[i] : (uint8_t)0, has_s, ds, has_q, dq_up, has_o, dop_up, has_n, dn, ex,
                     gs, gq, go, a.normals_world != 0);
    // ---- densification statistics (scene/gaussian_model.py:669-671, trainer.py:345)
    if (REGS ? a.accum != nullptr : a.grad2d != nullptr) {
        const int r = REGS ? t.radius : a.radii[i];
        if (r > 0) {
            const float gx = REGS ? t.dm2[0] : a.grad2d[i3], gy = REGS ? t.dm2[1] : a.grad2d[i3 + 1];
            a.accum[i] += sqrtf(gx * gx + gy * gy);
            a.denom[i] += 1.f;
            a.max_radii[i] = fmaxf(a.max_radii[i], (float)r);
        }
    }
    // ---- Adam
    const float b1 = a.beta1, b2 = a.beta2, eps = a.eps;
    if (has_p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float m = a.m_xyz[i3 + k], v = a.v_xyz[i3 + k];
            adam_one(p[k], m, v, gsc * (REGS ? t.dp[k] : a.d_means3D[i3 + k]), b1, b2, eps, st_xyz, gb.bc2[0]);
            a.m_xyz[i3 + k] = m; a.v_xyz[i3 + k] = v; a.xyz[i3 + k] = p[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float m = a.m_scaling[i3 + k], v = a.v_scaling[i3 + k];
        adam_one(l[k], m, v, gs[k], b1, b2, eps, st_scaling, gb.bc2[1]);
        a.m_scaling[i3 + k] = m; a.v_scaling[i3 + k] = v; a.scaling[i3 + k] = l[k];
    }
    {
        float4 m = reinterpret_cast<float4*>(a.m_rotation)[i], v = reinterpret_cast<float4*>(a.v_rotation)[i];
        adam_one(qr.x, m.x, v.x, gq.x, b1, b2, eps, st_rotation, gb.bc2[2]);
        adam_one(qr.y, m.y, v.y, gq.y, b1, b2, eps, st_rotation, gb.bc2[2]);
        adam_one(qr.z, m.z, v.z, gq.z, b1, b2, eps, st_rotation, gb.bc2[2]);
        adam_one(qr.w, m.w, v.w, gq.w, b1, b2, eps, st_rotation, gb.bc2[2]);
        reinterpret_cast<float4*>(a.m_rotation)[i] = m; reinterpret_cast<float4*>(a.v_rotation)[i] = v;
        reinterpret_cast<float4*>(a.rotation)[i] = qr;
    }
    {
        float m = a.m_opacity[i], v = a.v_opacity[i];
        adam_one(oraw, m, v, go, b1, b2, eps, st_opacity, gb.bc2[3]);
        a.m_opacity[i] = m; a.v_opacity[i] = v; a.opacity[i] = oraw;
    }
    // ---- activation of the updated parameters for the next render
    if (a.next_scales) {
        const ActOut n = activate_one(l, qr, oraw, p, a.next_campos, a.next_Rw2c, a.next_normals != nullptr);
        a.next_scales[i3] = n.s[0]; a.next_scales[i3 + 1] = n.s[1]; a.next_scales[i3 + 2] = n.s[2];
        reinterpret_cast<float4*>(a.next_rots)[i] = n.q;
        a.next_opac[i] = n.o;
        if (a.next_normals) {
            a.next_normals[i3] = n.n[0]; a.next_normals[i3 + 1] = n.n[1]; a.next_normals[i3 + 2] = n.n[2];
            a.next_aux[i] = n.aux;
        }
    }
}

}  // namespace
