// K1 / K8: per-Gaussian projection stage and its adjoint, one lane per Gaussian (HBM-streaming).
//
// Contract: SURVEY.md Appendix A.3 steps 1-7, restated in oracle/raster_torch.py::preprocess.
// Reference call sites: gaussian_renderer/__init__.py:61-120 (inputs), :83-87 (colour rule),
// tools/general_utils.py:98-130 (covariance from scale/rotation), scene/cameras.py:68-70 (matrices).
#include "vcr_common.h"
#include "model_math.h"
#include "composite_math.h"
#include <stdlib.h>

namespace {

struct Cam {
    float V[16], P[16], c[3];
};

__device__ __forceinline__ Cam load_cam(const VcrRasterArgs& a) {
    Cam cam;
#pragma unroll
    for (int k = 0; k < 16; ++k) { cam.V[k] = a.viewmatrix[k]; cam.P[k] = a.projmatrix[k]; }
    cam.c[0] = a.campos[0]; cam.c[1] = a.campos[1]; cam.c[2] = a.campos[2];
    return cam;
}

// (T = float: the forward; T = double: the backward, see preprocess_bwd_kernel)
template <typename T>
__device__ __forceinline__ void quat_to_R(const float4 q, T R[9]) {
    const T r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = T(1) - T(2) * (y * y + z * z); R[1] = T(2) * (x * y - r * z); R[2] = T(2) * (x * z + r * y);
    R[3] = T(2) * (x * y + r * z); R[4] = T(1) - T(2) * (x * x + z * z); R[5] = T(2) * (y * z - r * x);
    R[6] = T(2) * (x * z - r * y); R[7] = T(2) * (y * z + r * x); R[8] = T(1) - T(2) * (x * x + y * y);
}

// Sigma (xx,xy,xz,yy,yz,zz)
template <typename T>
__device__ __forceinline__ void load_cov3d(const VcrRasterArgs& a, int i, T S[6], T R[9], T s[3]) {
    if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = a.cov3D_precomp[6 * (size_t)i + k];
        return;
    }
    const float4 q = reinterpret_cast<const float4*>(a.rotations)[i];
    quat_to_R<T>(q, R);
    s[0] = (T)a.scales[3 * (size_t)i + 0] * (T)a.scale_modifier;
    s[1] = (T)a.scales[3 * (size_t)i + 1] * (T)a.scale_modifier;
    s[2] = (T)a.scales[3 * (size_t)i + 2] * (T)a.scale_modifier;
    const T s0 = s[0] * s[0], s1 = s[1] * s[1], s2 = s[2] * s[2];
    S[0] = R[0] * R[0] * s0 + R[1] * R[1] * s1 + R[2] * R[2] * s2;
    S[1] = R[0] * R[3] * s0 + R[1] * R[4] * s1 + R[2] * R[5] * s2;
    S[2] = R[0] * R[6] * s0 + R[1] * R[7] * s1 + R[2] * R[8] * s2;
    S[3] = R[3] * R[3] * s0 + R[4] * R[4] * s1 + R[5] * R[5] * s2;
    S[4] = R[3] * R[6] * s0 + R[4] * R[7] * s1 + R[5] * R[8] * s2;
    S[5] = R[6] * R[6] * s0 + R[7] * R[7] * s1 + R[8] * R[8] * s2;
}

template <typename T>
struct ProjT {
    T t[3];              // view-space position
    T u, v;              // clamped tx/tz, ty/tz
    bool uc, vc;         // clamp active
    T M0[3], M1[3];      // rows of J * Rv
    T fx, fy;
};
typedef ProjT<float> Proj;

template <typename T>
__device__ __forceinline__ void project(const VcrRasterArgs& a, const Cam& cam, const T p[3], ProjT<T>& pr) {
    const float* V = cam.V;
    pr.t[0] = p[0] * (T)V[0] + p[1] * (T)V[4] + p[2] * (T)V[8] + (T)V[12];
    pr.t[1] = p[0] * (T)V[1] + p[1] * (T)V[5] + p[2] * (T)V[9] + (T)V[13];
    pr.t[2] = p[0] * (T)V[2] + p[1] * (T)V[6] + p[2] * (T)V[10] + (T)V[14];
    pr.fx = (T)a.W / (T(2) * (T)a.tanfovx);
    pr.fy = (T)a.H / (T(2) * (T)a.tanfovy);
}

template <typename T>
__device__ __forceinline__ void jacobian_rows(const VcrRasterArgs& a, const Cam& cam, ProjT<T>& pr) {
    const float* V = cam.V;
    const T tz = pr.t[2], itz = T(1) / tz;
    const T limx = (T)1.3f * (T)a.tanfovx, limy = (T)1.3f * (T)a.tanfovy;
    const T ru = pr.t[0] * itz, rv = pr.t[1] * itz;
    pr.uc = (ru < -limx) || (ru > limx);
    pr.vc = (rv < -limy) || (rv > limy);
    pr.u = ru < -limx ? -limx : (ru > limx ? limx : ru);
    pr.v = rv < -limy ? -limy : (rv > limy ? limy : rv);
    const T J00 = pr.fx * itz, J02 = -pr.fx * pr.u * itz;
    const T J11 = pr.fy * itz, J12 = -pr.fy * pr.v * itz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {           // Rv[c][k] = V[k*4+c]
        pr.M0[k] = J00 * (T)V[k * 4 + 0] + J02 * (T)V[k * 4 + 2];
        pr.M1[k] = J11 * (T)V[k * 4 + 1] + J12 * (T)V[k * 4 + 2];
    }
}

template <typename T>
__device__ __forceinline__ void sym_mul(const T S[6], const T m[3], T o[3]) {
    o[0] = S[0] * m[0] + S[1] * m[1] + S[2] * m[2];
    o[1] = S[1] * m[0] + S[3] * m[1] + S[4] * m[2];
    o[2] = S[2] * m[0] + S[4] * m[1] + S[5] * m[2];
}

// SH basis of tools/sh_utils.py:57-112 with signs folded in; optional derivatives w.r.t. the direction.
template <bool GRAD>
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float b[16], float bx[16], float by[16],
                                         float bz[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        b[k] = 0.f;
        if (GRAD) { bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
    }
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (GRAD) { by[1] = -SH_C1; bz[2] = SH_C1; bx[3] = -SH_C1; }
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
    b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
    if (GRAD) {
        bx[4] = SH_C2_0 * y; by[4] = SH_C2_0 * x;
        by[5] = SH_C2_1 * z; bz[5] = SH_C2_1 * y;
        bx[6] = -2.f * SH_C2_2 * x; by[6] = -2.f * SH_C2_2 * y; bz[6] = 4.f * SH_C2_2 * z;
        bx[7] = SH_C2_3 * z; bz[7] = SH_C2_3 * x;
        bx[8] = 2.f * SH_C2_4 * x; by[8] = -2.f * SH_C2_4 * y;
    }
    if (deg < 3) return;
    b[9] = SH_C3_0 * y * (3.f * xx - yy);
    b[10] = SH_C3_1 * xy * z;
    b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
    b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
    b[14] = SH_C3_5 * z * (xx - yy);
    b[15] = SH_C3_6 * x * (xx - 3.f * yy);
    if (GRAD) {
        bx[9] = SH_C3_0 * 6.f * xy; by[9] = SH_C3_0 * (3.f * xx - 3.f * yy);
        bx[10] = SH_C3_1 * yz; by[10] = SH_C3_1 * xz; bz[10] = SH_C3_1 * xy;
        bx[11] = SH_C3_2 * (-2.f * xy); by[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); bz[11] = SH_C3_2 * 8.f * yz;
        bx[12] = SH_C3_3 * (-6.f * xz); by[12] = SH_C3_3 * (-6.f * yz); bz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
        bx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = SH_C3_4 * (-2.f * xy); bz[13] = SH_C3_4 * 8.f * xz;
        bx[14] = SH_C3_5 * 2.f * xz; by[14] = SH_C3_5 * (-2.f * yz); bz[14] = SH_C3_5 * (xx - yy);
        bx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = SH_C3_6 * (-6.f * xy);
    }
}

// SH -> RGB of one Gaussian (before the + 0.5 / clamp) together with its derivative w.r.t. the MEAN through the normalised view
// direction d = (mean - campos) il:  M[c] = il (I - d d^T) sum_k grad(basis_k)(d) sh[k][c]  (GeomState::cjac).  `SH` is indexable
// as SH[3 k + c] (a global row or the block's LDS row).
#define VCR_SH_COLOUR_JAC(DEG, NB, SH, DX, DY, DZ, IL, C0, C1, C2, MOUT)                                                  \
    do {                                                                                                                    \
        float b_[16], bx_[16], by_[16], bz_[16];                                                                            \
        sh_basis<true>((DEG), (DX), (DY), (DZ), b_, bx_, by_, bz_);                                                         \
        float jx_[3] = {0.f, 0.f, 0.f}, jy_[3] = {0.f, 0.f, 0.f}, jz_[3] = {0.f, 0.f, 0.f};                                \
        float cc_[3] = {0.5f, 0.5f, 0.5f};                                                                                  \
_Pragma("unroll")                                                                                                           \
        for (int k = 0; k < 16; ++k) {                                                                                      \
            if (k < (NB)) {                                                                                                 \
_Pragma("unroll")                                                                                                           \
                for (int c = 0; c < 3; ++c) {                                                                               \
                    const float v_ = (SH)[3 * k + c];                                                                       \
                    cc_[c] += b_[k] * v_; jx_[c] += bx_[k] * v_; jy_[c] += by_[k] * v_; jz_[c] += bz_[k] * v_;              \
                }                                                                                                           \
            }                                                                                                               \
        }                                                                                                                   \
        C0 = cc_[0]; C1 = cc_[1]; C2 = cc_[2];                                                                              \
_Pragma("unroll")                                                                                                           \
        for (int c = 0; c < 3; ++c) {                                                                                       \
            const float dot_ = (DX) * jx_[c] + (DY) * jy_[c] + (DZ) * jz_[c];                                               \
            (MOUT)[c] = make_float4((jx_[c] - (DX) * dot_) * (IL), (jy_[c] - (DY) * dot_) * (IL), (jz_[c] - (DZ) * dot_) * (IL), 0.f); \
        }                                                                                                                   \
    } while (0)


// ---- cooperative SH staging ------------------------------------------------------------------------------
// A lane-per-Gaussian read of 48 consecutive floats is a 192-byte-stride access: every load instruction of
// a wave touches 64 different lines and the 32 KiB L1 thrashes.  Instead the 256 Gaussians of a block are
// streamed with fully coalesced 16-byte loads into LDS (row stride 49 dwords -> conflict-free per-lane rows)
// and each lane then reads its own row.  Works for the combined [N,16,3] layout and for the reference's split
// storage (_features_dc [N,1,3] + _features_rest [N,15,3]) so no torch.cat is needed.
#ifndef VCR_SIDE_NT
#define VCR_SIDE_NT true
#endif
#define SH_ROW 49
#define SH_K 16

__device__ __forceinline__ void coop_copy_in(const float* __restrict__ src, int total, int per, int off, float* s) {
    // src: `total` contiguous floats = rows of `per` floats; row g goes to s[g*SH_ROW + off ...]
    const int n4 = (reinterpret_cast<uintptr_t>(src) & 15) ? 0 : (total >> 2);      // unaligned views: scalar path
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int e0 = threadIdx.x; e0 < n4; e0 += 4 * 256) {                             // four 16-byte loads in flight per lane
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e0 + u * 256 < n4) v[u] = s4[e0 + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e4 = e0 + u * 256;
            if (e4 >= n4) continue;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e4 * 4 + j, g = e / per;
                s[g * SH_ROW + off + (e - g * per)] = vv[j];
            }
        }
    }
    for (int e = (n4 << 2) + threadIdx.x; e < total; e += 256) {
        const int g = e / per;
        s[g * SH_ROW + off + (e - g * per)] = src[e];
    }
}

__device__ __forceinline__ void coop_copy_out(float* __restrict__ dst, int total, int per, int off, const float* s) {
    const int n4 = (reinterpret_cast<uintptr_t>(dst) & 15) ? 0 : (total >> 2);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int e4 = threadIdx.x; e4 < n4; e4 += 256) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e4 * 4 + j, g = e / per;
            vv[j] = s[g * SH_ROW + off + (e - g * per)];
        }
        d4[e4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int e = (n4 << 2) + threadIdx.x; e < total; e += 256) {
        const int g = e / per;
        dst[e] = s[g * SH_ROW + off + (e - g * per)];
    }
}

__device__ __forceinline__ void stage_sh_in(const VcrRasterArgs& a, int base, int cnt, float* s) {
    if (a.shs_rest) {
        coop_copy_in(a.shs + (size_t)base * 3, cnt * 3, 3, 0, s);
        coop_copy_in(a.shs_rest + (size_t)base * 45, cnt * 45, 45, 3, s);
    } else {
        coop_copy_in(a.shs + (size_t)base * 48, cnt * 48, 48, 0, s);
    }
}

// COLOUR = false: geometry only (the SH -> RGB evaluation runs as colour_fwd_kernel on VcrRasterArgs.colour_stream)
// Returns the number of tiles of the Gaussian's 3-sigma rectangle (0 = culled); `emit` receives the number of tile instances
// it will emit: rectangles of up to VCR_RECT_MASK_TILES tiles are tested tile by tile (see vcr_common.h, `tile_touch`) and
// carry the result as a bit mask in their 8-byte rectangle record, larger ones emit every tile.
template <bool STAGE, bool COLOUR, bool QL>
__device__ __forceinline__ uint32_t preprocess_one(const VcrRasterArgs& a, const GeomState& g, int32_t* __restrict__ radii,
                                                   uint32_t* __restrict__ depth_key, uint32_t* __restrict__ ids,
                                                   const float* s_sh, int i, uint32_t& emit, bool& far) {
    emit = 0;
    far = false;
    if (i >= a.N) return 0;
    radii[i] = 0;
    g.tiles[i] = 0;
    g.rect[i] = make_uint2(0u, 0u);
    if (depth_key) depth_key[i] = 0xFFFFFFFFu;        // (NULL: the keys come from depth_key_kernel on the sort stream)

    const Cam cam = load_cam(a);
    const float p[3] = {a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2]};
    Proj pr;
    project(a, cam, p, pr);
    if (pr.t[2] <= VCR_NEAR) return 0;

    const float* P = cam.P;
    const float hx = p[0] * P[0] + p[1] * P[4] + p[2] * P[8] + P[12];
    const float hy = p[0] * P[1] + p[1] * P[5] + p[2] * P[9] + P[13];
    const float hw = p[0] * P[3] + p[1] * P[7] + p[2] * P[11] + P[15];
    const float pw = 1.f / (hw + 1e-7f);

    float S[6], R[9], s[3];
    load_cov3d(a, i, S, R, s);
    jacobian_rows(a, cam, pr);
    float SM0[3], SM1[3];
    sym_mul(S, pr.M0, SM0);
    sym_mul(S, pr.M1, SM1);
    const float ca = pr.M0[0] * SM0[0] + pr.M0[1] * SM0[1] + pr.M0[2] * SM0[2] + VCR_LOWPASS;
    const float cb = pr.M0[0] * SM1[0] + pr.M0[1] * SM1[1] + pr.M0[2] * SM1[2];
    const float cc = pr.M1[0] * SM1[0] + pr.M1[1] * SM1[1] + pr.M1[2] * SM1[2] + VCR_LOWPASS;
    const float det = ca * cc - cb * cb;
    if (det == 0.f) return 0;
    const float idet = 1.f / det;
    const float mid = 0.5f * (ca + cc);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float rad = ceilf(3.f * sqrtf(lam));

    const float px = ((hx * pw + 1.f) * a.W - 1.f) * 0.5f;
    const float py = ((hy * pw + 1.f) * a.H - 1.f) * 0.5f;
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE, gy = (a.H + VCR_TILE - 1) / VCR_TILE;
    const int xmin = min(gx, max(0, (int)floorf((px - rad) / VCR_TILE)));
    const int xmax = min(gx, max(0, (int)floorf((px + rad + VCR_TILE - 1) / VCR_TILE)));
    const int ymin = min(gy, max(0, (int)floorf((py - rad) / VCR_TILE)));
    const int ymax = min(gy, max(0, (int)floorf((py + rad + VCR_TILE - 1) / VCR_TILE)));
    const int ntiles = (xmax - xmin) * (ymax - ymin);
    if (ntiles <= 0) return 0;

    GeomRec rec;
    uint8_t clampbits = 0;
    if (!COLOUR) {
        rec.r = rec.g = rec.b = 0.f;
    } else if (a.colors_precomp) {
        rec.r = a.colors_precomp[3 * (size_t)i]; rec.g = a.colors_precomp[3 * (size_t)i + 1];
        rec.b = a.colors_precomp[3 * (size_t)i + 2];
    } else {
        float dx = p[0] - cam.c[0], dy = p[1] - cam.c[1], dz = p[2] - cam.c[2];
        const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= il; dy *= il; dz *= il;
        const float* sh = STAGE ? (s_sh + threadIdx.x * SH_ROW) : (a.shs + (size_t)i * a.K * 3);
        const int nb = (a.sh_degree + 1) * (a.sh_degree + 1);
        float c0, c1, c2;
        float4 M[3];
        VCR_SH_COLOUR_JAC(a.sh_degree, nb, sh, dx, dy, dz, il, c0, c1, c2, M);
        if (c0 < 0.f) { c0 = 0.f; clampbits |= 1; }
        if (c1 < 0.f) { c1 = 0.f; clampbits |= 2; }
        if (c2 < 0.f) { c2 = 0.f; clampbits |= 4; }
        rec.r = c0; rec.g = c1; rec.b = c2;
        g.cjac[3 * (size_t)i] = M[0]; g.cjac[3 * (size_t)i + 1] = M[1]; g.cjac[3 * (size_t)i + 2] = M[2];
    }
    rec.px = px; rec.py = py; rec.z = pr.t[2]; rec.opacity = a.opacities[i];
    rec.ca = cc * idet; rec.cb = -cb * idet; rec.cc = ca * idet;
    if (a.normals_precomp) {
        rec.nx = a.normals_precomp[3 * (size_t)i]; rec.ny = a.normals_precomp[3 * (size_t)i + 1];
        rec.nz = a.normals_precomp[3 * (size_t)i + 2];
        rec.plane = rec.nx * pr.t[0] + rec.ny * pr.t[1] + rec.nz * pr.t[2];
    } else {
        rec.nx = rec.ny = rec.nz = 0.f; rec.plane = 0.f;
    }
    // the first two semantic features ride in the record's pad slots: the forward compositing of S <= 2 reads them from there
    rec.pad0 = a.S > 0 ? a.semantics_precomp[(size_t)i * a.S] : 0.f;
    rec.pad1 = a.S > 1 ? a.semantics_precomp[(size_t)i * a.S + 1] : 0.f;
    float4* dst = reinterpret_cast<float4*>(g.rec + i);
    const float4* src = reinterpret_cast<const float4*>(&rec);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    for (int k = 0; k < a.S; ++k) g.sem[(size_t)i * a.S + k] = a.semantics_precomp[(size_t)i * a.S + k];
    if (COLOUR) g.clamped[i] = clampbits;
    g.tiles[i] = (uint32_t)ntiles;
    const int rw = xmax - xmin, rh = ymax - ymin;
    if (QL) {      // (compile-time: the per-tile form keeps its registers and its 32-bit masks)
        // quad lists: the same record in units of 8x8 cells (2 gx cells per row); the exact test runs per cell
        // The cells are those of the TILE rectangle (the reference renders a Gaussian at every pixel of its tile rectangle where
        // alpha >= 1/255, also beyond the 3-sigma square), the mask drops the cells it cannot reach.
        const int cx0 = 2 * xmin, cx1 = 2 * xmax, cy0 = 2 * ymin, cy1 = 2 * ymax;
        const int cw = cx1 - cx0, ch = cy1 - cy0;
        if (cw * ch <= 2 * VCR_RECT_MASK_TILES && cw <= 32 && ch <= 32) {       // up to 64 cells = 16 tiles: exact cell mask
            const float4 q0 = make_float4(rec.px, rec.py, rec.z, rec.opacity), q1 = make_float4(rec.ca, rec.cb, rec.cc, rec.plane);
            unsigned long long mask = 0;
            for (int k = 0, cy = cy0; cy < cy1; ++cy)
                for (int cx = cx0; cx < cx1; ++cx, ++k)
                    if (quad_touch(q0, q1, (float)(cx * 8), (float)(cy * 8), 7.f, 7.f)) mask |= 1ull << k;
            emit = (uint32_t)__popcll(mask);
            const uint32_t wide = cw * ch > VCR_RECT_MASK_TILES ? VCR_RECT_MASK64 : 0u;
            g.rect[i] = (cw > 0 && ch > 0) ? make_uint2(VCR_RECT_MASKED | wide | (uint32_t)cx0 | ((uint32_t)cy0 << 10) |
                                                         ((uint32_t)(cw - 1) << 20) | ((uint32_t)(ch - 1) << 25), (uint32_t)mask)
                                           : make_uint2(0u, 0u);
            if (wide) g.rect_hi[i] = (uint32_t)(mask >> 32);
        } else {
            emit = (uint32_t)(cw * ch);
            g.rect[i] = make_uint2((uint32_t)cx0 | ((uint32_t)cy0 << 10), (uint32_t)cw | ((uint32_t)ch << 16));
        }
    } else if (ntiles <= VCR_RECT_MASK_TILES) {
        const float4 q0 = make_float4(rec.px, rec.py, rec.z, rec.opacity), q1 = make_float4(rec.ca, rec.cb, rec.cc, rec.plane);
        uint32_t mask = 0;
        for (int k = 0, ty = ymin; ty < ymax; ++ty)
            for (int tx = xmin; tx < xmax; ++tx, ++k)
                if (tile_touch(q0, q1, tx, ty)) mask |= 1u << k;
        emit = (uint32_t)__popc(mask);
        g.rect[i] = make_uint2(VCR_RECT_MASKED | (uint32_t)xmin | ((uint32_t)ymin << 10) | ((uint32_t)(rw - 1) << 20) |
                               ((uint32_t)(rh - 1) << 25), mask);
    } else {
        emit = (uint32_t)ntiles;
        g.rect[i] = make_uint2((uint32_t)xmin | ((uint32_t)ymin << 10), (uint32_t)rw | ((uint32_t)rh << 16));
    }
    radii[i] = (int32_t)rad;
    if (depth_key) depth_key[i] = vcr_depth_key(pr.t[2]);
    far = (vcr_depth_key(pr.t[2]) >> VCR_DEPTH_KEY_BITS) != 0;
    return (uint32_t)ntiles;
}

// vis_slots (optional; then blk_counts and host too): the library-owned ticket / flag words (vcr_common.h); blk_counts: three words
// per workgroup, [visible Gaussians | tile instances of the 3-sigma rectangles | tile instances emitted]
template <bool STAGE, bool COLOUR, bool QL>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(VcrRasterArgs a, GeomState g, int32_t* __restrict__ radii,
                                                             uint32_t* __restrict__ depth_key, uint32_t* __restrict__ ids,
                                                             uint32_t* __restrict__ vis_slots, uint32_t* __restrict__ blk_counts,
                                                             VcrPublished* host, uint32_t seq) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    if (STAGE) {
        const int base = blockIdx.x * 256;
        stage_sh_in(a, base, min(256, a.N - base), s_sh);
        __syncthreads();
    }
    uint32_t em;
    bool far;
    const uint32_t nt = preprocess_one<STAGE, COLOUR, QL>(a, g, radii, depth_key, ids, s_sh, blockIdx.x * 256 + threadIdx.x, em, far);
    if (vis_slots) {
        __shared__ uint32_t s_cnt[4][4];
        __shared__ bool s_last;
        uint32_t c = nt != 0 ? 1u : 0u, r = nt;
        for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); r += __shfl_xor(r, o); em += __shfl_xor(em, o); }
        const uint32_t anyfar = __builtin_amdgcn_ballot_w64(far) != 0 ? 1u : 0u;       // (a visible depth beyond 27 key bits: rare)
        if ((threadIdx.x & 63) == 0) {
            s_cnt[0][threadIdx.x >> 6] = c; s_cnt[1][threadIdx.x >> 6] = r; s_cnt[2][threadIdx.x >> 6] = em; s_cnt[3][threadIdx.x >> 6] = anyfar;
        }
        __syncthreads();
        // Round 5 (was: three contended slot atomics per workgroup, a memset in front of every call and publish_counts_kernel
        // behind it): thread 0 STORES the workgroup's three counts into the workgroup's own words of `blk_counts` (device-scope
        // stores: write-through), waits for them to complete and draws the done-ticket; the workgroup that draws the last
        // ticket sums all rows, hands the totals to the spinning host and leaves the ticket words zero for the next call.
        // No fence anywhere: a __threadfence() here writes back and invalidates the XCD's whole L2, and 3907 of them made this
        // 38 us kernel take 730 us; the ordering that is needed -- counts before ticket -- is the s_waitcnt between them.
        if (threadIdx.x == 0) {
            uint32_t* row = blk_counts + 3 * (size_t)blockIdx.x;
            __hip_atomic_store(row + 0, s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(row + 1, s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(row + 2, s_cnt[2][0] + s_cnt[2][1] + s_cnt[2][2] + s_cnt[2][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_cnt[3][0] | s_cnt[3][1] | s_cnt[3][2] | s_cnt[3][3]) {
                uint32_t was = atomicOr(vis_slots + VCR_FAR_FLAG_WORD, 1u);
                asm volatile("" : "+v"(was));
            }
            // (This is NOT the HIP memory model's release / acquire: it relies on two gfx9-family facts -- stores are counted in
            //  vmcnt (gfx10+ count them in vscnt), and device-scope stores / loads go through to the memory side of the L2s (sc1).
            //  The library is built for gfx950 only; any other target must use __ATOMIC_RELEASE on the ticket and an acquire
            //  fence in the last workgroup instead.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the count-rows-before-ticket ordering below is written for gfx942 / gfx950 (stores counted in vmcnt)"
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // two-level ticket (vcr_common.h)
            const uint32_t grp = blockIdx.x % VCR_DONE_GROUPS;
            const uint32_t in_grp = (gridDim.x - grp + VCR_DONE_GROUPS - 1) / VCR_DONE_GROUPS;
            const uint32_t groups = gridDim.x < VCR_DONE_GROUPS ? gridDim.x : VCR_DONE_GROUPS;
            bool last = false;
            if (atomicAdd(vis_slots + VCR_DONE_GROUP_WORD + grp, 1u) == in_grp - 1) {
                uint32_t was = atomicExch(vis_slots + VCR_DONE_GROUP_WORD + grp, 0u);    // (reset for the next call, performed ...
                asm volatile("" : "+v"(was));                                            //  ... before the top ticket is drawn)
                last = atomicAdd(vis_slots + VCR_DONE_WORD, 1u) == groups - 1;
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        __shared__ unsigned long long s_r[4], s_e[4];
        __shared__ uint32_t s_v[4];
        unsigned long long rr = 0, ee = 0; uint32_t vv = 0;
        for (uint32_t k = threadIdx.x; k < gridDim.x; k += 256) {
            vv += __hip_atomic_load(blk_counts + 3 * (size_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rr += __hip_atomic_load(blk_counts + 3 * (size_t)k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ee += __hip_atomic_load(blk_counts + 3 * (size_t)k + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int o = 32; o > 0; o >>= 1) { vv += __shfl_xor(vv, o); rr += __shfl_xor(rr, o); ee += __shfl_xor(ee, o); }
        if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = vv; s_r[threadIdx.x >> 6] = rr; s_e[threadIdx.x >> 6] = ee; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t farw = __hip_atomic_load(vis_slots + VCR_FAR_FLAG_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(vis_slots + VCR_FAR_FLAG_WORD, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(vis_slots + VCR_DONE_WORD, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host->R = s_r[0] + s_r[1] + s_r[2] + s_r[3];
            host->E = s_e[0] + s_e[1] + s_e[2] + s_e[3];
            host->V = s_v[0] + s_v[1] + s_v[2] + s_v[3];
            host->far = farw;
            __threadfence_system();
            host->seq = seq;
        }
    }
}

// SH -> RGB of the visible Gaussians (tiles > 0) into the colour slot of their GeomRec + the clamp bits; the second half
// of preprocess_fwd_kernel<.., false>.
// Launched with a small persistent grid (vcr_side_grid()): a kernel of the side stream must leave wave slots and LDS for
// the main stream's sort workgroups (1024 threads + 82 KB LDS each), which a full grid of it starves for its whole duration.
template <bool STAGE>
__global__ void __launch_bounds__(256) colour_fwd_kernel(VcrRasterArgs a, GeomState g) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int nblk = (a.N + 255) / 256;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int i = blk * 256 + threadIdx.x;
        if (STAGE) {
            __syncthreads();                                  // rows of the previous tile are consumed
            const int base = blk * 256;
            stage_sh_in(a, base, min(256, a.N - base), s_sh);
            __syncthreads();
        }
        if (i >= a.N || g.tiles[i] == 0) continue;
        const float* c = a.campos;
        float dx = a.means3D[3 * (size_t)i] - c[0], dy = a.means3D[3 * (size_t)i + 1] - c[1], dz = a.means3D[3 * (size_t)i + 2] - c[2];
        const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= il; dy *= il; dz *= il;
        const float* sh = STAGE ? (s_sh + threadIdx.x * SH_ROW) : (a.shs + (size_t)i * a.K * 3);
        const int nb = (a.sh_degree + 1) * (a.sh_degree + 1);
        float c0, c1, c2;
        float4 M[3];
        VCR_SH_COLOUR_JAC(a.sh_degree, nb, sh, dx, dy, dz, il, c0, c1, c2, M);
        g.cjac[3 * (size_t)i] = M[0]; g.cjac[3 * (size_t)i + 1] = M[1]; g.cjac[3 * (size_t)i + 2] = M[2];
        uint8_t clampbits = 0;
        if (c0 < 0.f) { c0 = 0.f; clampbits |= 1; }
        if (c1 < 0.f) { c1 = 0.f; clampbits |= 2; }
        if (c2 < 0.f) { c2 = 0.f; clampbits |= 4; }
        { float* q2_ = &g.rec[i].r; q2_[0] = c0; q2_[1] = c1; q2_[2] = c2; }      // (pad0 keeps the semantic feature)
        g.clamped[i] = clampbits;
    }
}

// TAIL: the static tail of the training iteration (model_math.h: activation adjoint, l1_scale gradient, densification
// statistics, Adam on xyz / scaling / rotation / opacity, activation for the next camera) follows in the SAME thread, on the
// gradients still in registers: dL/d(means3D, means2D, scales, rotations, opacities, normals) are never written.
struct TailArgs { VcrGeometryStep t; GeomBias gb; };

template <bool STAGE, bool TAIL>
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(VcrRasterArgs a, GeomState g, const int32_t* __restrict__ radii,
                                                             GradRec* __restrict__ sgrad,
                                                             float* __restrict__ sgrad_sem, VcrBackwardIO io, TailArgs ta) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int blk_base = blockIdx.x * 256, blk_cnt = min(256, a.N - blk_base);
    const bool live = i < a.N;
    const size_t i3 = 3 * (size_t)(live ? i : 0);
    const bool vis = live && radii[i] > 0;
    if (!STAGE && !live) return;
    const int nb = (a.sh_degree + 1) * (a.sh_degree + 1);

    // Arithmetic of the adjoint chain below.  profiles/r5_grad_stage_errors.txt separated the stages: the screen-space sums the
    // compositing backward leaves in the GradRec are as accurate as an fp32 evaluation of the oracle (error ratio ~1.0), the
    // PARAMETER gradients were 2-5x worse -- the excess was made HERE: conic -> Sigma2D -> Sigma3D -> (scale, quaternion) and the
    // Jacobian's dependence on the mean are sums of products that cancel for flat / needle-shaped Gaussians.  Round 5 ran the whole
    // chain in fp64 (142 VGPRs, 3 waves per SIMD, +21 us at 1 M Gaussians).  Round 6 ships the MIXED form: fp64 only for what
    // cancels -- Sigma3D, M = J W, Sigma2D, its determinant and the conic adjoint (ga, gb, gc) -- and fp32 behind them (106 VGPRs,
    // 4 waves per SIMD, 129 against 143 us): the same error figures on c1 (profiles/r5_ratio_mixed_precision_experiment.txt) and
    // the whole GPU suite in report mode (profiles/r6_grad_report_mixed_k4.txt, r6_grad_vs_1e-4.txt).
    // -DVCR_BWD_MIXED=0 -DVCR_BWD_REAL=double: round 5's full-fp64 chain; -DVCR_BWD_MIXED=0 -DVCR_BWD_REAL=float: rounds 1-4.
#ifndef VCR_BWD_MIXED
#define VCR_BWD_MIXED 1
#endif
#ifndef VCR_BWD_REAL
#if VCR_BWD_MIXED
#define VCR_BWD_REAL float
#else
#define VCR_BWD_REAL double
#endif
#endif
    typedef VCR_BWD_REAL BR;
    BR dp[3] = {0, 0, 0};                          // dL/dmeans3D
    float dm2[2] = {0.f, 0.f}, dm2a[2] = {0.f, 0.f};
    float dn[3] = {0.f, 0.f, 0.f};
    float dop = 0.f;
    BR dS[6] = {0, 0, 0, 0, 0, 0};                 // dL/dSigma (symmetric, full-matrix entries)
    float dcol[3] = {0.f, 0.f, 0.f};
    BR R[9], s[3], S[6];
    BR dsc[3] = {0, 0, 0};
    BR dq[4] = {0, 0, 0, 0};
    float dsem[VCR_MAX_SEM] = {0.f, 0.f, 0.f, 0.f};

    if (vis) {
        const Cam cam = load_cam(a);
        const float* V = cam.V;
        const float* P = cam.P;
        GradRec gr = sgrad[i];
        // The accumulators are library-owned and zero between calls (round 5: no 64 B x N memset in front of every backward):
        // this kernel is their only reader, so it clears each record behind its own read.  Records of culled Gaussians were
        // never written (they are in no list).
        {
            const float4 z4 = {0.f, 0.f, 0.f, 0.f};
            float4* zr = reinterpret_cast<float4*>(sgrad + i);
            zr[0] = z4; zr[1] = z4; zr[2] = z4; zr[3] = z4;
            for (int k = 0; k < a.S; ++k) { dsem[k] = sgrad_sem[(size_t)i * a.S + k]; sgrad_sem[(size_t)i * a.S + k] = 0.f; }
        }
        gr.finish(a.opacities[i]);
#if VCR_BWD_MIXED
        // fp64 only for what cancels -- Sigma3D, M = J W, Sigma2D, its determinant and the conic adjoint (ga, gb, gc) -- and the
        // chain behind them in fp32 again (BR = float in this build)
        float gaf, gbf, gcf;
        {
            typedef double DD;
            const DD pd[3] = {(DD)a.means3D[i3], (DD)a.means3D[i3 + 1], (DD)a.means3D[i3 + 2]};
            ProjT<DD> prd;
            project<DD>(a, cam, pd, prd);
            DD Sd[6], Rd[9], sd[3];
            load_cov3d<DD>(a, i, Sd, Rd, sd);
            jacobian_rows<DD>(a, cam, prd);
            DD SMd0[3], SMd1[3];
            sym_mul<DD>(Sd, prd.M0, SMd0);
            sym_mul<DD>(Sd, prd.M1, SMd1);
            const DD cad = prd.M0[0] * SMd0[0] + prd.M0[1] * SMd0[1] + prd.M0[2] * SMd0[2] + (DD)VCR_LOWPASS;
            const DD cbd = prd.M0[0] * SMd1[0] + prd.M0[1] * SMd1[1] + prd.M0[2] * SMd1[2];
            const DD ccd = prd.M1[0] * SMd1[0] + prd.M1[1] * SMd1[1] + prd.M1[2] * SMd1[2] + (DD)VCR_LOWPASS;
            const DD detd = cad * ccd - cbd * cbd;
            const DD id2d = 1.0 / (detd * detd);
            const DD gAd = gr.ca, gBd = gr.cb, gCd = gr.cc;
            gaf = (float)((-ccd * ccd * gAd + cbd * ccd * gBd - cbd * cbd * gCd) * id2d);
            gcf = (float)((-cad * cad * gCd + cad * cbd * gBd - cbd * cbd * gAd) * id2d);
            gbf = (float)((2.0 * cbd * ccd * gAd - (cad * ccd + cbd * cbd) * gBd + 2.0 * cad * cbd * gCd) * id2d);
        }
        const BR p[3] = {(BR)a.means3D[i3], (BR)a.means3D[i3 + 1], (BR)a.means3D[i3 + 2]};
        ProjT<BR> pr;
        project<BR>(a, cam, p, pr);
        load_cov3d<BR>(a, i, S, R, s);
        jacobian_rows<BR>(a, cam, pr);
        BR SM0[3], SM1[3];
        sym_mul<BR>(S, pr.M0, SM0);
        sym_mul<BR>(S, pr.M1, SM1);
        const BR ga = gaf, gb = gbf, gc = gcf;
        const BR hb = BR(0.5) * gb;
        // dSigma = M^T G M
        const BR* M0 = pr.M0; const BR* M1 = pr.M1;
        dS[0] = ga * M0[0] * M0[0] + gb * M0[0] * M1[0] + gc * M1[0] * M1[0];
        dS[3] = ga * M0[1] * M0[1] + gb * M0[1] * M1[1] + gc * M1[1] * M1[1];
        dS[5] = ga * M0[2] * M0[2] + gb * M0[2] * M1[2] + gc * M1[2] * M1[2];
        dS[1] = ga * M0[0] * M0[1] + hb * (M0[0] * M1[1] + M1[0] * M0[1]) + gc * M1[0] * M1[1];
        dS[2] = ga * M0[0] * M0[2] + hb * (M0[0] * M1[2] + M1[0] * M0[2]) + gc * M1[0] * M1[2];
        dS[4] = ga * M0[1] * M0[2] + hb * (M0[1] * M1[2] + M1[1] * M0[2]) + gc * M1[1] * M1[2];
        // dM = 2 G M Sigma
        BR dM0[3], dM1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dM0[k] = BR(2) * (ga * SM0[k] + hb * SM1[k]);
            dM1[k] = BR(2) * (hb * SM0[k] + gc * SM1[k]);
        }
#else
        const BR p[3] = {(BR)a.means3D[i3], (BR)a.means3D[i3 + 1], (BR)a.means3D[i3 + 2]};
        ProjT<BR> pr;
        project<BR>(a, cam, p, pr);
        load_cov3d<BR>(a, i, S, R, s);
        jacobian_rows<BR>(a, cam, pr);
        BR SM0[3], SM1[3];
        sym_mul<BR>(S, pr.M0, SM0);
        sym_mul<BR>(S, pr.M1, SM1);
        const BR ca = pr.M0[0] * SM0[0] + pr.M0[1] * SM0[1] + pr.M0[2] * SM0[2] + (BR)VCR_LOWPASS;
        const BR cb = pr.M0[0] * SM1[0] + pr.M0[1] * SM1[1] + pr.M0[2] * SM1[2];
        const BR cc = pr.M1[0] * SM1[0] + pr.M1[1] * SM1[1] + pr.M1[2] * SM1[2] + (BR)VCR_LOWPASS;
        const BR det = ca * cc - cb * cb;
        const BR id2 = BR(1) / (det * det);
        // conic (A,B,C) = (c,-b,a)/det  ->  cov2D (a,b,c)
        const BR gA = gr.ca, gB = gr.cb, gC = gr.cc;
        const BR ga = (-cc * cc * gA + cb * cc * gB - cb * cb * gC) * id2;
        const BR gc = (-ca * ca * gC + ca * cb * gB - cb * cb * gA) * id2;
        const BR gb = (BR(2) * cb * cc * gA - (ca * cc + cb * cb) * gB + BR(2) * ca * cb * gC) * id2;
        const BR hb = BR(0.5) * gb;
        // dSigma = M^T G M
        const BR* M0 = pr.M0; const BR* M1 = pr.M1;
        dS[0] = ga * M0[0] * M0[0] + gb * M0[0] * M1[0] + gc * M1[0] * M1[0];
        dS[3] = ga * M0[1] * M0[1] + gb * M0[1] * M1[1] + gc * M1[1] * M1[1];
        dS[5] = ga * M0[2] * M0[2] + gb * M0[2] * M1[2] + gc * M1[2] * M1[2];
        dS[1] = ga * M0[0] * M0[1] + hb * (M0[0] * M1[1] + M1[0] * M0[1]) + gc * M1[0] * M1[1];
        dS[2] = ga * M0[0] * M0[2] + hb * (M0[0] * M1[2] + M1[0] * M0[2]) + gc * M1[0] * M1[2];
        dS[4] = ga * M0[1] * M0[2] + hb * (M0[1] * M1[2] + M1[1] * M0[2]) + gc * M1[1] * M1[2];
        // dM = 2 G M Sigma
        BR dM0[3], dM1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dM0[k] = BR(2) * (ga * SM0[k] + hb * SM1[k]);
            dM1[k] = BR(2) * (hb * SM0[k] + gc * SM1[k]);
        }
#endif
        // dJ = dM Rv^T, Rv[c][k] = V[k*4+c]
        const BR dJ00 = dM0[0] * (BR)V[0] + dM0[1] * (BR)V[4] + dM0[2] * (BR)V[8];
        const BR dJ02 = dM0[0] * (BR)V[2] + dM0[1] * (BR)V[6] + dM0[2] * (BR)V[10];
        const BR dJ11 = dM1[0] * (BR)V[1] + dM1[1] * (BR)V[5] + dM1[2] * (BR)V[9];
        const BR dJ12 = dM1[0] * (BR)V[2] + dM1[1] * (BR)V[6] + dM1[2] * (BR)V[10];
        const BR tz = pr.t[2], itz = BR(1) / tz, itz2 = itz * itz;
        BR dt[3] = {0, 0, 0};
        dt[2] = (-dJ00 * pr.fx + dJ02 * pr.fx * pr.u - dJ11 * pr.fy + dJ12 * pr.fy * pr.v) * itz2;
        const BR du = -dJ02 * pr.fx * itz, dv = -dJ12 * pr.fy * itz;
        if (!pr.uc) { dt[0] += du * itz; dt[2] -= du * pr.t[0] * itz2; }
        if (!pr.vc) { dt[1] += dv * itz; dt[2] -= dv * pr.t[1] * itz2; }
        // depth and plane offset
        dt[2] += (BR)gr.z;
        if (a.normals_precomp) {
            const BR n[3] = {(BR)a.normals_precomp[i3], (BR)a.normals_precomp[i3 + 1], (BR)a.normals_precomp[i3 + 2]};
            dt[0] += n[0] * (BR)gr.plane; dt[1] += n[1] * (BR)gr.plane; dt[2] += n[2] * (BR)gr.plane;
            BR dnr[3] = {(BR)gr.nx + pr.t[0] * (BR)gr.plane, (BR)gr.ny + pr.t[1] * (BR)gr.plane, (BR)gr.nz + pr.t[2] * (BR)gr.plane};
            if (!TAIL && io.normals_Rw2c) {        // data parallel: as the gradient w.r.t. the world-space axis column (vcr_raster.h)
                const float* Rw = io.normals_Rw2c;
                const BR sg = (io.normals_aux[i] & 4) ? BR(-1) : BR(1);
                const BR w0 = sg * ((BR)Rw[0] * dnr[0] + (BR)Rw[3] * dnr[1] + (BR)Rw[6] * dnr[2]);
                const BR w1 = sg * ((BR)Rw[1] * dnr[0] + (BR)Rw[4] * dnr[1] + (BR)Rw[7] * dnr[2]);
                const BR w2 = sg * ((BR)Rw[2] * dnr[0] + (BR)Rw[5] * dnr[1] + (BR)Rw[8] * dnr[2]);
                dnr[0] = w0; dnr[1] = w1; dnr[2] = w2;
            }
            dn[0] = (float)dnr[0]; dn[1] = (float)dnr[1]; dn[2] = (float)dnr[2];
        }
        // t = p Rv^T + tv  ->  dp_k += sum_c dt_c V[k*4+c]
#pragma unroll
        for (int k = 0; k < 3; ++k) dp[k] += dt[0] * (BR)V[k * 4 + 0] + dt[1] * (BR)V[k * 4 + 1] + dt[2] * (BR)V[k * 4 + 2];
        // pixel position through the full projection
        const BR hx = p[0] * (BR)P[0] + p[1] * (BR)P[4] + p[2] * (BR)P[8] + (BR)P[12];
        const BR hy = p[0] * (BR)P[1] + p[1] * (BR)P[5] + p[2] * (BR)P[9] + (BR)P[13];
        const BR hw = p[0] * (BR)P[3] + p[1] * (BR)P[7] + p[2] * (BR)P[11] + (BR)P[15];
        const BR pw = BR(1) / (hw + (BR)1e-7f);
        dm2[0] = gr.gx * 0.5f * a.W; dm2[1] = gr.gy * 0.5f * a.H;
        dm2a[0] = gr.agx * 0.5f * a.W; dm2a[1] = gr.agy * 0.5f * a.H;
        const BR g2x = (BR)gr.gx * BR(0.5) * (BR)a.W, g2y = (BR)gr.gy * BR(0.5) * (BR)a.H;
        const BR dhx = g2x * pw, dhy = g2y * pw;
        const BR dhw = -(hx * g2x + hy * g2y) * pw * pw;
#pragma unroll
        for (int k = 0; k < 3; ++k) dp[k] += (BR)P[k * 4 + 0] * dhx + (BR)P[k * 4 + 1] * dhy + (BR)P[k * 4 + 3] * dhw;
        dop = gr.opacity;
        // colour
        dcol[0] = gr.r; dcol[1] = gr.g; dcol[2] = gr.b;
        if (!a.colors_precomp) {
            const uint8_t cl = g.clamped[i];
            if (cl & 1) dcol[0] = 0.f;
            if (cl & 2) dcol[1] = 0.f;
            if (cl & 4) dcol[2] = 0.f;
            const float pf[3] = {a.means3D[i3], a.means3D[i3 + 1], a.means3D[i3 + 2]};
            float dx = pf[0] - cam.c[0], dy = pf[1] - cam.c[1], dz = pf[2] - cam.c[2];
            const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            dx *= il; dy *= il; dz *= il;
            // colour -> mean through the view direction: M was stored with the colour (GeomState::cjac), so the 192 B of SH
            // coefficients are not read here; the SH gradient itself (only when asked for) is basis (x) dL/drgb
            float* dsh = STAGE ? (s_sh + threadIdx.x * SH_ROW) : (io.dL_dshs ? io.dL_dshs + (size_t)i * a.K * 3 : nullptr);
            if (dsh) {
                float b[16];
                sh_basis<false>(a.sh_degree, dx, dy, dz, b, nullptr, nullptr, nullptr);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < nb) { dsh[3 * k] = b[k] * dcol[0]; dsh[3 * k + 1] = b[k] * dcol[1]; dsh[3 * k + 2] = b[k] * dcol[2]; }
                }
                for (int k = nb; k < a.K; ++k) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
            }
            if (io.view_dirs) { io.view_dirs[i3] = dx; io.view_dirs[i3 + 1] = dy; io.view_dirs[i3 + 2] = dz; }
            const float4 m0 = g.cjac[3 * (size_t)i], m1 = g.cjac[3 * (size_t)i + 1], m2 = g.cjac[3 * (size_t)i + 2];
            dp[0] += (BR)dcol[0] * (BR)m0.x + (BR)dcol[1] * (BR)m1.x + (BR)dcol[2] * (BR)m2.x;
            dp[1] += (BR)dcol[0] * (BR)m0.y + (BR)dcol[1] * (BR)m1.y + (BR)dcol[2] * (BR)m2.y;
            dp[2] += (BR)dcol[0] * (BR)m0.z + (BR)dcol[1] * (BR)m1.z + (BR)dcol[2] * (BR)m2.z;
        }
        // Sigma = (R s)(R s)^T
        if (!a.cov3D_precomp) {
            BR dR[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const BR d0 = (r == 0 ? dS[0] : (r == 1 ? dS[1] : dS[2]));
                const BR d1 = (r == 0 ? dS[1] : (r == 1 ? dS[3] : dS[4]));
                const BR d2 = (r == 0 ? dS[2] : (r == 1 ? dS[4] : dS[5]));
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // dL_rk = 2 * sum_j dSigma[r][j] L[j][k], L[j][k] = R[j][k] s_k
                    const BR dL = BR(2) * (d0 * R[0 * 3 + k] + d1 * R[1 * 3 + k] + d2 * R[2 * 3 + k]) * s[k];
                    dsc[k] += dL * R[r * 3 + k];
                    dR[r * 3 + k] = dL * s[k];
                }
            }
            const float4 qf = reinterpret_cast<const float4*>(a.rotations)[i];
            const BR r = qf.x, x = qf.y, y = qf.z, z = qf.w;
            dq[0] = BR(2) * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            dq[1] = BR(2) * (y * dR[1] + z * dR[2] + y * dR[3] - BR(2) * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - BR(2) * x * dR[8]);
            dq[2] = BR(2) * (-BR(2) * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - BR(2) * y * dR[8]);
            dq[3] = BR(2) * (-BR(2) * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - BR(2) * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
            dsc[0] *= (BR)a.scale_modifier; dsc[1] *= (BR)a.scale_modifier; dsc[2] *= (BR)a.scale_modifier;
        }
    } else if (io.dL_dshs && (STAGE || live)) {
        float* dsh = STAGE ? (s_sh + threadIdx.x * SH_ROW) : (io.dL_dshs + (size_t)i * a.K * 3);
        for (int k = 0; k < a.K * 3; ++k) dsh[k] = 0.f;
    }
    if (STAGE && io.dL_dshs) {                 // coalesced write-back of the block's SH gradients
        __syncthreads();
        if (io.dL_dshs_rest) {
            coop_copy_out(io.dL_dshs + (size_t)blk_base * 3, blk_cnt * 3, 3, 0, s_sh);
            coop_copy_out(io.dL_dshs_rest + (size_t)blk_base * 45, blk_cnt * 45, 45, 3, s_sh);
        } else {
            coop_copy_out(io.dL_dshs + (size_t)blk_base * 48, blk_cnt * 48, 48, 0, s_sh);
        }
    }
    if (!live) return;

    if (TAIL) {
        if (io.dL_dcolors) { io.dL_dcolors[i3] = dcol[0]; io.dL_dcolors[i3 + 1] = dcol[1]; io.dL_dcolors[i3 + 2] = dcol[2]; }
        if (io.dL_drgb) { io.dL_drgb[i3] = dcol[0]; io.dL_drgb[i3 + 1] = dcol[1]; io.dL_drgb[i3 + 2] = dcol[2]; }
        if (io.dL_dsemantics)
            for (int k = 0; k < a.S; ++k) io.dL_dsemantics[(size_t)i * a.S + k] = dsem[k];
        TailGrads tg;
        tg.dp[0] = (float)dp[0]; tg.dp[1] = (float)dp[1]; tg.dp[2] = (float)dp[2];
        tg.ds[0] = (float)dsc[0]; tg.ds[1] = (float)dsc[1]; tg.ds[2] = (float)dsc[2];
        tg.dn[0] = dn[0]; tg.dn[1] = dn[1]; tg.dn[2] = dn[2];
        const bool dens = io.dL_dmeans2D_densify != nullptr;         // (which screen gradient feeds the statistics: a flag here)
        tg.dm2[0] = dens ? dm2a[0] : dm2[0]; tg.dm2[1] = dens ? dm2a[1] : dm2[1];
        tg.dq = make_float4((float)dq[0], (float)dq[1], (float)dq[2], (float)dq[3]);
        tg.dop = dop;
        tg.radius = radii[i];
        tg.has_n = a.normals_precomp != nullptr;
        geometry_step_one<true>(ta.t, ta.gb, i, tg);
        return;
    }
    io.dL_dmeans3D[i3] = (float)dp[0]; io.dL_dmeans3D[i3 + 1] = (float)dp[1]; io.dL_dmeans3D[i3 + 2] = (float)dp[2];
    io.dL_dmeans2D[i3] = dm2[0]; io.dL_dmeans2D[i3 + 1] = dm2[1]; io.dL_dmeans2D[i3 + 2] = 0.f;
    if (io.dL_dmeans2D_densify) {
        io.dL_dmeans2D_densify[i3] = dm2a[0]; io.dL_dmeans2D_densify[i3 + 1] = dm2a[1];
        io.dL_dmeans2D_densify[i3 + 2] = 0.f;
    }
    io.dL_dopacities[i] = dop;
    if (io.dL_dcolors) { io.dL_dcolors[i3] = dcol[0]; io.dL_dcolors[i3 + 1] = dcol[1]; io.dL_dcolors[i3 + 2] = dcol[2]; }
    if (io.dL_drgb) { io.dL_drgb[i3] = dcol[0]; io.dL_drgb[i3 + 1] = dcol[1]; io.dL_drgb[i3 + 2] = dcol[2]; }
    if (io.dL_dnormals) { io.dL_dnormals[i3] = dn[0]; io.dL_dnormals[i3 + 1] = dn[1]; io.dL_dnormals[i3 + 2] = dn[2]; }
    if (io.dL_dsemantics)
        for (int k = 0; k < a.S; ++k) io.dL_dsemantics[(size_t)i * a.S + k] = dsem[k];
    if (io.dL_dscales) {
        io.dL_dscales[i3] = (float)dsc[0]; io.dL_dscales[i3 + 1] = (float)dsc[1]; io.dL_dscales[i3 + 2] = (float)dsc[2];
        reinterpret_cast<float4*>(io.dL_drotations)[i] = make_float4((float)dq[0], (float)dq[1], (float)dq[2], (float)dq[3]);
    }
    if (io.dL_dcov3D) {
        float* d = io.dL_dcov3D + 6 * (size_t)i;
        d[0] = (float)dS[0]; d[1] = (float)(BR(2) * dS[1]); d[2] = (float)(BR(2) * dS[2]); d[3] = (float)dS[3];
        d[4] = (float)(BR(2) * dS[4]); d[5] = (float)dS[5];
    }
}

// ---- factorised SH-gradient exchange (data parallel) ----------------------------------------------------------------
// For one view the SH gradient of a Gaussian is the rank-1 product basis_k(dir_view) x dL/drgb (clamp already applied),
// so ranks exchange only dL/drgb [N,3] (all-gather, 12 B/Gaussian/view) and rebuild sum_views basis_k * dL/drgb_view
// locally instead of all-reducing 192 B/Gaussian of SH gradients.
__global__ void __launch_bounds__(256) sh_grad_from_rgb_kernel(int N, int deg, int nviews, const float* __restrict__ xyz,
                                                               const float* __restrict__ campos,
                                                               const float* __restrict__ drgb, float* __restrict__ d_dc,
                                                               float* __restrict__ d_rest) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int blk_base = blockIdx.x * 256, blk_cnt = min(256, N - blk_base);
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.f;
    if (i < N) {
        const float p0 = xyz[3 * (size_t)i], p1 = xyz[3 * (size_t)i + 1], p2 = xyz[3 * (size_t)i + 2];
        for (int v = 0; v < nviews; ++v) {
            const float* g = drgb + ((size_t)v * N + i) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
            float dx = p0 - campos[3 * v], dy = p1 - campos[3 * v + 1], dz = p2 - campos[3 * v + 2];
            const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            float b[16];
            sh_basis<false>(deg, dx * il, dy * il, dz * il, b, nullptr, nullptr, nullptr);
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc[3 * k] += b[k] * g0; acc[3 * k + 1] += b[k] * g1; acc[3 * k + 2] += b[k] * g2; }
        }
    }
    float* row = s_sh + threadIdx.x * SH_ROW;
#pragma unroll
    for (int k = 0; k < 48; ++k) row[k] = acc[k];
    __syncthreads();
    coop_copy_out(d_dc + (size_t)blk_base * 3, blk_cnt * 3, 3, 0, s_sh);
    coop_copy_out(d_rest + (size_t)blk_base * 45, blk_cnt * 45, 45, 3, s_sh);
}


// ---- SH Adam with the gradient formed on the fly (single-view training) ------------------------------------------------
// grad[g][k][c] = basis_k(view_dirs[g]) * drgb[g][c]; every lane builds its Gaussian's 48-value row in LDS, then the
// block streams params / moments with coalesced 16-byte accesses (same row mapping as coop_copy_*) and applies Adam.
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, const float v[4]) {
    __builtin_nontemporal_store(vf4{v[0], v[1], v[2], v[3]}, reinterpret_cast<vf4*>(p));
}

// KEEP: the updated parameter replaces the gradient in the LDS row (for the fused colour evaluation)
// NT: streaming (non-temporal) loads / stores of p, m, v
template <bool KEEP = false, bool NT = false>
__device__ __forceinline__ void coop_adam(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int total, int per,
                                          int off, float* s, float lr_bc1, float b1, float b2, float eps, float bc2_sqrt) {
    const int n4 = total >> 2;                              // rows start 16-byte aligned: 256*3 and 256*45 floats per block
    float4* p4 = reinterpret_cast<float4*>(p); float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    // software-pipelined by hand: all 12 loads of four 16-byte groups are issued before the first use, so that a small
    // persistent grid (few waves per CU, see colour_fwd_kernel) still keeps enough bytes in flight for HBM
    for (int e0 = threadIdx.x; e0 < n4; e0 += 4 * 256) {
        float4 pp[4], mm[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e4 = e0 + u * 256;
            if (e4 < n4) {
                if (NT) { pp[u] = nt_load4(p4 + e4); mm[u] = nt_load4(m4 + e4); vv[u] = nt_load4(v4 + e4); }
                else { pp[u] = p4[e4]; mm[u] = m4[e4]; vv[u] = v4[e4]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e4 = e0 + u * 256;
            if (e4 >= n4) continue;
            float P[4] = {pp[u].x, pp[u].y, pp[u].z, pp[u].w}, M[4] = {mm[u].x, mm[u].y, mm[u].z, mm[u].w};
            float V[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e4 * 4 + j, g = e / per;
                float* slot = s + g * SH_ROW + off + (e - g * per);
                const float gr = *slot;
                M[j] = b1 * M[j] + (1.f - b1) * gr;
                V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
                P[j] -= lr_bc1 * (M[j] / (sqrtf(V[j]) / bc2_sqrt + eps));
                if (KEEP) *slot = P[j];
            }
            if (NT) {
                nt_store4(p4 + e4, P); nt_store4(m4 + e4, M); nt_store4(v4 + e4, V);
            } else {
                p4[e4] = make_float4(P[0], P[1], P[2], P[3]); m4[e4] = make_float4(M[0], M[1], M[2], M[3]);
                v4[e4] = make_float4(V[0], V[1], V[2], V[3]);
            }
        }
    }
    for (int e = (n4 << 2) + threadIdx.x; e < total; e += 256) {
        const int g = e / per;
        float* slot = s + g * SH_ROW + off + (e - g * per);
        const float gr = *slot;
        const float M = b1 * m[e] + (1.f - b1) * gr, V = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = M; v[e] = V;
        const float P = p[e] - lr_bc1 * (M / (sqrtf(V) / bc2_sqrt + eps));
        p[e] = P;
        if (KEEP) *slot = P;
    }
}

__global__ void __launch_bounds__(256) sh_adam_from_rgb_kernel(int N, int deg, const float* __restrict__ dirs,
                                                               const float* __restrict__ drgb, float* p_dc, float* p_rest,
                                                               float* m_dc, float* v_dc, float* m_rest, float* v_rest,
                                                               float lr_dc_bc1, float lr_rest_bc1, float b1, float b2, float eps,
                                                               float bc2_sqrt, float gscale) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int nblk = (N + 255) / 256;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {      // small persistent grid, see colour_fwd_kernel
        const int i = blk * 256 + threadIdx.x;
        const int blk_base = blk * 256, blk_cnt = min(256, N - blk_base);
        float* row = s_sh + threadIdx.x * SH_ROW;
        __syncthreads();
        if (i < N) {
            const float g0 = drgb[3 * (size_t)i] * gscale, g1 = drgb[3 * (size_t)i + 1] * gscale, g2 = drgb[3 * (size_t)i + 2] * gscale;
            float b[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = 0.f;
            if (g0 != 0.f || g1 != 0.f || g2 != 0.f)
                sh_basis<false>(deg, dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2], b, nullptr, nullptr, nullptr);
            const int nb = (deg + 1) * (deg + 1);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float bk = k < nb ? b[k] : 0.f;
                row[3 * k] = bk * g0; row[3 * k + 1] = bk * g1; row[3 * k + 2] = bk * g2;
            }
        }
        __syncthreads();
        coop_adam(p_dc + (size_t)blk_base * 3, m_dc + (size_t)blk_base * 3, v_dc + (size_t)blk_base * 3, blk_cnt * 3, 3, 0, s_sh,
                  lr_dc_bc1, b1, b2, eps, bc2_sqrt);
        coop_adam(p_rest + (size_t)blk_base * 45, m_rest + (size_t)blk_base * 45, v_rest + (size_t)blk_base * 45, blk_cnt * 45, 45,
                  3, s_sh, lr_rest_bc1, b1, b2, eps, bc2_sqrt);
    }
}

// Data-parallel form: gradient = sum over the step's views of basis_k(normalize(xyz - campos_v)) x dL/drgb_v (drgb_all
// [nviews,N,3] from the all-gather), formed on the fly like sh_grad_from_rgb_kernel, then the same Adam as above.
__global__ void __launch_bounds__(256) sh_adam_from_views_kernel(int N, int deg, int nviews, const float* __restrict__ xyz,
                                                                 const float* __restrict__ campos,
                                                                 const float* __restrict__ drgb, float* p_dc, float* p_rest,
                                                                 float* m_dc, float* v_dc, float* m_rest, float* v_rest,
                                                                 float lr_dc_bc1, float lr_rest_bc1, float b1, float b2, float eps,
                                                                 float bc2_sqrt, float gscale) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int nblk = (N + 255) / 256;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int i = blk * 256 + threadIdx.x;
        const int blk_base = blk * 256, blk_cnt = min(256, N - blk_base);
        float* row = s_sh + threadIdx.x * SH_ROW;
        __syncthreads();
        if (i < N) {
            float acc[48];
#pragma unroll
            for (int k = 0; k < 48; ++k) acc[k] = 0.f;
            const float p0 = xyz[3 * (size_t)i], p1 = xyz[3 * (size_t)i + 1], p2 = xyz[3 * (size_t)i + 2];
            for (int v = 0; v < nviews; ++v) {
                const float* g = drgb + ((size_t)v * N + i) * 3;
                const float g0 = g[0] * gscale, g1 = g[1] * gscale, g2 = g[2] * gscale;
                if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
                float dx = p0 - campos[3 * v], dy = p1 - campos[3 * v + 1], dz = p2 - campos[3 * v + 2];
                const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
                float b[16];
                sh_basis<false>(deg, dx * il, dy * il, dz * il, b, nullptr, nullptr, nullptr);
                const int nb = (deg + 1) * (deg + 1);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float bk = k < nb ? b[k] : 0.f;
                    acc[3 * k] += bk * g0; acc[3 * k + 1] += bk * g1; acc[3 * k + 2] += bk * g2;
                }
            }
#pragma unroll
            for (int k = 0; k < 48; ++k) row[k] = acc[k];
        }
        __syncthreads();
        coop_adam(p_dc + (size_t)blk_base * 3, m_dc + (size_t)blk_base * 3, v_dc + (size_t)blk_base * 3, blk_cnt * 3, 3, 0, s_sh,
                  lr_dc_bc1, b1, b2, eps, bc2_sqrt);
        coop_adam(p_rest + (size_t)blk_base * 45, m_rest + (size_t)blk_base * 45, v_rest + (size_t)blk_base * 45, blk_cnt * 45, 45,
                  3, s_sh, lr_rest_bc1, b1, b2, eps, bc2_sqrt);
    }
}

// SH update (VcrShUpdate) + SH -> RGB of the same forward call in one pass over the coefficients: the Adam step leaves the
// updated coefficients in the LDS rows, from which every lane evaluates its Gaussian's colour for THIS call's camera.
__global__ void __launch_bounds__(256) sh_update_colour_kernel(VcrRasterArgs a, GeomState g, VcrShUpdate u, float lr_dc_bc1,
                                                               float lr_rest_bc1, float bc2_sqrt) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int N = a.N, nblk = (N + 255) / 256;
    float* p_dc = const_cast<float*>(a.shs);
    float* p_rest = const_cast<float*>(a.shs_rest);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int i = blk * 256 + threadIdx.x;
        const int blk_base = blk * 256, blk_cnt = min(256, N - blk_base);
        float* row = s_sh + threadIdx.x * SH_ROW;
        __syncthreads();
        if (i < N) {
            float acc[48];
#pragma unroll
            for (int k = 0; k < 48; ++k) acc[k] = 0.f;
            const int nbu = (u.sh_degree + 1) * (u.sh_degree + 1);
            const int nv = u.nviews > 0 ? u.nviews : 1;
            for (int v = 0; v < nv; ++v) {
                const float* gp = u.drgb + ((size_t)v * N + i) * 3;
                const float g0 = gp[0] * u.grad_scale, g1 = gp[1] * u.grad_scale, g2 = gp[2] * u.grad_scale;
                if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
                float dx, dy, dz;
                if (u.nviews > 0) {
                    dx = u.xyz[3 * (size_t)i] - u.campos_all[3 * v]; dy = u.xyz[3 * (size_t)i + 1] - u.campos_all[3 * v + 1];
                    dz = u.xyz[3 * (size_t)i + 2] - u.campos_all[3 * v + 2];
                    const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
                    dx *= il; dy *= il; dz *= il;
                } else {
                    dx = u.view_dirs[3 * (size_t)i]; dy = u.view_dirs[3 * (size_t)i + 1]; dz = u.view_dirs[3 * (size_t)i + 2];
                }
                float b[16];
                sh_basis<false>(u.sh_degree, dx, dy, dz, b, nullptr, nullptr, nullptr);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float bk = k < nbu ? b[k] : 0.f;
                    acc[3 * k] += bk * g0; acc[3 * k + 1] += bk * g1; acc[3 * k + 2] += bk * g2;
                }
            }
#pragma unroll
            for (int k = 0; k < 48; ++k) row[k] = acc[k];
        }
        __syncthreads();
        coop_adam<true, VCR_SIDE_NT>(p_dc + (size_t)blk_base * 3, u.m_dc + (size_t)blk_base * 3, u.v_dc + (size_t)blk_base * 3, blk_cnt * 3, 3, 0,
                        s_sh, lr_dc_bc1, u.beta1, u.beta2, u.eps, bc2_sqrt);
        coop_adam<true, VCR_SIDE_NT>(p_rest + (size_t)blk_base * 45, u.m_rest + (size_t)blk_base * 45, u.v_rest + (size_t)blk_base * 45,
                        blk_cnt * 45, 45, 3, s_sh, lr_rest_bc1, u.beta1, u.beta2, u.eps, bc2_sqrt);
        __syncthreads();
        if (i < N && g.tiles[i] != 0) {            // colour of the visible Gaussians from the freshly updated rows
            const float* c = a.campos;
            float dx = a.means3D[3 * (size_t)i] - c[0], dy = a.means3D[3 * (size_t)i + 1] - c[1], dz = a.means3D[3 * (size_t)i + 2] - c[2];
            const float il = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            dx *= il; dy *= il; dz *= il;
            const int nb = (a.sh_degree + 1) * (a.sh_degree + 1);
            float c0, c1, c2;
            float4 M[3];
            VCR_SH_COLOUR_JAC(a.sh_degree, nb, row, dx, dy, dz, il, c0, c1, c2, M);
            g.cjac[3 * (size_t)i] = M[0]; g.cjac[3 * (size_t)i + 1] = M[1]; g.cjac[3 * (size_t)i + 2] = M[2];
            uint8_t clampbits = 0;
            if (c0 < 0.f) { c0 = 0.f; clampbits |= 1; }
            if (c1 < 0.f) { c1 = 0.f; clampbits |= 2; }
            if (c2 < 0.f) { c2 = 0.f; clampbits |= 4; }
            { float* q2_ = &g.rec[i].r; q2_[0] = c0; q2_[1] = c1; q2_[2] = c2; }      // (pad0 keeps the semantic feature)
            g.clamped[i] = clampbits;
        }
    }
}

}  // namespace

extern "C" int vcr_sh_adam_from_rgb(int N, int sh_degree, const float* view_dirs, const float* drgb, float* features_dc,
                                    float* features_rest, float* m_dc, float* v_dc, float* m_rest, float* v_rest, float lr_dc,
                                    float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale,
                                    void* stream) {
    if (N <= 0) return 0;
    if (sh_degree < 0 || sh_degree > 3 || step < 1) { vcr_set_error("vcr_sh_adam_from_rgb: bad degree/step"); return 1; }
    if ((((uintptr_t)features_dc) | ((uintptr_t)features_rest) | ((uintptr_t)m_dc) | ((uintptr_t)v_dc) | ((uintptr_t)m_rest) |
         ((uintptr_t)v_rest)) & 15) { vcr_set_error("vcr_sh_adam_from_rgb: tensors must be 16-byte aligned"); return 1; }
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int nblk = (N + 255) / 256, grid = nblk < vcr_side_grid(N) ? nblk : vcr_side_grid(N);
    hipLaunchKernelGGL(sh_adam_from_rgb_kernel, dim3(grid), dim3(256), 256 * SH_ROW * sizeof(float), (hipStream_t)stream,
                       N, sh_degree, view_dirs, drgb, features_dc, features_rest, m_dc, v_dc, m_rest, v_rest,
                       (float)(lr_dc / bc1), (float)(lr_rest / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int vcr_sh_adam_from_rgb_views(int N, int sh_degree, int nviews, const float* xyz, const float* campos_all,
                                          const float* drgb_all, float* features_dc, float* features_rest, float* m_dc,
                                          float* v_dc, float* m_rest, float* v_rest, float lr_dc, float lr_rest, float beta1,
                                          float beta2, float eps, int step, float grad_scale, void* stream) {
    if (N <= 0) return 0;
    if (sh_degree < 0 || sh_degree > 3 || step < 1 || nviews <= 0) { vcr_set_error("vcr_sh_adam_from_rgb_views: bad degree/step/views"); return 1; }
    if ((((uintptr_t)features_dc) | ((uintptr_t)features_rest) | ((uintptr_t)m_dc) | ((uintptr_t)v_dc) | ((uintptr_t)m_rest) |
         ((uintptr_t)v_rest)) & 15) { vcr_set_error("vcr_sh_adam_from_rgb_views: tensors must be 16-byte aligned"); return 1; }
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int nblk = (N + 255) / 256, grid = nblk < vcr_side_grid(N) ? nblk : vcr_side_grid(N);
    hipLaunchKernelGGL(sh_adam_from_views_kernel, dim3(grid), dim3(256), 256 * SH_ROW * sizeof(float), (hipStream_t)stream, N,
                       sh_degree, nviews, xyz, campos_all, drgb_all, features_dc, features_rest, m_dc, v_dc, m_rest, v_rest,
                       (float)(lr_dc / bc1), (float)(lr_rest / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}


extern "C" int vcr_sh_grad_from_rgb(int N, int sh_degree, int nviews, const float* xyz, const float* campos_all,
                                    const float* drgb_all, float* d_features_dc, float* d_features_rest, void* stream) {
    if (N <= 0) return 0;
    if (sh_degree < 0 || sh_degree > 3 || nviews <= 0) { vcr_set_error("vcr_sh_grad_from_rgb: bad degree/views"); return 1; }
    hipLaunchKernelGGL(sh_grad_from_rgb_kernel, dim3((N + 255) / 256), dim3(256), 256 * SH_ROW * sizeof(float),
                       (hipStream_t)stream, N, sh_degree, nviews, xyz, campos_all, drgb_all, d_features_dc, d_features_rest);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// View depth of every Gaussian as a radix-sortable key, without the rest of the projection: lets the depth sort start on
// its own stream before (and beside) preprocess_fwd.  Same project() as preprocess_one, so the keys are bit-identical;
// Gaussians that the projection culls for other reasons keep a real key here, which is harmless (they emit no instances).
__global__ void __launch_bounds__(256) depth_key_kernel(VcrRasterArgs a, uint32_t* __restrict__ depth_key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const Cam cam = load_cam(a);
    const float p[3] = {a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2]};
    Proj pr;
    project(a, cam, p, pr);
    depth_key[i] = pr.t[2] > VCR_NEAR ? vcr_depth_key(pr.t[2]) : 0xFFFFFFFFu;
}

int vcr_launch_depth_keys(const VcrRasterArgs& a, uint32_t* depth_key, hipStream_t st) {
    if (a.N == 0) return 0;
    hipLaunchKernelGGL(depth_key_kernel, dim3((a.N + 255) / 256), dim3(256), 0, st, a, depth_key);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

int vcr_launch_preprocess(const VcrRasterArgs& a, GeomState g, int32_t* radii, uint32_t* depth_key, uint32_t* ids,
                          uint32_t* vis_slots, bool colour, hipStream_t st, uint32_t* blk_counts, VcrPublished* host, uint32_t seq) {
    if (vis_slots && (!blk_counts || !host)) { vcr_set_error("vcr_launch_preprocess: counters without rows / host record"); return 1; }
    if (a.N == 0) return 0;
    const int blocks = (a.N + 255) / 256;
#define VCR_PRE(STAGE, COLOUR, SMEM)                                                                                          \
    do {                                                                                                                       \
        if (a.quad_lists)                                                                                                      \
            hipLaunchKernelGGL((preprocess_fwd_kernel<STAGE, COLOUR, true>), dim3(blocks), dim3(256), SMEM, st, a, g, radii,   \
                               depth_key, ids, vis_slots, blk_counts, host, seq);                                                          \
        else                                                                                                                   \
            hipLaunchKernelGGL((preprocess_fwd_kernel<STAGE, COLOUR, false>), dim3(blocks), dim3(256), SMEM, st, a, g, radii,  \
                               depth_key, ids, vis_slots, blk_counts, host, seq);                                                          \
    } while (0)
    if (!colour) VCR_PRE(false, false, 0);
    else if (a.shs && a.K == SH_K) VCR_PRE(true, true, 256 * SH_ROW * sizeof(float));
    else VCR_PRE(false, true, 0);
#undef VCR_PRE
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// Workgroups of a side-stream kernel (persistent grid).  The side stream (1.15 kB per Gaussian of SH update) is the critical path of
// the front end at every size (profiles/r4_step_timeline.txt: it ends 20 us after the sort chain beside it), so it gets 1.5
// workgroups per CU up to 3 M Gaussians (round 4, profiles/r4_side_grid.txt: 256 / 320 / 384 / 512 / 768 workgroups -> step 1.355 /
// 1.374 / 1.345 / 1.356 / 1.367 ms at 1 M, 2.54 / 2.50 / 2.45 / 2.53 / 2.54 at 2 M: more of them slow the chain down by more
// than the update gains) and four per CU above (5 M, profiles/r4_side_grid_c5.txt: 512 / 640 / 768 / 1024 -> 3.79 / 3.92 / 3.78 / 3.735
// ms/step).
int vcr_side_grid(int N) {
    return N > 3000000 ? 1024 : 384;
}

int vcr_launch_sh_update_colour(const VcrRasterArgs& a, GeomState g, hipStream_t st) {
    const VcrShUpdate& u = *a.sh_update;
    if (a.N == 0) return 0;
    if (!a.shs || !a.shs_rest || a.K != SH_K) { vcr_set_error("sh_update needs the split SH storage (shs + shs_rest, K = 16)"); return 1; }
    if (u.sh_degree < 0 || u.sh_degree > 3 || u.step < 1 || u.nviews < 0 || !u.drgb || !u.m_dc || !u.v_dc || !u.m_rest || !u.v_rest ||
        (u.nviews == 0 ? !u.view_dirs : (!u.xyz || !u.campos_all))) { vcr_set_error("sh_update: bad arguments"); return 1; }
    const double bc1 = 1.0 - pow((double)u.beta1, u.step), bc2 = 1.0 - pow((double)u.beta2, u.step);
    const int nblk = (a.N + 255) / 256, grid = nblk < vcr_side_grid(a.N) ? nblk : vcr_side_grid(a.N);
    hipLaunchKernelGGL(sh_update_colour_kernel, dim3(grid), dim3(256), 256 * SH_ROW * sizeof(float), st, a, g, u,
                       (float)(u.lr_dc / bc1), (float)(u.lr_rest / bc1), (float)sqrt(bc2));
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

int vcr_launch_colour(const VcrRasterArgs& a, GeomState g, hipStream_t st) {
    if (a.N == 0) return 0;
    const int nblk = (a.N + 255) / 256, blocks = nblk < vcr_side_grid(a.N) ? nblk : vcr_side_grid(a.N);
    if (a.K == SH_K)
        hipLaunchKernelGGL(colour_fwd_kernel<true>, dim3(blocks), dim3(256), 256 * SH_ROW * sizeof(float), st, a, g);
    else
        hipLaunchKernelGGL(colour_fwd_kernel<false>, dim3(blocks), dim3(256), 0, st, a, g);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

int vcr_launch_preprocess_backward(const VcrRasterArgs& a, GeomState g, const int32_t* radii, GradRec* sgrad,
                                   float* sgrad_sem, VcrBackwardIO& io, hipStream_t st) {
    if (a.N == 0) return 0;
    const int blocks = (a.N + 255) / 256;
    const TailArgs none = {};
    if (a.shs && a.K == SH_K && io.dL_dshs)        // (LDS rows only for the coalesced write-back of a materialised SH gradient)
        hipLaunchKernelGGL((preprocess_bwd_kernel<true, false>), dim3(blocks), dim3(256), 256 * SH_ROW * sizeof(float), st, a, g,
                           radii, sgrad, sgrad_sem, io, none);
    else
        hipLaunchKernelGGL((preprocess_bwd_kernel<false, false>), dim3(blocks), dim3(256), 0, st, a, g, radii, sgrad, sgrad_sem, io,
                           none);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

// projection backward + the static tail of the iteration in one kernel (vcr_rasterize_backward_tail).  `io.dL_dmeans2D_densify`
// non-NULL only says WHICH screen gradient the densification statistics take (nothing is written through it).
int vcr_launch_preprocess_backward_tail(const VcrRasterArgs& a, GeomState g, const int32_t* radii, GradRec* sgrad,
                                        float* sgrad_sem, VcrBackwardIO& io, const VcrGeometryStep& t, hipStream_t st) {
    if (a.N == 0) return 0;
    TailArgs ta;
    ta.t = t;
    vcr_geometry_bias(t, ta.gb);
    hipLaunchKernelGGL((preprocess_bwd_kernel<false, true>), dim3((a.N + 255) / 256), dim3(256), 0, st, a, g, radii, sgrad,
                       sgrad_sem, io, ta);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
