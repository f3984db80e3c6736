/*
 * vcr_raster.h — C ABI of libvcr_raster.so, the MI355X (gfx950) differentiable Gaussian rasterizer
 * and fused D-Normal loss / optimizer kernels.
 *
 * This is the drop-in boundary for the reference's un-vendored CUDA extension
 * `diff_gaussian_rasterization` (reference: .gitmodules:4-6; Python call sites
 * gaussian_renderer/__init__.py:43-59 (settings), :107-120 (forward kwargs), :332-344 (f_count=1),
 * :441-453 (f_count=2), :550-562 (f_count=3)).  Plain pointers and sizes only: every pointer is a
 * DEVICE pointer to contiguous fp32 / int32 data unless stated otherwise; `stream` is a hipStream_t
 * passed as void*.  All functions return 0 on success, non-zero on error; `vcr_last_error()` gives
 * the message (the Python shim raises RuntimeError, like the reference extension does).
 */
#ifndef VCR_RASTER_H
#define VCR_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCR_ABI_VERSION 1

/* Buffers whose size depends on the number of tile instances R are obtained through this callback
 * (the Python shim backs it with torch's caching allocator, so no hipMalloc on the hot path).
 * tag: VCR_BUF_*.  Must return a device pointer aligned to 256 B that stays valid until the caller
 * releases it (state buffers: after the matching backward; scratch: after the call returns and
 * the stream has consumed it). */
typedef void* (*vcr_alloc_fn)(void* user, int32_t tag, size_t bytes);
enum { VCR_BUF_GEOM = 0, VCR_BUF_BINNING = 1, VCR_BUF_IMAGE = 2, VCR_BUF_SCRATCH = 3 };

/* Mirrors GaussianRasterizationSettings + the forward kwargs
 * (gaussian_renderer/__init__.py:43-57,107-120). */
typedef struct VcrRasterArgs {
    int32_t N;            /* Gaussians */
    int32_t H, W;         /* image_height, image_width */
    int32_t S;            /* semantic channels in semantics_precomp (0..4) */
    int32_t K;            /* SH coefficients stored per Gaussian in `shs` ((max_sh_degree+1)^2) */
    int32_t sh_degree;    /* active degree 0..3 */
    int32_t f_count;      /* 0 render, 1 count+score+image, 2 same (countlist), 3 count only */
    int32_t num_dist;     /* trailing channels: 0 none, 1 distortion, 2 depth moments (sum w d, sum w d^2) */
    int32_t debug;
    float tanfovx, tanfovy, scale_modifier;
    const float* bg;            /* [3]  */
    const float* viewmatrix;    /* [4,4] world_view_transform (row-vector convention) */
    const float* projmatrix;    /* [4,4] full_proj_transform */
    const float* campos;        /* [3]  */
    const float* means3D;       /* [N,3] */
    const float* shs;           /* [N,K,3] or NULL */
    const float* colors_precomp;/* [N,3] or NULL (exactly one of shs/colors_precomp) */
    const float* normals_precomp;   /* [N,3] camera-space unit normals or NULL */
    const float* semantics_precomp; /* [N,S] or NULL */
    const float* opacities;     /* [N] */
    const float* scales;        /* [N,3] or NULL */
    const float* rotations;     /* [N,4] (w,x,y,z) unit, or NULL */
    const float* cov3D_precomp; /* [N,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float* dirs;          /* [3,H,W] unit pixel rays -> ray/plane ("intersection") depth; NULL -> centre depth */
} VcrRasterArgs;

/* Forward outputs.  `out`, `radii`, counters are caller-allocated. */
typedef struct VcrForwardOut {
    float*   out;        /* [C,H,W], C = 8 + S + num_dist: colour3 depth1 normal3 alpha1 sem S dist */
    int32_t* radii;      /* [N] */
    int32_t* count;      /* [N] accumulated (+=) when f_count != 0, else may be NULL */
    float*   score;      /* [N] accumulated (+=) when f_count is 1 or 2, else may be NULL */
    /* state for backward, filled by the call (pointers obtained through the allocator) */
    void*    geom;       /* VCR_BUF_GEOM    */
    void*    binning;    /* VCR_BUF_BINNING */
    void*    image;      /* VCR_BUF_IMAGE   */
    int64_t  num_rendered;   /* R = number of (Gaussian, tile) instances */
    int32_t  num_visible;    /* V = Gaussians with radii > 0 */
    int32_t  max_tile_len;   /* longest per-tile list (only when debug != 0, else -1) */
} VcrForwardOut;

/* Backward.  All gradient outputs are caller-allocated and fully overwritten (no pre-zeroing
 * needed); NULL for the member of an either/or pair that was not used in forward. */
typedef struct VcrBackwardIO {
    const float* dL_dout;    /* [C,H,W] */
    const void*  geom;       /* state from forward */
    const void*  binning;
    const void*  image;
    const int32_t* radii;    /* [N] as returned by forward */
    int64_t      num_rendered;
    float* dL_dmeans3D;      /* [N,3] */
    float* dL_dmeans2D;      /* [N,3] (x,y in NDC units, z = 0) */
    float* dL_dmeans2D_densify; /* [N,3] sum over pixels of |per-pixel dL/dxy| (NDC units), or NULL */
    float* dL_dshs;          /* [N,K,3] or NULL */
    float* dL_dcolors;       /* [N,3] or NULL */
    float* dL_dnormals;      /* [N,3] or NULL */
    float* dL_dsemantics;    /* [N,S] or NULL */
    float* dL_dopacities;    /* [N] */
    float* dL_dscales;       /* [N,3] or NULL */
    float* dL_drotations;    /* [N,4] or NULL */
    float* dL_dcov3D;        /* [N,6] or NULL */
} VcrBackwardIO;

int vcr_abi_version(void);
const char* vcr_last_error(void);

/* replaces diff_gaussian_rasterization._C.rasterize_gaussians (forward of the autograd function
 * behind GaussianRasterizer.forward, gaussian_renderer/__init__.py:107) */
int vcr_rasterize_forward(const VcrRasterArgs* args, VcrForwardOut* out,
                          vcr_alloc_fn alloc, void* user, void* stream);
/* replaces diff_gaussian_rasterization._C.rasterize_gaussians_backward (reached from
 * loss.backward(), trainer.py:338) */
int vcr_rasterize_backward(const VcrRasterArgs* args, VcrBackwardIO* io,
                           vcr_alloc_fn alloc, void* user, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VCR_RASTER_H */
