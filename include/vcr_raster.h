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

#define VCR_ABI_VERSION 19

/* Buffers whose size depends on the number of tile instances R are obtained through this callback
 * (the Python shim backs it with torch's caching allocator, so no hipMalloc on the hot path).
 * tag: VCR_BUF_*.  Must return a device pointer aligned to 256 B that stays valid until the caller
 * releases it (state buffers: after the matching backward; scratch: after the call returns and
 * the stream has consumed it). */
typedef void* (*vcr_alloc_fn)(void* user, int32_t tag, size_t bytes);
enum { VCR_BUF_GEOM = 0, VCR_BUF_BINNING = 1, VCR_BUF_IMAGE = 2, VCR_BUF_SCRATCH = 3 };

/* Mirrors GaussianRasterizationSettings + the forward kwargs
 * (gaussian_renderer/__init__.py:43-57,107-120). */
/* Optional SH-coefficient update applied on the colour stream right before the SH -> RGB evaluation of a forward call
 * (VcrRasterArgs.sh_update): the Adam step of vcr_sh_adam_from_rgb (nviews == 0, gradient = basis(view_dirs) x drgb) or of
 * vcr_sh_adam_from_rgb_views (nviews > 0, gradient summed over the views from xyz / campos_all), fused with the colour
 * evaluation so that the updated coefficients are not read back from HBM.  Updates shs / shs_rest of the call IN PLACE
 * (split storage required). */
typedef struct VcrShUpdate {
    int32_t nviews, sh_degree, step;
    float grad_scale;
    const float* view_dirs;     /* [N,3], nviews == 0 */
    const float* drgb;          /* [N,3] or [nviews,N,3] */
    const float* xyz;           /* [N,3] means the views were rendered with, nviews > 0 */
    const float* campos_all;    /* [nviews,3], nviews > 0 */
    float* m_dc; float* v_dc; float* m_rest; float* v_rest;     /* Adam moments of features_dc / features_rest */
    float lr_dc, lr_rest, beta1, beta2, eps;
} VcrShUpdate;

typedef struct VcrRasterArgs {
    int32_t N;            /* Gaussians */
    int32_t H, W;         /* image_height, image_width */
    int32_t S;            /* semantic channels in semantics_precomp (0..4) */
    int32_t K;            /* SH coefficients stored per Gaussian in `shs` ((max_sh_degree+1)^2) */
    int32_t sh_degree;    /* active degree 0..3 */
    int32_t f_count;      /* 0 render, 1 count+score+image, 2 same (countlist), 3 count only; 4 (extension): like 3, but
                             count[i] is SET to 1 when Gaussian i contributes to any pixel instead of being incremented by
                             the number of pixels -- all that the consumer of the visibility passes reads (`> 0`,
                             tools/prune.py:64-66) */
    int32_t num_dist;     /* trailing channels: 0 none; 1 depth distortion A*M2 - M1^2 of the mapped depth
                             m = far/(far-near)*(1-near/d) (2DGS form; near .01, far 100); 2 depth moments (sum w d, sum w d^2) */
    int32_t debug;
    float tanfovx, tanfovy, scale_modifier;
    const float* bg;            /* [3]  */
    const float* viewmatrix;    /* [4,4] world_view_transform (row-vector convention) */
    const float* projmatrix;    /* [4,4] full_proj_transform */
    const float* campos;        /* [3]  */
    const float* means3D;       /* [N,3] */
    const float* shs;           /* [N,K,3] or NULL; with shs_rest: the DC coefficient only, [N,1,3] */
    const float* shs_rest;      /* NULL, or [N,K-1,3]: the reference's split storage (_features_dc/_features_rest,
                                   scene/gaussian_model.py:139-142) passed without the torch.cat of get_features */
    const float* colors_precomp;/* [N,3] or NULL (exactly one of shs/colors_precomp) */
    const float* normals_precomp;   /* [N,3] camera-space unit normals or NULL */
    const float* semantics_precomp; /* [N,S] or NULL */
    const float* opacities;     /* [N] */
    const float* scales;        /* [N,3] or NULL */
    const float* rotations;     /* [N,4] (w,x,y,z) unit, or NULL */
    const float* cov3D_precomp; /* [N,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float* dirs;          /* [3,H,W] unit pixel rays -> ray/plane ("intersection") depth; NULL -> centre depth */
    void* colour_stream;        /* optional second HIP stream: the SH -> RGB evaluation is launched there, behind the
                                   projection of this call, and joined before compositing, so that it overlaps the
                                   latency-bound sort chain; NULL -> one stream */
    void (*colour_stream_hook)(void* user);   /* optional: called once, on the host, after colour_stream has been made to
                                   wait for the projection and before the SH -> RGB launch: work the caller enqueues on
                                   colour_stream here (e.g. vcr_sh_adam_from_rgb of the previous iteration) runs beside
                                   the sort chain and ahead of the colour evaluation */
    void* colour_stream_hook_user;
    const VcrShUpdate* sh_update;   /* optional, with colour_stream: see VcrShUpdate */
    void* sort_stream;          /* optional third HIP stream: the depth keys and the depth sort of the N Gaussians run there,
                                   beside the projection, and are joined before the tile instances are emitted */
    int32_t quad_lists;         /* 0: tile instances are binned per 16x16 tile, the four 8x8-quad waves of a tile share its
                                   list.  1: binned per 8x8 quad -- the exact rejection of the projection runs per quad, the sort
                                   key is the quad index (2 more bits), every compositing wave walks the list of ITS quad only:
                                   no entry is gathered or culled four times (-33 % HBM fetch of both compositing kernels at
                                   1 M Gaussians / 1080p) at the price of 1.3-1.6x sort entries when footprints are small and
                                   up to 4x when they cover whole tiles.  Same image, same gradients (positions in a list are
                                   internal to a forward / backward pair).  vcr_rasterize_backward MUST be called with the value its forward was
                                   called with: the state buffers hold the lists in that form.  Pays for
                                   small footprints (R / V below ~4 tiles per visible Gaussian); images up to 8192 pixels. */
    int32_t forward_form;       /* (ABI 19) which kernel composites an f_count = 0 frame.  0: the library decides per frame from its
                                   counts (two-phase below VCR_TP_MAX_TILES_PER_GAUSSIAN = 5 3-sigma tiles per visible Gaussian,
                                   csrc/composite.hip); 1: the uniform loop (one survivor per iteration on all 64 lanes of a quad);
                                   2: the two-phase form (row-span candidate masks, per-lane candidate lists; S <= 2).  The two
                                   write bit-identical images and image state (tests/test_raster_parity_gpu.py), so the value
                                   need not be repeated to the backward. */
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
    int64_t  num_emitted;    /* R' <= R: tile instances really emitted -- tiles of a 3-sigma rectangle in which the Gaussian
                              * cannot reach alpha >= 1/255 at any pixel centre are rejected (result-preserving) */
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
    int64_t      num_emitted;    /* VcrForwardOut.num_emitted of the forward call (ABI 18): locates the per-chunk transmittance
                                    checkpoints the forward left behind the lists of the BINNING buffer */
    float* dL_dmeans3D;      /* [N,3] */
    float* dL_dmeans2D;      /* [N,3] (x,y in NDC units, z = 0) */
    float* dL_dmeans2D_densify; /* [N,3] sum over pixels of |per-pixel dL/dxy| (NDC units), or NULL */
    float* dL_dshs;          /* [N,K,3] (or [N,1,3] with dL_dshs_rest) or NULL */
    float* dL_dshs_rest;     /* [N,K-1,3] when shs_rest was given, else NULL */
    float* dL_dcolors;       /* [N,3] or NULL */
    float* dL_drgb;          /* optional [N,3]: dL/d(SH-evaluated colour) after the clamp mask, for the factorised
                                data-parallel exchange (vcr_sh_grad_from_rgb); dL_dshs may then be NULL */
    float* view_dirs;        /* optional [N,3]: unit view direction (mean - campos) used for the SH basis, written
                                together with dL_drgb for vcr_sh_adam_from_rgb */
    float* dL_dnormals;      /* [N,3] or NULL */
    float* dL_dsemantics;    /* [N,S] or NULL */
    float* dL_dopacities;    /* [N] */
    float* dL_dscales;       /* [N,3] or NULL */
    float* dL_drotations;    /* [N,4] or NULL */
    float* dL_dcov3D;        /* [N,6] or NULL */
    /* optional (ABI 18), both or neither: this render's camera rotation [3,3] and the aux bytes of vcr_activate_forward.  When
     * given, dL_dnormals is written as the gradient w.r.t. the WORLD-space shortest-axis column the normal was built from
     * (flip sign and camera rotation of this view undone: sgn * Rw2c^T dL/dn_cam) -- the form ranks of a data-parallel step
     * can sum, since it no longer depends on the rank's camera (vcr_geometry_step with normals_world = 1 consumes it). */
    const float*   normals_Rw2c;
    const uint8_t* normals_aux;
} VcrBackwardIO;

int vcr_abi_version(void);
const char* vcr_last_error(void);

/* replaces diff_gaussian_rasterization._C.rasterize_gaussians (forward of the autograd function
 * behind GaussianRasterizer.forward, gaussian_renderer/__init__.py:107) */
int vcr_rasterize_forward(const VcrRasterArgs* args, VcrForwardOut* out,
                          vcr_alloc_fn alloc, void* user, void* stream);
/* The visibility passes of a densification step (trainer.py:357-370,688-702; tools/prune.py:51-69 `get_visi_list`): the
 * reference renders `sample_cams.num` (TNT: 200) virtual cameras with f_count = 3, one rasterizer call each, and sums the
 * per-Gaussian counters.  Here ONE call takes all B cameras (same resolution): geometry-only projection (no SH -> RGB),
 * depth order, instance emission, tile sort and count-only compositing of up to `inflight` cameras run concurrently on
 * internal HIP streams (forked from / joined into `stream`), the host reads the instance counts of a camera while the
 * later cameras' front halves run, and the counters accumulate on the device.
 *   flags_only = 0: count[i] += number of pixels over all B cameras where Gaussian i passed the alpha and transmittance
 *                   tests -- identical to B calls of vcr_rasterize_forward with f_count = 3;
 *   flags_only = 1: count[i] = 1 if that number is > 0 (f_count = 4), else untouched.
 * Buffers come from `alloc` (tag VCR_BUF_SCRATCH) and may be released when the call returns. */
typedef struct VcrVisibilityBatch {
    int32_t N, H, W, B;
    int32_t flags_only;
    int32_t inflight;             /* cameras in flight (= buffer sets), 0 = 8, at most 16; they share up to 8 internal streams */
    int32_t quad_lists;           /* as VcrRasterArgs.quad_lists */
    float scale_modifier;
    const float* tanfovx;         /* HOST [B] */
    const float* tanfovy;         /* HOST [B] */
    const float* viewmatrix;      /* device [B,4,4] world_view_transform of every camera */
    const float* projmatrix;      /* device [B,4,4] full_proj_transform */
    const float* campos;          /* device [B,3] */
    const float* means3D;         /* [N,3] */
    const float* opacities;       /* [N] */
    const float* scales;          /* [N,3] or NULL */
    const float* rotations;       /* [N,4] or NULL */
    const float* cov3D_precomp;   /* [N,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    int32_t* count;               /* [N] device, accumulated */
    int64_t* num_rendered;        /* optional HOST [B]: 3-sigma tile instances of every camera */
    int32_t* num_visible;         /* optional HOST [B]: Gaussians with radii > 0 */
} VcrVisibilityBatch;
int vcr_visibility_batch(const VcrVisibilityBatch* args, vcr_alloc_fn alloc, void* user, void* stream);

/* replaces diff_gaussian_rasterization._C.rasterize_gaussians_backward (reached from
 * loss.backward(), trainer.py:338) */
int vcr_rasterize_backward(const VcrRasterArgs* args, VcrBackwardIO* io,
                           vcr_alloc_fn alloc, void* user, void* stream);

/* ---- per-Gaussian parameter kernels around the rasterizer call -------------------------------------
 * vcr_activate_*: replaces the chain GaussianModel.get_scaling/get_rotation/get_opacity
 * (scene/gaussian_model.py:125-162), get_normal (:168-192, tools/general_utils.py:98-119) and the
 * orientation + camera rotation of gaussian_renderer/__init__.py:95-101 by one pass each way.
 * R_w2c: [3,3] row-major device matrix with n_cam = R_w2c * n_world (= cam.R.T).  aux: [N] bytes. */
int vcr_activate_forward(int N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                         const float* xyz, const float* campos, const float* R_w2c, float* scales, float* rots,
                         float* opac, float* normals_cam /* may be NULL */, uint8_t* aux, void* stream);
int vcr_activate_backward(int N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                          const float* R_w2c, const uint8_t* aux, const float* d_scales, const float* d_rots,
                          const float* d_opac, const float* d_normals /* any may be NULL */, const float* d_scaling_extra /* optional [N,3]: added to d_scaling_raw (a gradient that reaches the raw scales on another path, e.g. l1_scale) */,
                          float* d_scaling_raw,
                          float* d_rotation_raw, float* d_opacity_raw, void* stream);
/* Data-parallel SH gradients without all-reducing them: per view the SH gradient is basis_k(dir) x dL/drgb, so ranks
 * all-gather dL/drgb (drgb_all [nviews,N,3]) and rebuild sum_v basis_k(normalize(xyz - campos_all[v])) * drgb_all[v]
 * into the split storage d_features_dc [N,1,3] / d_features_rest [N,15,3].  (No reference counterpart: DP is new.) */
int vcr_sh_grad_from_rgb(int N, int sh_degree, int nviews, const float* xyz, const float* campos_all,
                         const float* drgb_all, float* d_features_dc, float* d_features_rest, void* stream);
/* Single-view training: torch.optim.Adam on _features_dc [N,1,3] / _features_rest [N,15,3] (scene/gaussian_model.py:251-252,
 * trainer.py:389) with the SH gradient formed on the fly as basis_k(view_dirs) x drgb (both [N,3], written by
 * vcr_rasterize_backward through dL_drgb / view_dirs), so the 192 B/Gaussian SH gradient is never written or read.
 * Same update rule as vcr_adam_step (bias correction from `step`, eps added to sqrt(v)/sqrt(bc2)); coefficients above
 * sh_degree get a zero gradient, exactly as in the dense update.  Meant for a second stream (VcrRasterArgs.colour_stream). */
int vcr_sh_adam_from_rgb(int N, int sh_degree, const float* view_dirs, const float* drgb, float* features_dc,
                         float* features_rest, float* m_dc, float* v_dc, float* m_rest, float* v_rest, float lr_dc,
                         float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* Data-parallel form of vcr_sh_adam_from_rgb: gradient = sum_v basis_k(normalize(xyz - campos_all[v])) x drgb_all[v]
 * (drgb_all [nviews,N,3] from the all-gather of dL_drgb, xyz = the means the views were rendered with), scaled by
 * grad_scale (1/world), never materialised. */
int vcr_sh_adam_from_rgb_views(int N, int sh_degree, int nviews, const float* xyz, const float* campos_all,
                               const float* drgb_all, float* features_dc, float* features_rest, float* m_dc, float* v_dc,
                               float* m_rest, float* v_rest, float lr_dc, float lr_rest, float beta1, float beta2, float eps,
                               int step, float grad_scale, void* stream);
/* The rasterizer's own stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit) (at most
 * 32 bits = 4 passes), exposed for testing and reuse; replaces cub/rocPRIM DeviceRadixSort::SortPairs in the public
 * rasterizer's binning.  vals_in == NULL sorts the identity permutation.  scratch: vcr_sort_pairs_u32_scratch_bytes(n). */
size_t vcr_sort_pairs_u32_scratch_bytes(int64_t n);
int vcr_sort_pairs_u32(int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                       int begin_bit, int end_bit, void* scratch, size_t scratch_bytes, void* stream);
/* simple_knn._C.distCUDA2 (scene/gaussian_model.py:17-20,211): mean squared distance to the 3 nearest neighbours,
 * points [N,3] -> out [N].  Exact: a uniform grid of ~2 points per cell searched shell by shell (brute force up to 2048
 * points).  One-time initialisation from the SfM point cloud: synchronises the stream and allocates its scratch itself. */
int vcr_knn3_mean_dist2(int N, const float* points, float* out, void* stream);
/* One launch for all parameter groups; semantics of torch.optim.Adam(eps=1e-15) with per-group lr
 * (scene/gaussian_model.py:247-258).  Pointer arrays are HOST arrays of device pointers (<= 8 tensors).
 * grad_scale multiplies every gradient (1/world_size after a sum all-reduce). */
int vcr_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const int64_t* numel, const float* lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, void* stream);
/* The static tail of a training iteration in ONE pass over the Gaussians (no densify / prune / reset this iteration; data
 * parallel: on the all-reduced ACTIVATED-space gradients, grad_scale = 1 / world, normals_world = 1): adjoint of the fused activation (the four upstream gradients are those of vcr_rasterize_backward w.r.t. the
 * ACTIVATED scales / rotations / opacities / camera-space normals of this iteration's render; `aux`, `Rw2c` as saved by
 * vcr_activate_forward) plus the l1_scale gradient (trainer.py:243-245; scale_reg_* NULL = none) -> densification statistics
 * (scene/gaussian_model.py:669-671, trainer.py:345; grad2d NULL = none) -> one torch.optim.Adam(eps) step on
 * xyz / scaling / rotation / opacity (scene/gaussian_model.py:232-262; d_means3D NULL = xyz untouched) -> fused activation of
 * the UPDATED parameters for the next render's camera (next_* NULL = none).  Replaces vcr_activate_backward +
 * vcr_scale_reg_backward + vcr_densify_stats + vcr_adam_step + vcr_activate_forward on those four groups. */
typedef struct VcrGeometryStep {
    int32_t N;
    int32_t normals_world;                                           /* 1: d_normals is w.r.t. the world-space axis column
                                                                        (VcrBackwardIO.normals_Rw2c), summed over ranks */
    int32_t step_xyz, step_scaling, step_rotation, step_opacity;     /* torch counts Adam steps per tensor (>= 1) */
    float *xyz, *scaling, *rotation, *opacity;                       /* raw parameters, updated in place */
    const float *d_means3D, *d_scales, *d_rots, *d_opac, *d_normals; /* upstream gradients (any may be NULL) */
    const uint8_t* aux; const float* Rw2c;                           /* of this iteration's activation */
    const float* scale_reg_gout; const double* scale_reg_sums;       /* [1] weight x seed; sums[2] = Gaussians inside the box */
    const float *trans, *scale;                                      /* [3] each: normalised bounding box */
    float *m_xyz, *v_xyz, *m_scaling, *v_scaling, *m_rotation, *v_rotation, *m_opacity, *v_opacity;
    float lr_xyz, lr_scaling, lr_rotation, lr_opacity, beta1, beta2, eps;
    float grad_scale;                                                /* > 0: multiplies the five upstream gradients (1 / world
                                                                        after a sum all-reduce); the l1_scale term is added once */
    const float* grad2d; const int32_t* radii; float *accum, *denom, *max_radii;
    const float *next_campos, *next_Rw2c;
    float *next_scales, *next_rots, *next_opac, *next_normals; uint8_t* next_aux;
} VcrGeometryStep;
int vcr_geometry_step(const VcrGeometryStep* args, void* stream);
/* vcr_rasterize_backward with the static tail of the iteration INSIDE its last kernel: the projection backward hands its
 * per-Gaussian gradients (w.r.t. means3D, the densification screen gradient, activated scales / rotations / opacities,
 * camera-space normals) to vcr_geometry_step's arithmetic in registers -- they are never written, and io->dL_dmeans3D /
 * dL_dmeans2D / dL_dmeans2D_densify / dL_dopacities / dL_dscales / dL_drotations / dL_dnormals are not dereferenced
 * (dL_dmeans2D_densify non-NULL only selects the densify-variant of the screen gradient for the statistics, like the
 * `means2D_densify.grad` of trainer.py:345).  What still has a consumer is written as before: dL_drgb + view_dirs (the SH
 * update), dL_dcolors, dL_dsemantics.  `tail`: as for vcr_geometry_step with d_means3D / d_scales / d_rots / d_opac /
 * d_normals / grad2d / radii all NULL (statistics run when accum is non-NULL; radii are the render's own).  Requires the
 * scale / rotation form of the covariance and no materialised SH gradient (dL_dshs NULL).  Same per-Gaussian functions as the
 * two separate kernels (preprocess.hip / model_math.h). */
int vcr_rasterize_backward_tail(const VcrRasterArgs* args, VcrBackwardIO* io, const VcrGeometryStep* tail,
                                vcr_alloc_fn alloc, void* user, void* stream);

/* add_densification_stats + max_radii2D update (scene/gaussian_model.py:669-671, trainer.py:345) */
int vcr_densify_stats(int N, const float* grad2d /*[N,3]*/, const int32_t* radii, float* accum, float* denom,
                      float* max_radii, void* stream);

/* ---- image-space kernels of the D-Normal losses ------------------------------------------------------
 * compute_normals (tools/normal_utils.py:30-41): depth [H,W] -> unit normal [H,W,3]; scratch6: [H*W*6]. */
int vcr_depth_to_normal_forward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                float* normal, void* stream);
int vcr_depth_to_normal_backward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                                 const float* dnormal, float* scratch6, float* ddepth, void* stream);
/* rendered normal [3,H,W] -> F.normalize -> [H,W,3] (gaussian_renderer/__init__.py:133-134) */
int vcr_normalize_chw_forward(int P, const float* in_chw, float* out_hwc, void* stream);
int vcr_normalize_chw_backward(int P, const float* in_chw, const float* dout_hwc, float* din_chw, void* stream);
/* The *_forward functions of the loss kernels take `sums_prezeroed`: bit 0 = the sums buffer is already zero, bit 1 = do
 * not launch the per-loss finalize.  vcr_finalize_losses then finishes the L1+SSIM sums (sums2), the scale regulariser
 * (sums3 or NULL) and the three normal losses (sums9 or NULL) in ONE launch: res6 = {l1, ssim index, l1_scale,
 * mono_normal, depth_normal, consistent_normal}, total = sum res6[k] w[k] - (sub_index >= 0 ? w[sub_index] : 0). */
int vcr_finalize_losses(int H, int W, double* sums2, double* sums3, double* sums9, float* res6, const float* w, int sub_index,
                        float* total, void* stream);
/* total = sum_k res[k] * w[k] - (sub_index >= 0 ? w[sub_index] : 0): the weighted sum of the loss dictionary
 * (trainer.py:310-321) in one launch; sub_index marks the entry that enters as (1 - value), i.e. SSIM. */
int vcr_weighted_total(int K, const float* res, const float* w, int sub_index, float* total, void* stream);
/* monosdf_normal_loss with the cos_weight confidence and boolean mask fused in
 * (tools/loss_utils.py:122-143, trainer.py:261-293).  sums3 (device, fp64, vcr_sums_elems(3) doubles: results
 * first, reduction slots behind) = {sum w|p-g|_1, sum w(1-p.g), count}; loss = (sums[0]+sums[1])/sums[2].
 * wsrc NULL or exp_t<=0 -> w=1.  dgt may be NULL. */
int vcr_sums_elems(int k);
/* Extra pixel selection fused in: `mask` (bytes, may be NULL) AND, when `depth` != NULL, depth[i] < depth_max
 * (the mask_depth_thr test of gaussian_renderer/__init__.py:125-131).  `loss` (device float[1]) receives the value. */
int vcr_normal_loss_forward(int P, const float* pred, const float* gt, const float* wsrc, float exp_t,
                            const uint8_t* mask, const float* depth, float depth_max, double* sums3, float* loss,
                            int sums_prezeroed /* caller already zeroed sums3 (one memset for several losses) */, void* stream);
int vcr_normal_loss_backward(int P, const float* pred, const float* gt, const float* wsrc, float exp_t,
                             const uint8_t* mask, const float* depth, float depth_max, const double* sums3,
                             const float* gout, float* dpred, float* dgt,
                             int accumulate /* bit0: dpred +=, bit1: dgt += (several losses on one tensor) */, void* stream);
/* l1_scale regulariser: mean over Gaussians inside the bounding box of min_axis(exp(_scaling))
 * (trainer.py:243-245, tools/math_utils.py:50-74 with vector trans/scale).  sums3 as above. */
int vcr_scale_reg_forward(int N, const float* scaling_raw, const float* xyz, const float* trans, const float* scale,
                          double* sums3, float* loss, int sums_prezeroed, void* stream);
int vcr_scale_reg_backward(int N, const float* scaling_raw, const float* xyz, const float* trans, const float* scale,
                           const double* sums3, const float* gout, float* dscaling, void* stream);
/* All three normal losses of trainer.py:261-293 in one pass each way on the rasterizer output: mono_normal (bit 0 of
 * `active`) = monosdf_normal_loss(n, gt), depth_normal (bit 1) = the same on est with the cos_weight confidence, the
 * camera mask and the depth threshold (depth_max <= 0: none), consistent_normal (bit 2) = monosdf_normal_loss(est, n),
 * where n = F.normalize(normal_planes [3,H,W]) and est = compute_normals(depth [H,W]).  sums9: vcr_sums_elems(9) doubles;
 * res3 / seeds3: the three loss values / their upstream gradients.  The backward writes d(depth) [H,W] and
 * d(normal_planes) [3,H,W] followed by a zeroed fourth plane (the alpha plane of the rasterizer output: d_normal_planes
 * must have room for [4,H,W]); scratch6 is [H*W*6] floats. */
int vcr_normal_losses_forward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                              const float* normal_planes, const float* gt /* [H*W,3] or NULL */,
                              const uint8_t* mask /* [H*W] or NULL */, float depth_max, float exp_t, int active,
                              double* sums9, float* res3, int sums_prezeroed, void* stream);
int vcr_normal_losses_backward(int H, int W, float fx, float fy, float cx, float cy, const float* depth,
                               const float* normal_planes, const float* gt, const uint8_t* mask, float depth_max,
                               float exp_t, int active, const double* sums9, const float* seeds3, float* scratch6,
                               float* d_depth, float* d_normal_planes, void* stream);
/* Small regularisers of Trainer._compute_loss.  sums1 / sums3: vcr_sums_elems(1) / vcr_sums_elems(3) doubles (zeroed inside);
 * `loss`: device float[1].
 *  - edge-aware mean (tools/normal_utils.py:57-66 followed by .mean(), trainer.py:295-303): mean over H*W of
 *    map * exp(-max over the 4 neighbours of mean_c |I - I_nb|) on interior pixels, 0 on the border; gradient to `map`.
 *  - normal curvature (tools/loss_utils.py:287-300 + l1_loss(curv, 0), trainer.py:282-287): normal [H,W,3], mask bytes
 *    [H,W] (replicate padding, masked 4-neighbour Laplacian, L1 norm over xyz, mean over H*W); gradient to `normal`.
 *  - opacity entropy (tools/loss_utils.py:30-33, trainer.py:247-249) on sigmoid(opacity_raw) over the Gaussians inside the
 *    normalised bounding box (xyz NULL: all); gradient to the RAW opacities. */
int vcr_edge_aware_forward(int H, int W, const float* gt_image, const float* map, double* sums1, float* loss, void* stream);
int vcr_edge_aware_backward(int H, int W, const float* gt_image, const float* gout, float* dmap, void* stream);
int vcr_curv_forward(int H, int W, const float* normal_hwc, const uint8_t* mask, double* sums1, float* loss, void* stream);
int vcr_curv_backward(int H, int W, const float* normal_hwc, const uint8_t* mask, const float* gout, float* dnormal,
                      void* stream);
int vcr_entropy_forward(int N, const float* opacity_raw, const float* xyz, const float* trans, const float* scale,
                        double* sums3, float* loss, void* stream);
int vcr_entropy_backward(int N, const float* opacity_raw, const float* xyz, const float* trans, const float* scale,
                         const double* sums3, const float* gout, float* dopacity_raw, void* stream);
/* Densify / prune row surgery (scene/gaussian_model.py:425-531: _prune_optimizer, cat_tensors_to_optimizer,
 * densification_postfix) for ALL per-Gaussian arrays of the model (parameters, both Adam moments, statistics) in one launch.
 *   vcr_rows_plan : offsets (vcr_rows_plan_bytes(N) bytes) <- per-256-row-block exclusive scan of mask; offsets[nblk] = M.
 *   vcr_rows_move : mode 0: out[k] = in[k][mask]  ([M, width]);
 *                   mode 1: out[k] = cat(in[k], in[k][mask] x copies) ([N + copies*M, width]), appended rows zero-filled
 *                   where zero_new is set (the Adam moments of new Gaussians).  Arrays are float rows of `width` floats. */
#define VCR_MAX_ROW_ARRAYS 32
typedef struct VcrRowArray { const float* in; float* out; int32_t width; int32_t zero_new; } VcrRowArray;
typedef struct VcrRowArrays { VcrRowArray a[VCR_MAX_ROW_ARRAYS]; int32_t n; } VcrRowArrays;
size_t vcr_rows_plan_bytes(int N);
int vcr_rows_plan(int N, const uint8_t* mask, uint32_t* offsets, void* stream);
int vcr_rows_move(int N, const uint8_t* mask, const uint32_t* offsets, const VcrRowArrays* arrays, int mode, int copies,
                  void* stream);
/* Semantic loss (gaussian_renderer/__init__.py:146-148 + trainer.py:304-307): 1x1-conv classifier on the rendered feature
 * planes sem [S,P] + F.cross_entropy(logits, labels) / log(K) in one pass each way.  W [K,S] / b [K]: the classifier's
 * parameters (device); labels: device int64 [P]; S <= 4, 2 <= K <= 8.  The backward writes d(sem) [S,P] and the
 * classifier's gradients dW [K,S], db [K] (device). */
int vcr_semantic_ce_forward(long long P, int S, int K, const float* sem, const float* W, const float* b,
                            const long long* labels, double* sums1, float* loss, void* stream);
int vcr_semantic_ce_backward(long long P, int S, int K, const float* sem, const float* W, const float* b,
                             const long long* labels, const float* gout, float* dsem, float* dW, float* db, void* stream);
/* Depth -> TSDF input (tools/graphics_utils.py:134-141 depth2point, tools/depth2mesh.py:37-52): depth_out = depth, zeroed
 * where gt_alpha < 0.5 (gt_alpha may be NULL), alpha < alpha_thres (alpha may be NULL) or the back-projected world point
 * is outside the normalised bounding box |(p - trans) / scale| < 1 (trans NULL: no box test); xyz_cam / xyz_world
 * ([H,W,3], may be NULL) receive depth2point of the UNMASKED depth.  c2w_rowmajor16: HOST pointer to the row-major 4x4
 * camera-to-world matrix (inverse of the reference's `extrinsic_matrix` = world_view_transform^T). */
int vcr_tsdf_depth_input(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_rowmajor16, const float* trans,
                         const float* scale, const float* depth, const float* alpha, float alpha_thres, const float* gt_alpha,
                         float* depth_out, float* xyz_cam, float* xyz_world, void* stream);
/* l1_loss + ssim (tools/loss_utils.py:36,49-92) in one pass over [3,H,W] images.  sums2 (device, fp64) =
 * {sum|a-b|, sum ssim_map}; partials9: [9,H,W] scratch kept for backward (NULL for inference). */
int vcr_l1_ssim_forward(int H, int W, const float* img1, const float* img2, double* sums2, float* means2 /* {l1, ssim} */,
                        float* partials9, int sums_prezeroed, void* stream);
int vcr_l1_ssim_backward(int H, int W, const float* img1, const float* img2, const float* partials9, const float* g_l1,
                         const float* g_ssim, float* dimg1, void* stream);

/* A HIP stream confined to the compute units whose bits are set in `mask` (hipExtStreamCreateWithCUMask; consecutive bits
 * are dealt round-robin over the XCDs, so the lowest M bits are M CUs spread over the chip) -- for VcrRasterArgs.colour_stream:
 * the streaming SH kernel then leaves the remaining CUs to the sort chain.  Returns NULL on error. */
void* vcr_stream_create_cu_masked(const uint32_t* mask, int nwords);
int vcr_stream_destroy(void* stream);
/* (ABI 19) The library keeps a few device blocks per (host thread, device, stream) between calls -- ticket / look-back words and
 * the backward's per-Gaussian accumulators, (64 + 4 S) x N x 1.25 bytes (80 MB at 1 M Gaussians), see INTEGRATION.md section 3.
 * This frees the calling thread's blocks on every device (synchronises those devices); later calls allocate them again. */
int vcr_release_scratch(void);

/* Optional per-stage timing with HIP events recorded on the launch stream (used by bench.py for the
 * live roofline figure; no reference counterpart).  Stage order: preprocess, depth sort+scan,
 * duplicate+tile sort+ranges, composite forward, composite backward, preprocess backward. */
void vcr_profile_enable(int on);
void vcr_profile_select(unsigned stage_mask);   /* bit k = time stage k (default: all); an event pair costs a few us of stream time */
int  vcr_profile_num_stages(void);
int  vcr_profile_read(float* ms, int32_t* launches, int n);
/* Instrumented builds only (-DVCR_HITHIST, see profiles/hit_histogram.py): how many of the 64 pixels of an 8x8 quad a
 * surviving (quad, Gaussian) pair really hits.  out[0..64]: forward, out[65..129]: backward; accumulated over all launches
 * since the last reset.  Returns 1 in ordinary builds. */
int  vcr_debug_hit_histogram(uint32_t out[130], int reset);
/* Diagnostics (not thread-safe): with `on`, every vcr_rasterize_backward of the process keeps a copy of its screen-space accumulators
 * (GradRec [N] = 16 floats per Gaussian, raw sums as the compositing backward left them, see csrc/vcr_common.h) before the
 * projection backward consumes them; vcr_debug_read_sgrad copies the last one to HOST memory (synchronises the device).
 * Used by profiles/grad_stage_errors.py to tell the error of the compositing backward from that of the projection backward. */
int  vcr_debug_keep_sgrad(int on);
int  vcr_debug_read_sgrad(float* host_out, int N);

#ifdef __cplusplus
}
#endif
#endif /* VCR_RASTER_H */
