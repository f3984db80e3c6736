"""ctypes binding of libvcr_raster.so (C ABI: include/vcr_raster.h).

There is deliberately NO fallback: if the HIP library is missing or does not load, importing this
module raises, so a product path can never silently run on something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VCR_LIB: another build of the same library (A/B measurements of kernel variants); never a different backend
LIB_PATH = os.environ.get("VCR_LIB") or os.path.join(_HERE, "libvcr_raster.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t)
HOOK_FN = C.CFUNCTYPE(None, C.c_void_p)
BUF_GEOM, BUF_BINNING, BUF_IMAGE, BUF_SCRATCH = 0, 1, 2, 3
ABI_VERSION = 19         # VCR_ABI_VERSION of include/vcr_raster.h this binding was written against


class VcrShUpdate(C.Structure):
    _fields_ = [
        ("nviews", C.c_int32), ("sh_degree", C.c_int32), ("step", C.c_int32), ("grad_scale", C.c_float),
        ("view_dirs", C.c_void_p), ("drgb", C.c_void_p), ("xyz", C.c_void_p), ("campos_all", C.c_void_p),
        ("m_dc", C.c_void_p), ("v_dc", C.c_void_p), ("m_rest", C.c_void_p), ("v_rest", C.c_void_p),
        ("lr_dc", C.c_float), ("lr_rest", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
    ]


class VcrGeometryStep(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("N", "normals_world", "step_xyz", "step_scaling", "step_rotation", "step_opacity")] + [(k, C.c_void_p) for k in (
        "xyz", "scaling", "rotation", "opacity", "d_means3D", "d_scales", "d_rots", "d_opac", "d_normals", "aux", "Rw2c",
        "scale_reg_gout", "scale_reg_sums", "trans", "scale", "m_xyz", "v_xyz", "m_scaling", "v_scaling", "m_rotation",
        "v_rotation", "m_opacity", "v_opacity")] + [(k, C.c_float) for k in (
            "lr_xyz", "lr_scaling", "lr_rotation", "lr_opacity", "beta1", "beta2", "eps", "grad_scale")] + [(k, C.c_void_p) for k in (
                "grad2d", "radii", "accum", "denom", "max_radii", "next_campos", "next_Rw2c", "next_scales", "next_rots",
                "next_opac", "next_normals", "next_aux")]


MAX_ROW_ARRAYS = 32


class VcrRowArray(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("width", C.c_int32), ("zero_new", C.c_int32)]


class VcrRowArrays(C.Structure):
    _fields_ = [("a", VcrRowArray * MAX_ROW_ARRAYS), ("n", C.c_int32)]


class VcrRasterArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("S", C.c_int32), ("K", C.c_int32),
        ("sh_degree", C.c_int32), ("f_count", C.c_int32), ("num_dist", C.c_int32), ("debug", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("shs_rest", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("normals_precomp", C.c_void_p), ("semantics_precomp", C.c_void_p), ("opacities", C.c_void_p),
        ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("dirs", C.c_void_p),
        ("colour_stream", C.c_void_p), ("colour_stream_hook", C.c_void_p), ("colour_stream_hook_user", C.c_void_p), ("sh_update", C.c_void_p), ("sort_stream", C.c_void_p),
        ("quad_lists", C.c_int32), ("forward_form", C.c_int32),
    ]


class VcrForwardOut(C.Structure):
    _fields_ = [
        ("out", C.c_void_p), ("radii", C.c_void_p), ("count", C.c_void_p), ("score", C.c_void_p),
        ("geom", C.c_void_p), ("binning", C.c_void_p), ("image", C.c_void_p),
        ("num_rendered", C.c_int64), ("num_visible", C.c_int32), ("max_tile_len", C.c_int32), ("num_emitted", C.c_int64),
    ]


class VcrVisibilityBatch(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("B", C.c_int32), ("flags_only", C.c_int32),
                ("inflight", C.c_int32), ("quad_lists", C.c_int32), ("scale_modifier", C.c_float), ("tanfovx", c_float_p), ("tanfovy", c_float_p)] + \
               [(k, C.c_void_p) for k in ("viewmatrix", "projmatrix", "campos", "means3D", "opacities", "scales", "rotations",
                                          "cov3D_precomp", "count")] + \
               [("num_rendered", C.POINTER(C.c_int64)), ("num_visible", c_int_p)]


class VcrBackwardIO(C.Structure):
    _fields_ = [
        ("dL_dout", C.c_void_p), ("geom", C.c_void_p), ("binning", C.c_void_p), ("image", C.c_void_p),
        ("radii", C.c_void_p), ("num_rendered", C.c_int64), ("num_emitted", C.c_int64),
        ("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dmeans2D_densify", C.c_void_p),
        ("dL_dshs", C.c_void_p), ("dL_dshs_rest", C.c_void_p), ("dL_dcolors", C.c_void_p), ("dL_drgb", C.c_void_p), ("view_dirs", C.c_void_p), ("dL_dnormals", C.c_void_p),
        ("dL_dsemantics", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
        ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("normals_Rw2c", C.c_void_p), ("normals_aux", C.c_void_p),
    ]


# symbol -> (restype, argtypes); tests check that every one of these is exported.
SYMBOLS = {
    "vcr_abi_version": (C.c_int, []),
    "vcr_last_error": (C.c_char_p, []),
    "vcr_rasterize_forward": (C.c_int, [C.POINTER(VcrRasterArgs), C.POINTER(VcrForwardOut), ALLOC_FN, C.c_void_p, C.c_void_p]),
    "vcr_rasterize_backward": (C.c_int, [C.POINTER(VcrRasterArgs), C.POINTER(VcrBackwardIO), ALLOC_FN, C.c_void_p, C.c_void_p]),
    "vcr_visibility_batch": (C.c_int, [C.POINTER(VcrVisibilityBatch), ALLOC_FN, C.c_void_p, C.c_void_p]),
    "vcr_activate_forward": (C.c_int, [C.c_int] + [C.c_void_p] * 12),
    "vcr_activate_backward": (C.c_int, [C.c_int] + [C.c_void_p] * 14),
    "vcr_sort_pairs_u32_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "vcr_sort_pairs_u32": (C.c_int, [C.c_int64] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vcr_normal_losses_forward": (C.c_int, [C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vcr_normal_losses_backward": (C.c_int, [C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int]
                                   + [C.c_void_p] * 6),
    "vcr_finalize_losses": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]),
    "vcr_weighted_total": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vcr_sh_grad_from_rgb": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "vcr_sh_adam_from_rgb_views": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_float] * 5
                                   + [C.c_int, C.c_float, C.c_void_p]),
    "vcr_sh_adam_from_rgb": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_float] * 5 + [C.c_int, C.c_float, C.c_void_p]),
    "vcr_knn3_mean_dist2": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vcr_adam_step": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_int64), c_float_p, C.c_float, C.c_float,
                                C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "vcr_geometry_step": (C.c_int, [C.POINTER(VcrGeometryStep), C.c_void_p]),
    "vcr_rasterize_backward_tail": (C.c_int, [C.POINTER(VcrRasterArgs), C.POINTER(VcrBackwardIO), C.POINTER(VcrGeometryStep), ALLOC_FN,
                                              C.c_void_p, C.c_void_p]),
    "vcr_densify_stats": (C.c_int, [C.c_int] + [C.c_void_p] * 6),
    "vcr_depth_to_normal_forward": (C.c_int, [C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 3),
    "vcr_depth_to_normal_backward": (C.c_int, [C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 5),
    "vcr_normalize_chw_forward": (C.c_int, [C.c_int] + [C.c_void_p] * 3),
    "vcr_normalize_chw_backward": (C.c_int, [C.c_int] + [C.c_void_p] * 4),
    "vcr_normal_loss_forward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vcr_normal_loss_backward": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p]),
    "vcr_scale_reg_forward": (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "vcr_scale_reg_backward": (C.c_int, [C.c_int] + [C.c_void_p] * 8),
    "vcr_sums_elems": (C.c_int, [C.c_int]),
    "vcr_rows_plan_bytes": (C.c_size_t, [C.c_int]),
    "vcr_rows_plan": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vcr_rows_move": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(VcrRowArrays), C.c_int, C.c_int, C.c_void_p]),
    "vcr_semantic_ce_forward": (C.c_int, [C.c_longlong, C.c_int, C.c_int] + [C.c_void_p] * 7),
    "vcr_semantic_ce_backward": (C.c_int, [C.c_longlong, C.c_int, C.c_int] + [C.c_void_p] * 9),
    "vcr_tsdf_depth_input": (C.c_int, [C.c_int, C.c_int] + [C.c_float] * 4 + [C.POINTER(C.c_float)] + [C.c_void_p] * 4
                             + [C.c_float] + [C.c_void_p] * 4 + [C.c_void_p]),
    "vcr_edge_aware_forward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5),
    "vcr_edge_aware_backward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 4),
    "vcr_curv_forward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5),
    "vcr_curv_backward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5),
    "vcr_entropy_forward": (C.c_int, [C.c_int] + [C.c_void_p] * 7),
    "vcr_entropy_backward": (C.c_int, [C.c_int] + [C.c_void_p] * 8),
    "vcr_l1_ssim_forward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "vcr_l1_ssim_backward": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 7),
    "vcr_stream_create_cu_masked": (C.c_void_p, [C.POINTER(C.c_uint32), C.c_int]),
    "vcr_stream_destroy": (C.c_int, [C.c_void_p]),
    "vcr_release_scratch": (C.c_int, []),
    "vcr_profile_enable": (None, [C.c_int]),
    "vcr_profile_select": (None, [C.c_uint]),
    "vcr_profile_num_stages": (C.c_int, []),
    "vcr_profile_read": (C.c_int, [c_float_p, c_int_p, C.c_int]),
    "vcr_debug_hit_histogram": (C.c_int, [C.POINTER(C.c_uint32), C.c_int]),
    "vcr_debug_keep_sgrad": (C.c_int, [C.c_int]),
    "vcr_debug_read_sgrad": (C.c_int, [C.c_void_p, C.c_int]),
}
STAGES = ["preprocess", "depth_sort_scan", "binning", "composite_fwd", "composite_bwd", "preprocess_bwd"]

_lib = None


def load():
    """Load (once) and return the ctypes handle.  torch must be imported first so that the HIP
    runtime already mapped by torch (SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps libamdhip64 before our library resolves it)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). No CPU fallback exists.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.vcr_abi_version() != ABI_VERSION:
        raise ImportError("libvcr_raster.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("vcr_raster: " + last_error())


def stream_of(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def last_error():
    return load().vcr_last_error().decode()


def cu_masked_stream(num_cus, device, total_cus=256):
    """torch.cuda.ExternalStream confined to `num_cus` compute units spread evenly over the XCDs (the lowest `num_cus` bits of
    the driver's logical CU mask).  The HIP stream lives as long as the process."""
    import torch
    lib = load()
    words = (total_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(min(num_cus, total_cus)):
        mask[i // 32] |= 1 << (i % 32)
    with torch.cuda.device(device):
        ptr = lib.vcr_stream_create_cu_masked(mask, words)
    if not ptr:
        raise RuntimeError("vcr_raster: " + last_error())
    return torch.cuda.ExternalStream(ptr, device=device)


def profile_enable(on=True, stages=None):
    """`stages`: iterable of stage names to time (default all); every timed stage adds an event pair to the stream."""
    lib = load()
    mask = 0xFFFFFFFF if stages is None else sum(1 << STAGES.index(s) for s in stages)
    lib.vcr_profile_select(mask)
    lib.vcr_profile_enable(1 if on else 0)


def profile_read():
    """-> {stage: (total_ms, launches)} accumulated since the previous read (stream must be synchronised)."""
    lib = load()
    n = lib.vcr_profile_num_stages()
    ms = (C.c_float * n)()
    cnt = (C.c_int32 * n)()
    if lib.vcr_profile_read(ms, cnt, n) != 0:
        raise RuntimeError("vcr_raster: " + last_error())
    return {STAGES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}
