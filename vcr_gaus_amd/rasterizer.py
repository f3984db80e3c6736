"""`GaussianRasterizationSettings` / `GaussianRasterizer`: the Python face of the drop-in boundary.

Same names, keyword arguments, return tuples and error behaviour as the reference's un-vendored
extension (call sites `gaussian_renderer/__init__.py:43-59,107-120,332-344,441-453,550-562`);
the arithmetic is in libvcr_raster.so (HIP, gfx950) reached through ctypes.
"""
from typing import NamedTuple, Optional

import ctypes as _ct
import os as _os

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False
    f_count: int = 0          # optional: `render_fast` omits it (gaussian_renderer/__init__.py:183-196)


class _Allocator:
    """Backs the C-side allocation callback with torch's caching allocator (stream-ordered reuse)."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self.scratch = []

        def _cb(_user, tag, nbytes):
            # instance-count dependent sizes change from camera to camera: rounding up to 1/8-octave steps makes the
            # requests repeat, so the caching allocator reuses its blocks instead of growing (hipMalloc stalls)
            n = max(int(nbytes), 1)
            if n > (1 << 20):
                q = 1 << (n.bit_length() - 4)
                n = (n + q - 1) // q * q
            t = torch.empty(n, dtype=torch.uint8, device=self.device)
            if tag == _lib.BUF_SCRATCH:
                self.scratch.append(t)
            else:
                self.bufs[tag] = t
            return t.data_ptr()

        self.cb = _lib.ALLOC_FN(_cb)

    def close(self):
        """Call when the library call has returned.  The callback closes over this object and this object holds the callback:
        a reference cycle, which would keep every scratch tensor of the call (hundreds of MB per step) out of the caching
        allocator until the interpreter's cyclic collector happens to run -- the next steps then find their block sizes taken
        and fall through to `hipMalloc` (single steps of 5-10 ms in a run of 1.3 ms steps, round 4).  Scratch is only needed
        until the call returns (later work on the same stream is ordered behind it), so it goes back right away."""
        self.scratch.clear()
        self.cb = None


def _f32(t):
    return None if t is None else t.detach().contiguous().float()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _check(rc):
    if rc != 0:
        raise RuntimeError("vcr_raster: " + _lib.last_error())


class RasterOptions:
    """Per-call options of the f_count = 0 forward beyond the reference's keyword surface (all default to the reference's
    behaviour).  They travel with the call -- `GaussianRasterizer(settings, options=...)`, `render(..., raster_options=...)`
    -- and the backward of a forward keeps the options it was recorded with; nothing here is process-global.
      sh_grad          "full": the backward writes dL/dshs.  "rgb": it skips the 192 B/Gaussian SH gradient and leaves
                       dL/drgb [N,3] + the unit view directions [N,3] in the call's `RasterRecord` (data-parallel exchange
                       and the two-stream SH update form the SH gradient from those two factors).
      colour_stream    torch.cuda.Stream: SH -> RGB is evaluated there (VcrRasterArgs.colour_stream), joined before compositing.
      colour_hook      callable(): enqueue caller work on `colour_stream` ahead of the colour evaluation.
      colour_sh_update callable() -> (_lib.VcrShUpdate, keep-alive) or None: SH Adam step fused into the colour evaluation.
      sort_stream      torch.cuda.Stream: depth keys + depth sort run there, beside the projection.
      quad_lists       bin the tile instances per 8x8 quad instead of per 16x16 tile (`VcrRasterArgs.quad_lists`): identical
                       results, fewer gathers in the compositing kernels, more sort entries -- pays for small footprints.
      forward_form     "auto" (the library picks per frame from its counts), "uniform" or "two_phase": which kernel composites the
                       frame (`VcrRasterArgs.forward_form`); bit-identical results -- a test / A-B switch.
      tail             an object with `.armed`, `.done` and `.tail()` -> (_lib.VcrGeometryStep, commit) or None (the trainer's
                       `GeometrySink`): when armed, the BACKWARD of this call applies the static tail of the training
                       iteration inside its projection-backward kernel (`vcr_rasterize_backward_tail`), calls `commit()`,
                       sets `.done` and returns no gradient for means3D / means2D / opacities / scales / rotations / normals.
      world_normals    an object whose `.saved[3]` / `.saved[4]` are the camera rotation [3,3] and the aux bytes of this
                       render's fused activation (the trainer's `GeometrySink`): the backward then returns dL/dnormals as the
                       gradient w.r.t. the WORLD-space axis column (`VcrBackwardIO.normals_Rw2c`) -- the form the ranks of a
                       data-parallel step can sum before the one-kernel tail."""
    __slots__ = ("sh_grad", "colour_stream", "colour_hook", "colour_sh_update", "sort_stream", "quad_lists", "tail", "world_normals",
                 "forward_form")

    def __init__(self, sh_grad="full", colour_stream=None, colour_hook=None, colour_sh_update=None, sort_stream=None,
                 quad_lists=False, tail=None, world_normals=None, forward_form="auto"):
        if sh_grad not in ("full", "rgb"):
            raise ValueError("sh_grad must be 'full' or 'rgb'")
        self.sh_grad, self.colour_stream, self.colour_hook = sh_grad, colour_stream, colour_hook
        self.colour_sh_update, self.sort_stream, self.quad_lists = colour_sh_update, sort_stream, bool(quad_lists)
        self.tail = tail
        self.world_normals = world_normals
        if forward_form not in ("auto", "uniform", "two_phase"):
            raise ValueError("forward_form must be 'auto', 'uniform' or 'two_phase'")
        self.forward_form = forward_form


DEFAULT_OPTIONS = RasterOptions()


class RasterRecord:
    """What ONE rasterizer call leaves behind besides its return tuple: the work it did (N Gaussians, V visible, R tile
    instances of the 3-sigma rectangles -- the reference's count --, R' tile instances really emitted after the exact
    per-tile rejection, and with `debug` the longest tile list) and, after the backward of an `sh_grad="rgb"` call, the two factors of its SH
    gradient.  One object per call (`GaussianRasterizer.record`, `render()["raster"]`): a second render -- an evaluation
    pass, another model -- between a backward and the gradient exchange cannot replace them."""
    __slots__ = ("N", "V", "R", "max_tile_len", "emitted", "drgb", "view_dirs", "timing")

    def __init__(self):
        self.N = self.V = self.R = 0
        self.emitted = 0                               # R' <= R: tile instances really emitted (exact per-tile rejection)
        self.max_tile_len = -1                         # (only read back by `debug` renders)
        self.drgb = self.view_dirs = self.timing = None

    def take_sh_factors(self):
        """-> (dL/drgb [N,3], view directions [N,3]) of this call's backward; the record lets go of them."""
        if self.drgb is None:
            raise RuntimeError("vcr_raster: no dL/drgb recorded (the call was not made with sh_grad='rgb', or its backward has not run)")
        out = (self.drgb, self.view_dirs)
        self.drgb = self.view_dirs = None
        return out


def _num_dist_from_env():
    v = _os.environ.get("VCR_NUM_DIST", "0")
    if v not in ("0", "1", "2"):
        raise RuntimeError(f"vcr_raster: VCR_NUM_DIST={v!r}; expected 0, 1 (distortion) or 2 (depth moments)")
    return int(v)


# Trailing output channels of a `GaussianRasterizer(raster_settings=...)` built WITHOUT `num_dist` -- i.e. by the reference's
# unchanged render(), which reads `rendered_out[-1:]` as the distortion map or `[-2:-1]`, `[-1:]` as the depth moments
# (`gaussian_renderer/__init__.py:154-162`).  In the fork this is the compile-time constant `NUM_DIST` of
# cuda_rasterizer/config.h (README.md:152-155: "set NUM_DIST = 1 ... and reinstall"); here it is a run-time setting of the
# module: environment VCR_NUM_DIST at import, or `diff_gaussian_rasterization.set_num_dist(n)`.
#   0  no trailing channel (C = 8 + S)
#   1  depth distortion                          -> cfg.optim.loss_weight.distortion > 0 (the DTU configuration)
#   2  depth moments sum w d, sum w d^2          -> cfg.optim.loss_weight.depth_var > 0
NUM_DIST = _num_dist_from_env()


def set_num_dist(n):
    """Run-time counterpart of editing `NUM_DIST` in the fork's config.h and reinstalling.  Returns the previous value."""
    global NUM_DIST
    if n not in (0, 1, 2):
        raise ValueError("num_dist must be 0, 1 (distortion) or 2 (depth moments)")
    old, NUM_DIST = NUM_DIST, int(n)
    return old


def get_num_dist():
    return NUM_DIST


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, means2D_densify, sh, colors_precomp, normals_precomp,
                semantics_precomp, opacities, scales, rotations, cov3Ds_precomp, dirs, rs, sh_rest=None, num_dist=0,
                opts=None, rec=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)          # (no zero-filled [N] gradient for the non-differentiable radii)
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("vcr_raster: tensors must live on a HIP device (no CPU path exists)")
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        t = dict(means3D=_f32(means3D), shs=_f32(sh), shs_rest=_f32(sh_rest), colors=_f32(colors_precomp), normals=_f32(normals_precomp),
                 sem=_f32(semantics_precomp), opac=_f32(opacities), scales=_f32(scales), rots=_f32(rotations),
                 cov=_f32(cov3Ds_precomp), dirs=_f32(dirs), bg=_f32(rs.bg).to(dev), view=_f32(rs.viewmatrix).to(dev),
                 proj=_f32(rs.projmatrix).to(dev), campos=_f32(rs.campos).to(dev))
        S = 0 if t["sem"] is None else int(t["sem"].shape[1])
        K = 0 if t["shs"] is None else int(t["shs"].shape[1]) + (0 if t["shs_rest"] is None else int(t["shs_rest"].shape[1]))
        a = _lib.VcrRasterArgs(N=N, H=H, W=W, S=S, K=K, sh_degree=int(rs.sh_degree), f_count=int(rs.f_count),
                               num_dist=int(num_dist), debug=int(bool(rs.debug)), tanfovx=float(rs.tanfovx),
                               tanfovy=float(rs.tanfovy), scale_modifier=float(rs.scale_modifier),
                               bg=_ptr(t["bg"]), viewmatrix=_ptr(t["view"]), projmatrix=_ptr(t["proj"]),
                               campos=_ptr(t["campos"]), means3D=_ptr(t["means3D"]), shs=_ptr(t["shs"]),
                               shs_rest=_ptr(t["shs_rest"]), colors_precomp=_ptr(t["colors"]), normals_precomp=_ptr(t["normals"]),
                               semantics_precomp=_ptr(t["sem"]), opacities=_ptr(t["opac"]), scales=_ptr(t["scales"]),
                               rotations=_ptr(t["rots"]), cov3D_precomp=_ptr(t["cov"]), dirs=_ptr(t["dirs"]))
        fc = int(rs.f_count)
        opts = DEFAULT_OPTIONS if opts is None else opts
        rec = RasterRecord() if rec is None else rec
        a.quad_lists = 1 if (opts.quad_lists and max(H, W) <= 8192) else 0
        a.forward_form = {"auto": 0, "uniform": 1, "two_phase": 2}[opts.forward_form]
        hook = upd = None
        if opts.sort_stream is not None and fc == 0 and N > 0:
            a.sort_stream = opts.sort_stream.cuda_stream
        if opts.colour_stream is not None and fc == 0 and t["shs"] is not None:
            a.colour_stream = opts.colour_stream.cuda_stream
            if opts.colour_sh_update is not None and t["shs_rest"] is not None and N > 0:
                upd = opts.colour_sh_update()                     # (struct, tensors kept alive until the call returns)
                if upd is not None:
                    a.sh_update = _ct.addressof(upd[0])
            if opts.colour_hook is not None:
                fn = opts.colour_hook
                hook = _lib.HOOK_FN(lambda _user: fn())            # kept alive until the forward call returns
                a.colour_stream_hook = _ct.cast(hook, _ct.c_void_p)
        C = 8 + S + int(num_dist)
        out = torch.empty((C if fc == 0 else 3, H, W), dtype=torch.float32, device=dev) if fc not in (3, 4) else None
        radii = torch.empty(N, dtype=torch.int32, device=dev)      # fully written by the preprocess kernel
        count = torch.zeros(N, dtype=torch.int32, device=dev) if fc != 0 else None
        score = torch.zeros(N, dtype=torch.float32, device=dev) if fc in (1, 2) else None
        fo = _lib.VcrForwardOut(out=_ptr(out), radii=_ptr(radii), count=_ptr(count), score=_ptr(score))
        al = _Allocator(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            with torch.cuda.device(dev):
                _check(lib.vcr_rasterize_forward(a, fo, al.cb, None, stream))
        finally:
            al.close()
        rec.R, rec.V, rec.N, rec.max_tile_len = int(fo.num_rendered), int(fo.num_visible), N, int(fo.max_tile_len)
        rec.emitted = int(fo.num_emitted)
        if fc == 0:
            ctx.rs, ctx.args_t, ctx.state, ctx.rec = rs, t, al.bufs, rec
            ctx.num_rendered = int(fo.num_rendered)
            ctx.num_emitted = int(fo.num_emitted)
            ctx.has = (means2D_densify is not None)
            ctx.num_dist = int(num_dist)
            ctx.quad_lists = int(a.quad_lists)
            ctx.rgb_mode = opts.sh_grad == "rgb" and t["shs"] is not None
            ctx.tail = opts.tail
            ctx.world_normals = opts.world_normals
            ctx.save_for_backward(radii)
            ctx.mark_non_differentiable(radii)
            return out, radii
        if fc in (1, 2):                                   # forward-only modes: nothing to differentiate
            ctx.mark_non_differentiable(count, score, out, radii)
            return count, score, out, radii
        ctx.mark_non_differentiable(count, radii)
        return count, radii

    @staticmethod
    def backward(ctx, grad_out, _grad_radii=None):
        if grad_out is None:
            return (None,) * 17
        lib = _lib.load()
        rs, t = ctx.rs, ctx.args_t
        (radii,) = ctx.saved_tensors
        dev = radii.device
        N = t["means3D"].shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        S = 0 if t["sem"] is None else int(t["sem"].shape[1])
        K = 0 if t["shs"] is None else int(t["shs"].shape[1]) + (0 if t["shs_rest"] is None else int(t["shs_rest"].shape[1]))
        a = _lib.VcrRasterArgs(N=N, H=H, W=W, S=S, K=K, sh_degree=int(rs.sh_degree), f_count=0, num_dist=ctx.num_dist,
                               debug=int(bool(rs.debug)), tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
                               scale_modifier=float(rs.scale_modifier), bg=_ptr(t["bg"]), viewmatrix=_ptr(t["view"]),
                               projmatrix=_ptr(t["proj"]), campos=_ptr(t["campos"]), means3D=_ptr(t["means3D"]),
                               shs=_ptr(t["shs"]), shs_rest=_ptr(t["shs_rest"]), colors_precomp=_ptr(t["colors"]), normals_precomp=_ptr(t["normals"]),
                               semantics_precomp=_ptr(t["sem"]), opacities=_ptr(t["opac"]), scales=_ptr(t["scales"]),
                               rotations=_ptr(t["rots"]), cov3D_precomp=_ptr(t["cov"]), dirs=_ptr(t["dirs"]))
        a.quad_lists = ctx.quad_lists                 # (the state buffers hold the lists in the form the forward chose)
        g = grad_out.contiguous().float()

        def new(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        rgb_mode = ctx.rgb_mode
        # the static tail of the iteration inside this backward: gradients w.r.t. the geometry stay in registers
        sink, tail = ctx.tail, None
        if sink is not None and sink.armed and not sink.done and N > 0 and t["scales"] is not None and t["cov"] is None \
                and (rgb_mode or t["shs"] is None):
            tail = sink.tail()
        if tail is not None:
            d_means3D = d_means2D = d_opac = d_dens = None
        else:
            d_means3D, d_means2D, d_opac = new(N, 3), new(N, 3), new(N, 1)
            d_dens = new(N, 3) if ctx.has else None
        d_shs = new(*t["shs"].shape) if (t["shs"] is not None and not rgb_mode) else None
        d_shr = new(*t["shs_rest"].shape) if (t["shs_rest"] is not None and not rgb_mode) else None
        d_rgb = new(N, 3) if rgb_mode else None
        v_dirs = new(N, 3) if rgb_mode else None
        d_col = new(N, 3) if t["colors"] is not None else None
        d_nrm = new(N, 3) if (t["normals"] is not None and tail is None) else None
        d_sem = new(N, S) if t["sem"] is not None else None
        d_sc = new(N, 3) if (t["scales"] is not None and tail is None) else None
        d_rot = new(N, 4) if (t["rots"] is not None and tail is None) else None
        d_cov = new(N, 6) if t["cov"] is not None else None
        io = _lib.VcrBackwardIO(dL_dout=_ptr(g), geom=_ptr(ctx.state[_lib.BUF_GEOM]),
                                binning=_ptr(ctx.state[_lib.BUF_BINNING]), image=_ptr(ctx.state[_lib.BUF_IMAGE]),
                                radii=_ptr(radii), num_rendered=ctx.num_rendered, num_emitted=ctx.num_emitted, dL_dmeans3D=_ptr(d_means3D),
                                dL_dmeans2D=_ptr(d_means2D),
                                dL_dmeans2D_densify=_ptr(d_dens) if tail is None else (_ptr(radii) if ctx.has else None),   # (tail: a flag)
                                dL_dshs=_ptr(d_shs),
                                dL_dshs_rest=_ptr(d_shr), dL_drgb=_ptr(d_rgb), view_dirs=_ptr(v_dirs), dL_dcolors=_ptr(d_col), dL_dnormals=_ptr(d_nrm), dL_dsemantics=_ptr(d_sem),
                                dL_dopacities=_ptr(d_opac), dL_dscales=_ptr(d_sc), dL_drotations=_ptr(d_rot),
                                dL_dcov3D=_ptr(d_cov))
        wn = ctx.world_normals
        if wn is not None and tail is None and d_nrm is not None:
            if wn.saved is None:
                raise RuntimeError("vcr_raster: world_normals was requested but the fused activation left nothing in the sink")
            io.normals_Rw2c, io.normals_aux = _ptr(wn.saved[3]), _ptr(wn.saved[4])
        al = _Allocator(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            with torch.cuda.device(dev):
                if tail is not None:
                    sink.started = True             # (from here on the geometry groups may have been stepped: GeometrySink)
                    _check(lib.vcr_rasterize_backward_tail(a, io, tail[0], al.cb, None, stream))
                    tail[1]()
                    sink.done = True
                else:
                    _check(lib.vcr_rasterize_backward(a, io, al.cb, None, stream))
        finally:
            al.close()
        if rgb_mode:
            ctx.rec.drgb, ctx.rec.view_dirs = d_rgb, v_dirs
        return (d_means3D, d_means2D, d_dens, d_shs, d_col, d_nrm, d_sem, d_opac, d_sc, d_rot, d_cov, None, None, d_shr, None,
                None, None)


# Defaults of the batched visibility passes: cameras in flight (0 = the library's 8) and 8x8-cell lists.  Round 6 measured both knobs
# on the 200 virtual cameras of a densification at 1 M Gaussians (profiles/r6_visibility_ab.txt): 8 / 16 / 4 in flight 59.9 / 59.6 /
# 66.5 ms per batch, cell lists 69.6 ms -- the defaults stay.
VIS_INFLIGHT = 0
VIS_QUAD_LISTS = 0


@torch.no_grad()
def visibility_batch(viewmatrices, projmatrices, campos, tanfovx, tanfovy, image_height, image_width, means3D, opacities,
                     scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, flags_only=False, count=None,
                     inflight=0):
    """The visibility passes of one densification step as ONE library call (`vcr_visibility_batch`): B cameras of the same
    resolution, per-Gaussian counters accumulated on the device.  Equivalent to B `GaussianRasterizer` calls with
    `f_count=3` whose `countlist`s are summed (`tools/prune.py:51-69`), without SH -> RGB, without an image and without a
    host round trip per camera.
      viewmatrices / projmatrices [B,4,4], campos [B,3]: `world_view_transform`, `full_proj_transform`, `camera_center` stacked;
      tanfovx / tanfovy: sequences of B floats;  count: int32 [N] to accumulate into (default: a new zero tensor);
      flags_only: count[i] = 1 where the summed count would be > 0 (all that `get_visi_list` reads) instead of the sum.
    -> (count [N] int32, num_rendered [B] (3-sigma tile instances per camera), num_visible [B])"""
    lib = _lib.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("vcr_raster: tensors must live on a HIP device (no CPU path exists)")
    N, B = int(means3D.shape[0]), int(viewmatrices.shape[0])
    if count is None:
        count = torch.zeros(N, dtype=torch.int32, device=dev)
    if count.dtype != torch.int32 or count.shape != (N,) or not count.is_contiguous():
        raise ValueError("count must be a contiguous int32 tensor of shape [N]")
    if len(tanfovx) != B or len(tanfovy) != B or projmatrices.shape[0] != B or campos.shape[0] != B:
        raise ValueError("one view / projection matrix, camera centre and tan(fov) pair per camera")
    t = [_f32(x) for x in (viewmatrices, projmatrices, campos, means3D, opacities, scales, rotations, cov3D_precomp)]
    t = [None if x is None else x.to(dev) for x in t]
    tx, ty = (_ct.c_float * max(B, 1))(*map(float, tanfovx)), (_ct.c_float * max(B, 1))(*map(float, tanfovy))
    nr, nv = (_ct.c_int64 * max(B, 1))(), (_ct.c_int32 * max(B, 1))()
    a = _lib.VcrVisibilityBatch(N=N, H=int(image_height), W=int(image_width), B=B, flags_only=int(bool(flags_only)),
                                inflight=int(inflight) or VIS_INFLIGHT, quad_lists=VIS_QUAD_LISTS, scale_modifier=float(scale_modifier), tanfovx=tx, tanfovy=ty,
                                viewmatrix=_ptr(t[0]), projmatrix=_ptr(t[1]), campos=_ptr(t[2]), means3D=_ptr(t[3]),
                                opacities=_ptr(t[4]), scales=_ptr(t[5]), rotations=_ptr(t[6]), cov3D_precomp=_ptr(t[7]),
                                count=count.data_ptr(), num_rendered=nr, num_visible=nv)
    al = _Allocator(dev)
    try:
        with torch.cuda.device(dev):
            _check(lib.vcr_visibility_batch(a, al.cb, None, torch.cuda.current_stream(dev).cuda_stream))
    finally:
        al.close()
    return count, list(nr)[:B], list(nv)[:B]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings, num_dist=None, options=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.num_dist = NUM_DIST if num_dist is None else num_dist
        self.options = options            # RasterOptions or None (the reference's behaviour)
        self.record = None                # RasterRecord of the most recent call of THIS object

    def forward(self, means3D, means2D, opacities, means2D_densify=None, shs=None, colors_precomp=None,
                normals_precomp=None, semantics_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                dirs=None, inside=None, shs_rest=None):
        """`shs_rest` (extension): pass the reference's split SH storage as shs=_features_dc [N,1,3],
        shs_rest=_features_rest [N,15,3] and skip the torch.cat of `get_features`."""
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        self.record = RasterRecord()
        return _RasterizeGaussians.apply(means3D, means2D, means2D_densify, shs, colors_precomp, normals_precomp,
                                         semantics_precomp, opacities, scales, rotations, cov3D_precomp, dirs, rs, shs_rest,
                                         self.num_dist if rs.f_count == 0 else 0, self.options, self.record)
