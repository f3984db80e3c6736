"""Shared helpers for the parity tests (seeded inputs, oracle drivers, comparison metrics)."""
import math

import torch

from oracle import model_torch as OM
from oracle import raster_torch as OR
from vcr_gaus_amd import synthetic
from vcr_gaus_amd.graphics_utils import get_all_px_dir


def make_case(n, width, height, focal, seed=0, scale_mult=1.0, view=0, n_views=3, sem=0, sh_degree=3):
    raw = synthetic.make_gaussians(n, seed=seed, sem_channels=sem)
    raw["scaling"] = raw["scaling"] + math.log(scale_mult)
    cam = synthetic.make_cameras(n_views, width, height, focal)[view]
    act = OM.activations(raw)
    nw = OM.get_normal(act["rotation"], act["scaling"])
    ncam = OM.camera_normals(nw, act["xyz"], cam.camera_center, cam.R_w2c)
    inputs = dict(means3D=act["xyz"], shs=act["shs"], normals=ncam.contiguous(), opac=act["opacity"],
                  scales=act["scaling"], rots=act["rotation"],
                  sem=raw["obj_dc"].squeeze(1).contiguous() if sem else None)
    dirs = get_all_px_dir(cam.intr, height, width)
    return cam, inputs, dirs


def settings_for(cam, bg, cls, sh_degree=3, f_count=0, device=None):
    mv = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    return cls(image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
               tanfovy=math.tan(cam.FoVy * 0.5), bg=mv(bg), scale_modifier=1.0, viewmatrix=mv(cam.world_view_transform),
               projmatrix=mv(cam.full_proj_transform), sh_degree=sh_degree, campos=mv(cam.camera_center),
               prefiltered=False, debug=False, f_count=f_count)


def oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=False, sh_degree=3, f_count=0,
                   use_normals=True, num_dist=0, tile_stride=1):
    s = settings_for(cam, bg, OR.Settings, sh_degree=sh_degree, f_count=f_count)
    leaf = {}
    for k, v in inp.items():
        if v is None:
            leaf[k] = None
        else:
            leaf[k] = v.detach().to(dtype).clone().requires_grad_(requires_grad)
    N = inp["means3D"].shape[0]
    leaf["m2"] = torch.zeros(N, 3, dtype=dtype, requires_grad=requires_grad)
    leaf["m2d"] = torch.zeros(N, 3, dtype=dtype, requires_grad=requires_grad)
    res = OR.rasterize(s, leaf["means3D"], leaf["m2"], leaf["m2d"], leaf["shs"], None,
                       leaf["normals"] if use_normals else None, leaf["sem"], leaf["opac"], leaf["scales"],
                       leaf["rots"], None, dirs if use_normals else None, num_dist=num_dist, tile_stride=tile_stride)
    return res, leaf


def hip_forward(cam, inp, dirs, bg, device, requires_grad=False, sh_degree=3, f_count=0, use_normals=True, num_dist=0):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    s = settings_for(cam, bg, GaussianRasterizationSettings, sh_degree=sh_degree, f_count=f_count, device=device)
    leaf = {}
    for k, v in inp.items():
        leaf[k] = None if v is None else v.detach().float().to(device).clone().requires_grad_(requires_grad)
    N = inp["means3D"].shape[0]
    leaf["m2"] = torch.zeros(N, 3, device=device, requires_grad=requires_grad)
    leaf["m2d"] = torch.zeros(N, 3, device=device, requires_grad=requires_grad)
    rast = GaussianRasterizer(raster_settings=s, num_dist=num_dist)
    res = rast(means3D=leaf["means3D"], means2D=leaf["m2"], means2D_densify=leaf["m2d"] if f_count == 0 else None,
               shs=leaf["shs"], colors_precomp=None, normals_precomp=leaf["normals"] if use_normals else None,
               semantics_precomp=leaf["sem"], opacities=leaf["opac"], scales=leaf["scales"], rotations=leaf["rots"],
               cov3D_precomp=None, dirs=dirs.to(device) if (use_normals and dirs is not None) else None, inside=None)
    return res, leaf


def frac_bad(a, b, rtol, atol):
    """fraction of elements with |a-b| > atol + rtol*|b|"""
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).abs() > atol + rtol * b.abs()).double().mean())


def bad_pixels(a, b, rtol=1e-4, atol=2e-4):
    """Number of pixels of a [C,H,W] image with any channel outside |a-b| <= atol + rtol*|b|.  The alpha >= 1/255 and
    T < 1e-4 cut-offs are discontinuous, so an fp32-vs-fp64 rounding flip moves ALL channels of that pixel at once;
    parity is therefore asserted per pixel: at most max(4, 1e-3 * H*W) flipped pixels (every pixel sees ~100
    (pixel, Gaussian) pairs, each a potential flip)."""
    a, b = a.double().cpu(), b.double().cpu()
    bad = ((a - b).abs() > atol + rtol * b.abs()).reshape(a.shape[0], -1).any(0)
    return int(bad.sum())


def pixel_budget(img):
    return max(4, int(1e-3 * img.shape[-1] * img.shape[-2]))


def rel_err(a, b):
    """max-norm relative error of a tensor against its reference."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_err(a, b, floor_frac=1e-3):
    """Element-wise relative error |a-b| / (|b| + floor), floor = floor_frac * rms(b over its non-zero entries): small
    gradients are checked against their own magnitude (the max-norm figure of `rel_err` is dominated by the largest
    entries), while entries that are sums of cancelling terms are not divided by ~0."""
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    nz = b[b != 0]
    rms = float(nz.square().mean().sqrt()) if nz.numel() else 1.0
    return (a - b).abs() / (b.abs() + floor_frac * rms)


def grad_stats(a, b):
    """-> dict(maxnorm, med, p99, p999, max) of the gradient error (max-norm relative + element-wise quantiles)."""
    e = elem_err(a, b)
    q = torch.quantile(e, torch.tensor([0.5, 0.99, 0.999], dtype=e.dtype)) if e.numel() else torch.zeros(3)
    return dict(maxnorm=rel_err(a, b), med=float(q[0]), p99=float(q[1]), p999=float(q[2]), max=float(e.max()) if e.numel() else 0.0)


# Gradient acceptance used by every parity test: max-norm relative error < 1e-3 AND element-wise (abs+rel, see elem_err)
# error < 1e-3 on 99 % and < 1e-2 on 99.9 % of the entries.  (Max-norm: measured 6e-7 .. 5.7e-4 over the parity cases; the
# largest values belong to a Gaussian under a flipped pixel and move with the order of the fp32 atomics, which depends on
# how the tiles are launched, hence 1e-3 rather than the 5e-4 of round 1.)  (A pixel whose alpha >= 1/255 or T < 1e-4 decision flips between fp32 and fp64
# moves the gradients of the few Gaussians under it discretely -- those are the tolerated 0.1 %; the measured table is
# committed as profiles/r2_grad_error_table.txt.)
GRAD_MAXNORM_TOL = 1e-3
GRAD_ELEM_P99_TOL = 1e-3
GRAD_ELEM_P999_TOL = 1e-2


def assert_grads_close(got, ref, name, maxnorm_tol=GRAD_MAXNORM_TOL, p999_tol=GRAD_ELEM_P999_TOL, p99_tol=None):
    st = grad_stats(got, ref)
    p99_tol = GRAD_ELEM_P99_TOL * (p999_tol / GRAD_ELEM_P999_TOL) if p99_tol is None else p99_tol
    assert st["maxnorm"] < maxnorm_tol, f"grad {name}: max-norm rel err {st['maxnorm']:.2e} (stats {st})"
    assert st["p99"] < p99_tol, f"grad {name}: element-wise p99 err {st['p99']:.2e} (stats {st})"
    assert st["p999"] < p999_tol, f"grad {name}: element-wise p99.9 err {st['p999']:.2e} (stats {st})"
    return st


def flip_clean_mask(cam, inp, out, ref, bg, sh_degree=3):
    """Gaussians that do NOT contribute (alpha >= 0.5/255) at a pixel whose forward value differs beyond the pixel
    tolerance, i.e. at a pixel where an alpha >= 1/255 / T < 1e-4 decision flipped between fp32 and fp64.  The gradient of
    the few Gaussians under such a pixel moves discretely with the flip; gradient comparisons are made on the rest.
    -> (bool [N] mask, number of flipped pixels)."""
    a, b = out.detach().double().cpu(), ref.detach().double().cpu()
    bad = ((a - b).abs() > 2e-4 + 1e-4 * b.abs()).any(0)
    n = inp["means3D"].shape[0]
    clean = torch.ones(n, dtype=torch.bool)
    ys, xs = torch.nonzero(bad, as_tuple=True)
    if ys.numel():
        s = settings_for(cam, bg, OR.Settings, sh_degree=sh_degree)
        with torch.no_grad():
            pre = OR.preprocess(s, inp["means3D"].float(), torch.zeros(n, 3), inp["shs"].float(), None, None, None,
                                inp["opac"].float(), inp["scales"].float(), inp["rots"].float(), None)
        for y, x in zip(ys.tolist(), xs.tolist()):
            dx, dy = pre["px"] - x, pre["py"] - y
            power = -0.5 * (pre["conic"][:, 0] * dx * dx + pre["conic"][:, 2] * dy * dy) - pre["conic"][:, 1] * dx * dy
            clean &= ~(pre["vis"] & (power <= 0) & (pre["opacity"] * torch.exp(power) >= 0.5 / 255.0))
    return clean, int(bad.sum())
