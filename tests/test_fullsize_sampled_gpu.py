"""Oracle parity AT FULL SIZE on sampled tiles, for every BASELINE.json configuration that runs on one GPU:
c2 (300 k / 800x600), the metric configuration (1 M / 1080p), a c4 shard (2 M / 1080p / 2 semantic channels: what one
rank of the 8-GPU job renders) and a c5 shard (5 M / 1600x1200).

The HIP rasterizer renders the WHOLE scene; the fp64 oracle composites every `stride`-th tile from just the Gaussians
that touch those tiles (selected with the oracle's own fp32 projection, so the subset is exact: a tile's list and its
(depth, index) order are unchanged by dropping Gaussians that do not touch it).  Compared: the pixels of the sampled tiles
(forward), radii of the subset, and the gradients of a random loss restricted to those tiles -- for the subset against the
oracle's autograd, and exactly zero for every other Gaussian.

Tolerances at this size: every pixel of these scenes tests ~10x more (pixel, Gaussian) pairs against the discontinuous
alpha >= 1/255 / T < 1e-4 cut-offs than the small parity cases do, so the flipped-pixel budget is 2e-3 of the sampled
pixels (1e-3 there), and the Gaussians that contribute (alpha >= 0.5/255) at a flipped pixel (their gradient moves
discretely with the flip; < 5 % of the subset, asserted) are left out of the gradient comparison; everything else meets
the standard gradient tolerance of tests/util.py."""
import math

import pytest
import torch

from oracle import model_torch as OM
from oracle import raster_torch as OR
from tests import util

pytestmark = pytest.mark.gpu

# workload, camera index, tile stride (prime, so the samples walk across the image), minimum tile instances sampled
CASES = [("c2_dtu_300k_800x600", 1, 37, 3000), ("metric_1m_1080p", 2, 61, 8000), ("c4_tnt_2m_1080p", 0, 83, 8000),
         ("c5_360_5m_1600x1200", 3, 131, 8000)]


@pytest.mark.parametrize("wl,view,stride,min_inst", CASES)
def test_sampled_tiles_match_oracle_at_full_size(device, wl, view, stride, min_inst):
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
    raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
    cam = synthetic.make_cameras(8, W, H, focal)[view]
    act = OM.activations(raw)
    ncam = OM.camera_normals(OM.get_normal(act["rotation"], act["scaling"]), act["xyz"], cam.camera_center, cam.R_w2c)
    inp = dict(means3D=act["xyz"], shs=act["shs"], normals=ncam.contiguous(), opac=act["opacity"], scales=act["scaling"],
               rots=act["rotation"], sem=raw["obj_dc"].squeeze(1).contiguous() if sem else None)
    dirs = get_all_px_dir(cam.intr, H, W)
    bg = torch.tensor([0.15, 0.05, 0.3])
    # ---- HIP: the whole scene
    (out, radii), hl = util.hip_forward(cam, inp, dirs, bg, device, requires_grad=True)
    # ---- which Gaussians touch the sampled tiles (oracle projection, fp32, no autograd)
    s = util.settings_for(cam, bg, OR.Settings)
    with torch.no_grad():
        pre = OR.preprocess(s, inp["means3D"], torch.zeros(n, 3), inp["shs"], None, inp["normals"], inp["sem"], inp["opac"],
                            inp["scales"], inp["rots"], None)
    gx, gy = pre["grid"]
    tiles = torch.arange(0, gx * gy, stride)
    hit = torch.zeros(n, dtype=torch.bool)
    tmask = torch.zeros(gy * 16, gx * 16, dtype=torch.bool)
    inst = 0
    for t in tiles.tolist():
        x, y = t % gx, t // gx
        h = pre["vis"] & (pre["xmin"] <= x) & (x < pre["xmax"]) & (pre["ymin"] <= y) & (y < pre["ymax"])
        inst += int(h.sum())
        hit |= h
        tmask[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16] = True
    tmask = tmask[:H, :W]
    assert inst >= min_inst, f"sample too light: {inst} tile instances"
    spx, spy, scon, sop = pre["px"][hit], pre["py"][hit], pre["conic"][hit], pre["opacity"][hit]
    del pre
    sub = {k: (None if v is None else v[hit]) for k, v in inp.items()}
    # ---- oracle (fp64) on the subset, sampled tiles only
    (ref, rradii, _), rl = util.oracle_forward(cam, sub, dirs, bg, dtype=torch.float64, requires_grad=True, tile_stride=stride)
    mism = float((radii.cpu()[hit] != rradii).double().mean())
    assert mism < 1e-4, f"radii differ for {mism:.2e} of the subset"
    # forward on the sampled pixels
    o, r = out.detach().cpu().double()[:, tmask], ref.detach()[:, tmask]
    badmask = ((o - r).abs() > 2e-4 + 1e-4 * r.abs()).any(0)
    bad = int(badmask.sum())
    assert bad <= max(4, int(2e-3 * o.shape[1])), f"{bad} of {o.shape[1]} sampled pixels differ"
    ys, xs = torch.nonzero(tmask, as_tuple=True)
    clean = torch.ones(int(hit.sum()), dtype=torch.bool)          # Gaussians of the subset not covering a flipped pixel
    for y, x in zip(ys[badmask].tolist(), xs[badmask].tolist()):
        dx, dy = spx - x, spy - y
        power = -0.5 * (scon[:, 0] * dx * dx + scon[:, 2] * dy * dy) - scon[:, 1] * dx * dy
        clean &= ~((power <= 0) & (sop * torch.exp(power) >= 0.5 / 255.0))       # contributes (or nearly does) at that pixel
    assert float((~clean).double().mean()) < 0.05
    assert float(ref[7][tmask].max()) > 0.5                     # the sample sees real coverage
    # backward of a random loss restricted to the sampled tiles
    g = torch.Generator().manual_seed(stride)
    wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64) * tmask[None]
    (ref * wgt).sum().backward()
    (out * wgt.float().to(device)).sum().backward()
    for k in ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "sem"]:
        if rl.get(k) is None or hl[k] is None:
            continue
        gfull = hl[k].grad.cpu()
        assert float(gfull[~hit].abs().max()) == 0.0 if (~hit).any() else True, f"{k}: gradient outside the sampled subset"
        util.assert_grads_close(gfull[hit][clean], rl[k].grad[clean], f"{wl}:{k}")
    util.assert_grads_close(hl["m2d"].grad.cpu()[hit][clean][:, :2], rl["m2d"].grad[clean][:, :2], f"{wl}:m2d")
