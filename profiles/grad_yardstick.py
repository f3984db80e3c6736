"""The yardstick of the gradient tolerances in tests/util.py: the error of the ORACLE ITSELF evaluated in fp32 against its
fp64 self, per tensor, in the three regimes the GPU tests compare gradients in:

  small  the parity cases of tests/test_raster_parity_gpu.py (3 000 - 10 000 Gaussians, <= 256 x 256)
  full   the full-size sampled-tile cases of tests/test_fullsize_sampled_gpu.py (every pixel sees ~10x more pairs)
  step   one whole training iteration (tests/test_train_step_gpu.py: the image losses add their own fp32 rounding)

A HIP gradient may be at most `FACTOR` (3) times as noisy as a plain fp32 evaluation of the same algorithm.  CPU only:
    python profiles/grad_yardstick.py [small] [full] [step]   ->  tests/golden/grad_yardstick.json (merged)
"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model_torch as OM  # noqa: E402
from oracle import raster_torch as OR  # noqa: E402
from oracle import trainer_torch as OT  # noqa: E402
from tests import util  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "grad_yardstick.json")
KEYS = ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d", "sem"]


def merge(acc, name, st):
    cur = acc.setdefault(name, [0.0, 0.0, 0.0])
    for i, k in enumerate(("maxnorm", "p99", "p999")):
        cur[i] = max(cur[i], float(st[k]))


def small():
    import tests.test_raster_parity_gpu as T
    acc = {}
    for case in T.CASES:
        n, W, H, f, sm, sem = case
        for nd in (0, 2):
            cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
            bg = torch.tensor([0.2, 0.1, 0.4])
            g = torch.Generator().manual_seed(11)
            (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True, num_dist=nd)
            wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
            (ref * wgt).sum().backward()
            (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True, num_dist=nd)
            (o32 * wgt.float()).sum().backward()
            for k in KEYS:
                if rl[k] is not None:
                    merge(acc, k, util.grad_stats(l32[k].grad, rl[k].grad))
            print("small", case, nd, flush=True)
    return acc


def full(workloads=("c2_dtu_300k_800x600", "metric_1m_1080p", "c4_tnt_2m_1080p")):
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    import tests.test_fullsize_sampled_gpu as F
    acc = {}
    for wl, view, stride, _ in F.CASES:
        if wl not in workloads:
            continue
        n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
        raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
        cam = synthetic.make_cameras(8, W, H, focal)[view]
        act = OM.activations(raw)
        ncam = OM.camera_normals(OM.get_normal(act["rotation"], act["scaling"]), act["xyz"], cam.camera_center, cam.R_w2c)
        inp = dict(means3D=act["xyz"], shs=act["shs"], normals=ncam.contiguous(), opac=act["opacity"], scales=act["scaling"],
                   rots=act["rotation"], sem=raw["obj_dc"].squeeze(1).contiguous() if sem else None)
        dirs = get_all_px_dir(cam.intr, H, W)
        bg = torch.tensor([0.15, 0.05, 0.3])
        s = util.settings_for(cam, bg, OR.Settings)
        with torch.no_grad():
            pre = OR.preprocess(s, inp["means3D"], torch.zeros(n, 3), inp["shs"], None, inp["normals"], inp["sem"], inp["opac"],
                                inp["scales"], inp["rots"], None)
        gx, gy = pre["grid"]
        hit = torch.zeros(n, dtype=torch.bool)
        tmask = torch.zeros(gy * 16, gx * 16, dtype=torch.bool)
        for t in range(0, gx * gy, stride):
            x, y = t % gx, t // gx
            hit |= pre["vis"] & (pre["xmin"] <= x) & (x < pre["xmax"]) & (pre["ymin"] <= y) & (y < pre["ymax"])
            tmask[y * 16:(y + 1) * 16, x * 16:(x + 1) * 16] = True
        tmask = tmask[:H, :W]
        spx, spy, scon, sop = pre["px"][hit], pre["py"][hit], pre["conic"][hit], pre["opacity"][hit]
        del pre
        sub = {k: (None if v is None else v[hit]) for k, v in inp.items()}
        (ref, _, _), rl = util.oracle_forward(cam, sub, dirs, bg, dtype=torch.float64, requires_grad=True, tile_stride=stride)
        (o32, _, _), l32 = util.oracle_forward(cam, sub, dirs, bg, dtype=torch.float32, requires_grad=True, tile_stride=stride)
        g = torch.Generator().manual_seed(stride)
        wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64) * tmask[None]
        (ref * wgt).sum().backward()
        (o32 * wgt.float()).sum().backward()
        # as the test does: Gaussians under a pixel whose hit decision flipped (here: between the two oracle precisions) are left out
        o, r = o32.detach().double()[:, tmask], ref.detach()[:, tmask]
        badmask = ((o - r).abs() > 2e-4 + 1e-4 * r.abs()).any(0)
        ys, xs = torch.nonzero(tmask, as_tuple=True)
        clean = torch.ones(int(hit.sum()), dtype=torch.bool)
        for y, x in zip(ys[badmask].tolist(), xs[badmask].tolist()):
            dx, dy = spx - x, spy - y
            power = -0.5 * (scon[:, 0] * dx * dx + scon[:, 2] * dy * dy) - scon[:, 1] * dx * dy
            clean &= ~((power <= 0) & (sop * torch.exp(power) >= 0.5 / 255.0))
        for k in KEYS:
            if rl.get(k) is not None:
                a, b = l32[k].grad[clean], rl[k].grad[clean]
                if k == "m2d":
                    a, b = a[:, :2], b[:, :2]
                merge(acc, k, util.grad_stats(a, b))
        print("full", wl, "flipped", int(badmask.sum()), "of", int(tmask.sum()), flush=True)
    return acc


def step():
    """Whole iterations of oracle/trainer_torch.py in fp32 against fp64 on the scene of tests/test_train_step_gpu.py (ground
    truth: oracle renders of the jittered copy -- its exact content does not matter for the rounding level)."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    acc = {}
    raw = synthetic.make_gaussians(3000, seed=5)
    raw["scaling"] = raw["scaling"] + 1.8
    cams = synthetic.make_cameras(3, 96, 64, 80.0)
    extent = synthetic.cameras_extent(cams)
    dirs = get_all_px_dir(cams[0].intr, 64, 96)
    g = torch.Generator().manual_seed(1)
    raw2 = {k: v.clone() for k, v in raw.items()}
    raw2["xyz"] = raw2["xyz"] + 0.3 * 0.1 * torch.randn(raw["xyz"].shape, generator=g)
    raw2["f_dc"] = raw2["f_dc"] + 0.3 * 5 * torch.randn(raw["f_dc"].shape, generator=g)
    trans, scale = torch.zeros(3), torch.ones(3)
    for preset, it, over in [("dtu_c3", 1, {}), ("tnt", 1, {}), ("360", 1, {}), ("dtu", 15001, {}),
                             ("dtu_c3", 3, {"loss_weight": {"distortion": 100.0}})]:
        cfg = make_config(preset, optim=dict(over, densify_from_iter=10 ** 9, prune={"iterations": []}))
        cfg.optim.loss_weight.semantic = 0.0
        for cam in cams:
            with torch.no_grad():
                pkg = OT.render({k: v.float() for k, v in raw2.items()}, cam, cfg, extent, torch.zeros(3), dirs, 3)
            cam.original_image = pkg["render"].clamp(0, 1).contiguous()
            cam.normal = pkg["est_normal"].contiguous()
        bg = torch.tensor([0.3, 0.6, 0.1]) if cfg.optim.random_background else torch.zeros(3)
        cam = cams[1]
        r64 = OT.step(raw, cam, cfg, extent, bg, dirs, it, 3, trans, scale, extent, dtype=torch.float64)
        r32 = OT.step(raw, cam, cfg, extent, bg, dirs, it, 3, trans, scale, extent, dtype=torch.float32)
        for k in OT.GROUPS:
            merge(acc, k, util.grad_stats(r32["grads"][k], r64["grads"][k]))
        merge(acc, "means2D_densify", util.grad_stats(r32["densify_grad"][:, :2], r64["densify_grad"][:, :2]))
        print("step", preset, it, {k: f"{r64['losses'][k]:.4g}" for k in r64["losses"]}, flush=True)
    return acc


if __name__ == "__main__":
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    todo = sys.argv[1:] or ["small", "full", "step"]
    cur = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in todo:
        cur[name] = {"small": small, "full": full, "step": step}[name]()
        with open(OUT, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
        print("wrote", name, cur[name], flush=True)
