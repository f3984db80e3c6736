"""How far apart do two 14-step trajectories of tests/test_ops_gpu.py::test_two_stream_sh_path_trains_like_the_serial_loop land
(fraction of entries beyond the test's tolerance), for: the same configuration run twice (atomics' order only), serial against
two-stream with the separate tail kernel, serial against two-stream with the tail inside the rasterizer's backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import make_synthetic_trainer  # noqa: E402

dev = torch.device("cuda:0")
raw = synthetic.make_gaussians(8000, seed=5)
raw["scaling"] = raw["scaling"] + 1.0
KEYS = ["_features_dc", "_features_rest", "_xyz", "_opacity", "_scaling", "_rotation"]


def run(overlap, raster_tail, steps=14, hook=False):
    cams = synthetic.make_cameras(3, 128, 96, 110.0, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, preset="tnt", overlap_sh=overlap, overlap_min_gaussians=0,
                                optim={"densify_from_iter": 10 ** 9, "opacity_reset_interval": 9})
    tr.fuse_raster_tail = raster_tail
    tr.model.active_sh_degree = 2
    if hook:
        from vcr_gaus_amd.gaussian_renderer import render
        from vcr_gaus_amd.rasterizer import RasterOptions
        real_exchange = tr._exchange_grads

        def exchange_after_an_eval_render(ov, surgery, rec):
            if tr.current_iteration in (3, 4):
                keep = {g["name"]: g["params"][0].grad for g in tr.model.optimizer.param_groups}
                if os.environ.get("DIAG_CLEAR_GRADS"):       # (autograd accumulates IN PLACE into a gradient that is already there)
                    for g in tr.model.optimizer.param_groups:
                        g["params"][0].grad = None
                tr.join_side()
                with torch.enable_grad():
                    pkg = render(cams[2], tr.model, tr.cfg, tr.background, dirs=tr.dirs, raster_options=RasterOptions("rgb"))
                    (7.0 * pkg["render"]).sum().backward()
                for g in tr.model.optimizer.param_groups:
                    g["params"][0].grad = keep[g["name"]]
            return real_exchange(ov, surgery, rec)

        tr._exchange_grads = exchange_after_an_eval_render
    for it in range(steps):
        if it == 6:
            tr.model.active_sh_degree = 3
        if it == 11:
            tr.overlap_min_gaussians = 10 ** 9
        tr.train_step()
    tr.join_side()
    torch.cuda.synchronize()
    return {k: getattr(tr.model, k).detach().clone() for k in KEYS}


def frac(a, b):
    out = {}
    for k in KEYS:
        d = (a[k] - b[k]).abs()
        tol = 5e-3 * max(1.0, float(a[k].abs().max()))
        out[k] = round(float((d > tol).double().mean()) * 100, 3)
    return out


for rep in range(2):
    s0, s1 = run(False, False), run(False, False)
    t0, t1 = run(True, False), run(True, True)
    print("rep", rep)
    print("  serial vs serial (noise)      ", frac(s0, s1))
    print("  serial vs two-stream, kernel  ", frac(s0, t0))
    print("  serial vs two-stream, raster  ", frac(s0, t1))
    print("  two-stream kernel vs raster   ", frac(t0, t1))
    h0, h1, hs = run(True, False, hook=True), run(True, True, hook=True), run(False, False, hook=True)
    print("  with the test's evaluation render between backward and exchange at iterations 3, 4:")
    print("  kernel: hook vs none          ", frac(h0, t0))
    print("  raster: hook vs none          ", frac(h1, t1))
    print("  serial: hook vs none          ", frac(hs, s0))
    for steps in (3, 4, 5):
        a, b = run(True, True, steps=steps, hook=True), run(True, True, steps=steps)
        print("  raster, %d steps: hook vs none " % steps, frac(a, b))
