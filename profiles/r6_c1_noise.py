"""Round 6: run-to-run spread of the c1 gradient figures.  The compositing backward adds its per-pair values with fp32 atomics in
whatever order the waves arrive; the test figure (max-norm relative error on the non-fragile rows) is ONE row of a tensor.  Same
forward, REPS backward passes per camera: min / median / max of every tensor's figure, and the figure of the element-wise median of
three consecutive passes.    python profiles/r6_c1_noise.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402
import tests.test_raster_parity_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device("cuda:0")
n, W, H, f, sm, sem = T.CASES[2]
keys = ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d"]
for view in range(4):
    cam, inp, dirs = util.make_case(n, W, H, f, seed=0, scale_mult=sm, view=view, n_views=4)
    bg = torch.tensor([0.1, 0.3, 0.7])
    (ref, rradii, st), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True)
    wgt = torch.randn(ref.shape, generator=torch.Generator().manual_seed(40 + view), dtype=torch.float64)
    (ref * wgt).sum().backward()
    (out, radii), hl = util.hip_forward(cam, inp, dirs, bg, dev, requires_grad=True)
    clean, flipped = util.flip_clean_mask(cam, inp, out, ref, bg)
    keep = clean & ~rl["fragile"]
    loss = (out * wgt.float().to(dev)).sum()
    grads = {k: [] for k in keys}
    for r in range(reps):
        for k in keys:
            hl[k].grad = None
        loss.backward(retain_graph=True)
        for k in keys:
            grads[k].append(hl[k].grad.detach().cpu().clone())
    for k in keys:
        refk = rl[k].grad[keep]
        fig = sorted(util.rel_err(g[keep], refk) for g in grads[k])
        med3 = [util.rel_err(torch.stack(grads[k][i:i + 3]).median(0).values[keep], refk) for i in range(0, reps - 2, 3)]
        print(f"view {view} {k:8s} rows {int(keep.sum())}: min {fig[0]:.2e} median {fig[len(fig) // 2]:.2e} max {fig[-1]:.2e} | "
              f"above 1e-4: {sum(x >= 1e-4 for x in fig)}/{reps} | median-of-3 passes: max {max(med3):.2e}", flush=True)
