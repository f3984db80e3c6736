"""Round 6: what the 200 visibility renders of one densification event are made of.  Sets the metric trainer up, runs the visibility
batch of a densification (tnt preset: 200 virtual cameras at 1500 x 1500, flags only) a few times and prints its wall clock; run
under `rocprofv3 --kernel-trace --stats` the kernel summary is dominated by the 200 x reps renders.
    python profiles/r6_visibility_trace.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
n, views, W, H, focal, sem, smult = synthetic.workload("metric_1m_1080p")
raw = synthetic.make_gaussians(n, seed=0)
cams = synthetic.make_cameras(8, W, H, focal, radius=synthetic.camera_radius("metric_1m_1080p"), device=dev)
bt = BenchTrainer(raw, cams, dev)
bt.prime()
tr = bt.tr
dl = tr.cfg.optim.densify_large
for r in range(reps + 1):
    vcams = tr._visibility_cameras(dl.sample_cams)
    tr.join_side(); torch.cuda.synchronize(); t0 = time.perf_counter()
    visi = tr.visibility_mask(vcams)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    if r == reps:                      # what the cameras look like to the rasterizer: instances per camera, visible Gaussians
        from vcr_gaus_amd.gaussian_renderer import visibility_counts  # noqa: F401
        from vcr_gaus_amd import rasterizer as RZ
        st = vcams[0]._stack
        import math
        from vcr_gaus_amd.gaussian_renderer import fused_activate, _cam_rotation
        sc, ro, op = fused_activate(tr.model, vcams[0].camera_center, _cam_rotation(vcams[0], dev), False)
        cnt, nr, nv = RZ.visibility_batch(st[0][:16], st[1][:16], st[2][:16], [math.tan(c.FoVx * 0.5) for c in vcams[:16]],
                                          [math.tan(c.FoVy * 0.5) for c in vcams[:16]], int(vcams[0].image_height), int(vcams[0].image_width),
                                          tr.model.get_xyz, op, sc, ro, None, 1.0, True, None, 0)
        print("first 16 cameras: 3-sigma tile instances", nr, "visible", nv, flush=True)
    print(f"visibility batch {r}: {len(vcams)} cameras, {dt:.1f} ms, {dt / len(vcams) * 1e3:.0f} us per camera, visible {int(visi.sum())} of {visi.numel()}", flush=True)
