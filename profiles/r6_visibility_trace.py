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
    print(f"visibility batch {r}: {len(vcams)} cameras, {dt:.1f} ms, {dt / len(vcams) * 1e3:.0f} us per camera, visible {int(visi.sum())} of {visi.numel()}", flush=True)
