"""Host-side (Python) cost of a training step: cProfile over 200 steps of the metric workload, top functions by cumulative
time.  The step is GPU-bound on a fast host; on a slow host the window between the two compositing kernels (loss forward,
autograd backward set-up) becomes host-bound, which shows up as box-to-box variance of the benchmark.
    python profiles/host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

dev = torch.device("cuda", 0)
n, views, W, H, focal, sem = synthetic.WORKLOADS["metric_1m_1080p"]
raw = synthetic.make_gaussians(n, seed=0)
cams = synthetic.make_cameras(8, W, H, focal, device=dev)
bt = BenchTrainer(raw, cams, dev)
bt.prime()
for i in range(10):
    bt.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    bt.step(10 + i)
torch.cuda.synchronize()
pr.disable()
print(f"{1e3 * (time.perf_counter() - t0) / 200:.3f} ms/step under cProfile")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
