"""Where does the HOST spend a training step?  (i) a scene so small that the device has nothing to do (the step time is the host's
enqueue time + the one device round trip of the instance-count hand-over); (ii) the metric scene with the host's enqueue time of each
step taken with perf_counter and NO synchronisation (how far ahead of the device the host could run); (iii) cProfile of (i)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

dev = torch.device("cuda:0")


def make(n, W, H, focal):
    return BenchTrainer(synthetic.make_gaussians(n, seed=0), synthetic.make_cameras(8, W, H, focal, device=dev), dev)


def timed(bt, k, base):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for i in range(k):
        h0 = time.perf_counter()
        bt.step(base + i)
        host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    host.sort()
    return dict(ms_per_step=1e3 * t_all / k, host_enqueue_ms_per_step=1e3 * t_enq / k, host_step_median_ms=1e3 * host[k // 2],
                host_step_p10_ms=1e3 * host[k // 10])


tiny = make(20000, 160, 120, 150.0)
for i in range(30):
    tiny.step(i)
print("tiny (20 k Gaussians, 160x120):", timed(tiny, 200, 30))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(200):
    tiny.step(300 + i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:9000])
del tiny
n, views, W, H, focal, sem, smult = synthetic.workload("metric_1m_1080p")
bt = make(n, W, H, focal)
for i in range(20):
    bt.step(i)
print("metric:", timed(bt, 100, 20))
