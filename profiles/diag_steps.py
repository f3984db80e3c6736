"""Per-step wall time + caching-allocator traffic of a workload (host-side stall hunting).
Usage: python profiles/diag_steps.py [workload] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5_360_5m_1600x1200"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
cams = synthetic.make_cameras(8, W, H, focal, device=dev)
tr = BenchTrainer(raw, cams, dev, world=1, rank=0)
for i in range(steps):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st1 = torch.cuda.memory_stats()
    print(f"step {i:2d} {1e3 * dt:8.2f} ms  R={tr.last_R}  device_alloc +{st1['num_device_alloc'] - st0['num_device_alloc']}"
          f" device_free +{st1['num_device_free'] - st0['num_device_free']}  reserved {st1['reserved_bytes.all.current'] / 2**30:.2f} GiB"
          f"  retries {st1['num_alloc_retries']}", flush=True)

# pipelined (no per-step sync), as bench.py runs it
st0 = torch.cuda.memory_stats()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for i in range(steps):
    h0 = time.perf_counter()
    tr.step(steps + i)
    host.append(1e3 * (time.perf_counter() - h0))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st1 = torch.cuda.memory_stats()
print(f"pipelined: {1e3 * dt / steps:.2f} ms/step; host ms per step: " + " ".join(f"{h:.1f}" for h in host))
print(f"  device_alloc +{st1['num_device_alloc'] - st0['num_device_alloc']} device_free +{st1['num_device_free'] - st0['num_device_free']}"
      f" reserved {st1['reserved_bytes.all.current'] / 2**30:.2f} GiB retries {st1['num_alloc_retries']}")
