"""A/B of whole-step variants on the metric workload: ms/step (wall, pipelined) + stage times.
Usage: [ENV=...] python profiles/ab_step.py [workload] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "metric_1m_1080p"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
cams = synthetic.make_cameras(8, W, H, focal, device=dev)
tr = BenchTrainer(raw, cams, dev)
import contextlib
hp = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get("AB_HIGHPRIO") else None
ctx = (lambda: torch.cuda.stream(hp)) if hp is not None else contextlib.nullcontext
if hp is not None:
    hp.wait_stream(torch.cuda.current_stream(dev))
_step = tr.step
def step(i):
    with ctx():
        _step(i)
tr.step = step
tr.prime()
for i in range(10):
    tr.step(i)
torch.cuda.synchronize()
st = os.environ.get("AB_STAGES")
_lib.profile_enable(st != "none", None if not st or st == "none" else st.split(","))
_lib.profile_read()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(10 + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr = _lib.profile_read()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCR_"))
print(f"[{tag}] {wl}: {1e3 * dt / steps:.3f} ms/step  " + " ".join(f"{k}={1e3 * v[0] / max(v[1], 1):.0f}us" for k, v in pr.items()), flush=True)
