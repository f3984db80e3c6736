"""Host time of a training step, by C-ABI call (wall-clock inside each entry point, no synchronisation) and the Python remainder.
Run with VCR_HOST_TRACE=1 to get the split of vcr_rasterize_forward itself on stderr."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
acc = defaultdict(float)
cnt = defaultdict(int)


def wrap(name, f):
    def g(*a):
        t = time.perf_counter()
        r = f(*a)
        acc[name] += time.perf_counter() - t
        cnt[name] += 1
        return r
    return g


for name in _lib.SYMBOLS:
    if name.startswith("vcr_") and name not in ("vcr_last_error", "vcr_abi_version", "vcr_sums_elems"):
        try:
            setattr(lib, name, wrap(name, getattr(lib, name)))
        except AttributeError:
            pass

wl = sys.argv[1] if len(sys.argv) > 1 else "metric_1m_1080p"
n, views, W, H, focal, sem, smult = synthetic.workload(wl)
bt = BenchTrainer(synthetic.make_gaussians(n, seed=0), synthetic.make_cameras(8, W, H, focal, device=dev), dev)
for i in range(30):
    bt.step(i)
torch.cuda.synchronize()
acc.clear()
cnt.clear()
K = 400
t0 = time.perf_counter()
for i in range(K):
    bt.step(30 + i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{wl}: {1e3 * t_all / K:.4f} ms/step, host enqueue {1e3 * t_enq / K:.4f} ms/step")
tot = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {1e6 * v / K:8.1f} us/step  ({cnt[k] / K:.1f} calls)")
    tot += v
print(f"  inside the C ABI {1e6 * tot / K:.1f} us/step, Python + torch remainder {1e6 * (t_enq - tot) / K:.1f} us/step")
