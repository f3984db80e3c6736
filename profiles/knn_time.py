"""Time vcr_knn3_mean_dist2 (grid search) on synthetic clouds; `python profiles/knn_time.py` on the GPU box."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic

lib = _lib.load()
for n in (100_000, 1_000_000, 5_000_000):
    pts = synthetic.make_gaussians(n, seed=0)["xyz"].cuda().contiguous()
    out = torch.empty(n, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(lib.vcr_knn3_mean_dist2(n, pts.data_ptr(), out.data_ptr(), _lib.stream_of(pts)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"knn3 n={n}: {dt * 1e3:.2f} ms  mean dist2 {float(out.mean()):.3e}", flush=True)
