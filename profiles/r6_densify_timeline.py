"""Round 6: timeline of ONE densification event of the schedule bench.py measures (tnt preset, 1 M Gaussians): wall clock of its
phases, the device drained between them (so the sum is larger than the undisturbed event bench.py reports).
    python profiles/r6_densify_timeline.py > profiles/r6_densify_timeline.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

dev = torch.device("cuda:0")
n, views, W, H, focal, sem, smult = synthetic.workload("metric_1m_1080p")
raw = synthetic.make_gaussians(n, seed=0)
cams = synthetic.make_cameras(8, W, H, focal, radius=synthetic.camera_radius("metric_1m_1080p"), device=dev)
bt = BenchTrainer(raw, cams, dev)
bt.prime()
tr = bt.tr
o = tr.cfg.optim
o.densify_from_iter, o.densification_interval, o.densify_until_iter = tr.current_iteration, 100, 10 ** 9
marks = []


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        marks.append((label, 1e3 * (time.perf_counter() - t0)))
        return r
    setattr(obj, name, wrap)


timed(tr, "_visibility_cameras", "virtual cameras (host half formed ahead: copies + camera objects)")
timed(tr, "visibility_mask", "visibility_mask: 200 renders (flags) + inside-box test")
timed(tr, "sync_densify_stats", "sync_densify_stats")
timed(tr.model, "densify_and_prune", "densify_and_prune (selection + row surgery)")
for name in ("densify_and_clone", "densify_and_split_along_maxscaling", "prune_points"):
    if hasattr(tr.model, name):
        timed(tr.model, name, "  of which " + name)
step = 0
for ev in range(3):
    while True:
        before = tr.model._xyz.shape[0]
        marks.clear()
        tr.join_side(); torch.cuda.synchronize(); t0 = time.perf_counter()
        bt.step(step); step += 1
        tr.join_side(); torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        if tr.model._xyz.shape[0] != before:
            break
    print(f"event {ev}: iteration {tr.current_iteration}, N {before} -> {tr.model._xyz.shape[0]}, whole iteration {dt:.1f} ms (drained between phases)")
    acc = 0.0
    for label, ms in marks:
        print(f"    {ms:8.2f} ms  {label}")
        if not label.startswith("  of which"):
            acc += ms
    print(f"    {dt - acc:8.2f} ms  rest of the iteration (render, losses, backward, optimizer at the old and new sizes, allocator)")
    t0 = time.perf_counter()
    for k in range(3):
        bt.step(step); step += 1
    tr.join_side(); torch.cuda.synchronize()
    print(f"    next three iterations: {1e3 * (time.perf_counter() - t0) / 3:.2f} ms each")
