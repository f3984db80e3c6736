"""Full-size sanity (not a benchmark): a few training steps of the large workloads, asserting finite parameters / losses
(the loss level depends on the camera of the step, so it is only printed).  python profiles/sanity_fullsize.py [workload ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import make_synthetic_trainer  # noqa: E402

dev = torch.device("cuda", 0)
for wl in (sys.argv[1:] or ["c4_tnt_2m_1080p", "c5_360_5m_1600x1200", "metric_1m_1080p"]):
    n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
    raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
    cams = synthetic.make_cameras(4, W, H, focal, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, preset="tnt", gt_jitter=0.3,
                                optim={"densify_from_iter": 10 ** 9, "prune": {"iterations": []}})
    first = last = None
    for it in range(24):
        tr.train_step()
        if it in (3, 23):
            tr.join_side()
            v = float(tr.losses["total"])
            first, last = (v, last) if it == 3 else (first, v)
    tr.join_side()
    torch.cuda.synchronize()
    m = tr.model
    ok = all(bool(torch.isfinite(getattr(m, a)).all()) for a in ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"])
    print(f"{wl}: N={n} sem={sem} two-stream={tr.overlap_sh and n >= tr.overlap_min_gaussians} finite={ok} "
          f"loss {first:.5f} -> {last:.5f}", flush=True)
    assert ok and first == first and last == last
    del tr, raw, cams
    torch.cuda.empty_cache()
