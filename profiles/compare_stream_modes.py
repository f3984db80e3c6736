"""Loss sequence of the two-stream SH path vs the serial loop on a full-size workload (they must agree):
    python profiles/compare_stream_modes.py c4_tnt_2m_1080p"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic
from vcr_gaus_amd.trainer import make_synthetic_trainer
dev = torch.device("cuda", 0)
wl = sys.argv[1]
n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
res = {}
for overlap in (True, False):
    cams = synthetic.make_cameras(4, W, H, focal, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, preset="tnt", gt_jitter=0.3, overlap_sh=overlap,
                                optim={"densify_from_iter": 10 ** 9, "prune": {"iterations": []}})
    ls = []
    for it in range(16):
        tr.train_step(); tr.join_side()
        ls.append(round(float(tr.losses["total"]), 5))
    res[overlap] = ls
    print("two-stream" if overlap else "serial   ", ls, flush=True)
    del tr; torch.cuda.empty_cache()
print("max abs diff", max(abs(a - b) for a, b in zip(res[True], res[False])))
