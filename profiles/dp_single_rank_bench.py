"""Step time of the DATA-PARALLEL code path on one GPU: a one-rank RCCL group with the collectives forced on
(all_gather_into_tensor of dL/drgb + bucketed all_reduce, both trivial at one rank), two-stream form on / off.
Shows what the DP path costs besides communication.  python profiles/dp_single_rank_bench.py"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import make_synthetic_trainer  # noqa: E402

dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, views, W, H, focal, sem = synthetic.WORKLOADS["metric_1m_1080p"]
raw = synthetic.make_gaussians(n, seed=0)
for overlap in (True, False):
    cams = synthetic.make_cameras(8, W, H, focal, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, preset="tnt", overlap_sh=overlap, force_factorised=not overlap,
                                optim={"densify_from_iter": 10 ** 9, "prune": {"iterations": []}})
    tr.force_collectives = True
    for _ in range(12):
        tr.train_step()
    tr.join_side(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.train_step()
    tr.join_side(); torch.cuda.synchronize()
    print(f"DP code path, one rank, two-stream={overlap}: {1e3 * (time.perf_counter() - t0) / 30:.3f} ms/step", flush=True)
    del tr
dist.destroy_process_group()
