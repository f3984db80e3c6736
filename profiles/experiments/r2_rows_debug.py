"""Counters of the row-stream forward (-DVCR_ROWS_DEBUG build): drain iterations vs compacted survivors per frame."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VCR_ROWS_FWD"] = "1"
from vcr_gaus_amd import _lib, synthetic
from vcr_gaus_amd.config import make_config
from vcr_gaus_amd.gaussian_model import GaussianModel
from vcr_gaus_amd.gaussian_renderer import render
from vcr_gaus_amd.graphics_utils import get_all_px_dir
dev = torch.device("cuda:0")
n, views, W, H, focal, sem = synthetic.WORKLOADS["metric_1m_1080p"]
raw = synthetic.make_gaussians(n, seed=0)
cams = synthetic.make_cameras(8, W, H, focal, device=dev)
cfg = make_config("tnt")
m = GaussianModel(cfg.model); m.create_from_params(raw, 1.0, device=dev); m.active_sh_degree = 3; m.extent = 3.3
dirs = get_all_px_dir(cams[0].intr, H, W)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
for ci in (2, 0):
    with torch.no_grad():
        render(cams[ci], m, cfg, torch.zeros(3, device=dev), dirs=dirs)
    torch.cuda.synchronize()
    lib.vcr_rows_debug_read(buf, 1)
    with torch.no_grad():
        render(cams[ci], m, cfg, torch.zeros(3, device=dev), dirs=dirs)
    torch.cuda.synchronize()
    lib.vcr_rows_debug_read(buf, 1)
    it, surv, drains, blk, chunks = buf[0], buf[1], buf[2], buf[3], buf[4]
    print(f"cam{ci}: chunks={chunks} survivors={surv} drains={drains} block_entries={blk} iterations={it} "
          f"iter/surv={it/max(surv,1):.3f} blocks/surv={blk/max(surv,1):.2f} surv/drain={surv/max(drains,1):.1f}")
