"""How evenly does the compositing forward load the 1024 SIMDs?  Instrumented build (-DVCR_TIMING: per wave start / end clock,
chunks, survivors, and WHERE it ran: XCC / SE / SH / CU / SIMD from HW_REG_HW_ID + HW_REG_XCC_ID).
    make -C vcr_gaus_amd/csrc BUILD=../../build/csrc_timing LIB=../libvcr_raster_timing.so EXTRA=-DVCR_TIMING
    VCR_LIB=$PWD/vcr_gaus_amd/libvcr_raster_timing.so python profiles/simd_balance.py"""
import os
import sys

import torch

os.environ["VCR_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.config import make_config  # noqa: E402
from vcr_gaus_amd.gaussian_model import GaussianModel  # noqa: E402
from vcr_gaus_amd.gaussian_renderer import render  # noqa: E402
from vcr_gaus_amd.graphics_utils import get_all_px_dir  # noqa: E402
from vcr_gaus_amd.rasterizer import RasterOptions  # noqa: E402

dev = torch.device("cuda:0")
raw = synthetic.make_gaussians(1_000_000, seed=0)
cams = synthetic.make_cameras(8, 1920, 1080, 1165.0, device=dev)
cfg = make_config("tnt")
m = GaussianModel(cfg.model); m.create_from_params(raw, 1.0, device=dev); m.active_sh_degree = 3; m.extent = 3.3
dirs = get_all_px_dir(cams[0].intr, 1080, 1920)
for ql in (False, True):
    for c in cams[:2]:
        for rep in range(2):
            with torch.no_grad():
                pkg = render(c, m, cfg, torch.zeros(3, device=dev), dirs=dirs, raster_options=RasterOptions(quad_lists=ql))
        torch.cuda.synchronize()
        t4 = pkg["raster"].timing.view(torch.int64).view(-1, 4).cpu()
        t4 = t4[t4[:, 1] > 0]
        f2 = t4[:, 2]
        surv = (f2 & 0xFFFFFFFF).double(); chunks = ((f2 >> 32) & 0xFFFF).double(); place = (f2 >> 48) & 0xFFFF
        dur = (t4[:, 1] - t4[:, 0]).double() / 100.0
        span = float(t4[:, 1].max() - t4[:, 0].min()) / 100.0
        cost = surv + 2.2 * chunks                       # ~instruction-weighted work of a wave (a chunk's culling ~ 2 survivors)
        keys, inv = torch.unique(place, return_inverse=True)
        per = torch.zeros(len(keys), dtype=torch.float64).index_add_(0, inv, cost)
        persurv = torch.zeros(len(keys), dtype=torch.float64).index_add_(0, inv, surv)
        nw = torch.zeros(len(keys), dtype=torch.float64).index_add_(0, inv, torch.ones_like(cost))
        first = torch.full((len(keys),), 1e30, dtype=torch.float64).scatter_reduce(0, inv, t4[:, 0].double() / 100.0, "amin")
        last = torch.zeros(len(keys), dtype=torch.float64).scatter_reduce(0, inv, t4[:, 1].double() / 100.0, "amax")
        busy = last - first
        print(f"quad_lists={ql} cam={c.uid}: waves={len(t4)} busy(>=10 surv)={(surv >= 10).sum()} SIMDs seen={len(keys)} span={span:.1f}us "
              f"survivors={int(surv.sum())} chunks={int(chunks.sum())}")
        print(f"   per-SIMD work (surv + 2.2 chunks): mean={per.mean():.0f} max={per.max():.0f} p90={per.quantile(0.9):.0f} p10={per.quantile(0.1):.0f} "
              f"max/mean={per.max() / per.mean():.2f}; waves per SIMD mean={nw.mean():.1f} max={nw.max():.0f}")
        print(f"   per-SIMD busy time: mean={busy.mean():.1f}us max={busy.max():.1f} p90={busy.quantile(0.9):.1f}; corr(work, busy)={torch.corrcoef(torch.stack([per, busy]))[0, 1]:.2f}; "
              f"time per unit work on the busiest-decile SIMDs={(busy[per >= per.quantile(0.9)].sum() / per[per >= per.quantile(0.9)].sum()):.4f}us, all={(busy.sum() / per.sum()):.4f}us")
        print(f"   balanced bound: span x mean/max = {span * per.mean() / per.max():.1f}us; longest single wave={dur.max():.1f}us (surv {int(surv[dur.argmax()])}, chunks {int(chunks[dur.argmax()])})")
