"""Per-stage HIP-event times of the rasterizer (forward + backward) on a synthetic workload, without the trainer.
Usage: [VCR_LIB=build/libX.so] python profiles/stage_times.py [workload] [reps] [scale_mult]
Prints one line per camera and the mean over cameras.  scale_mult > 1 inflates the Gaussians (denser tile lists)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic, rasterizer  # noqa: E402
from vcr_gaus_amd.config import make_config  # noqa: E402
from vcr_gaus_amd.gaussian_model import GaussianModel  # noqa: E402
from vcr_gaus_amd.gaussian_renderer import render  # noqa: E402
from vcr_gaus_amd.graphics_utils import get_all_px_dir  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "metric_1m_1080p"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mult = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device("cuda:0")
n, views, W, H, focal, sem = synthetic.WORKLOADS[wl]
raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
raw["scaling"] = raw["scaling"] + math.log(mult)
cams = synthetic.make_cameras(8, W, H, focal, device=dev)
cfg = make_config("tnt")
m = GaussianModel(cfg.model)
m.create_from_params(raw, 1.0, device=dev)
m.active_sh_degree = 3
m.extent = 3.3
dirs = get_all_px_dir(cams[0].intr, H, W)
bg = torch.zeros(3, device=dev)
tot = {}
for ci, c in enumerate(cams[:4]):
    for rep in range(reps + 1):
        if rep == 1:
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            _lib.profile_read()
        pkg = render(c, m, cfg, bg, dirs=dirs, geometry=False)
        out = pkg["render_out"]
        out.square().sum().backward()
    torch.cuda.synchronize()
    pr = _lib.profile_read()
    _lib.profile_enable(False)
    line = {k: v[0] / max(v[1], 1) for k, v in pr.items()}
    for k, v in line.items():
        tot[k] = tot.get(k, 0.0) + v / 4
    print(f"cam{ci} R={pkg['raster'].R} " + " ".join(f"{k}={1e3 * v:.1f}us" for k, v in line.items()), flush=True)
print(f"MEAN {wl} x{mult}: " + " ".join(f"{k}={1e3 * v:.1f}us" for k, v in tot.items()), flush=True)
