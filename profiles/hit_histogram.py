"""How many of the 64 pixels of an 8x8 quad does a surviving (quad, Gaussian) pair really hit?  
Instrumented build of the library (-DVCR_HITHIST), one forward + backward per camera on a workload:

    make -C vcr_gaus_amd/csrc EXTRA=-DVCR_HITHIST LIB=../libvcr_hithist.so BUILD=../../build/hithist -j     # (build container)
    VCR_LIB=$PWD/vcr_gaus_amd/libvcr_hithist.so python profiles/hit_histogram.py [workload] > profiles/r3_hit_histogram.txt
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic  # noqa: E402
from vcr_gaus_amd.config import make_config  # noqa: E402
from vcr_gaus_amd.gaussian_model import GaussianModel  # noqa: E402
from vcr_gaus_amd.gaussian_renderer import render  # noqa: E402
from vcr_gaus_amd.graphics_utils import get_all_px_dir  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "metric_1m_1080p"
dev = torch.device("cuda:0")
n, views, W, H, focal, sem, smult = synthetic.workload(wl)
raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
if smult != 1.0:
    import math
    raw["scaling"] = raw["scaling"] + math.log(smult)
cams = synthetic.make_cameras(4, W, H, focal, device=dev)
cfg = make_config("tnt")
cfg.optim.loss_weight.semantic = 0.0
m = GaussianModel(cfg.model)
m.create_from_params(raw, spatial_lr_scale=1.0, device=dev)
m.active_sh_degree = 3
dirs = get_all_px_dir(cams[0].intr, H, W)
lib = _lib.load()
buf = (C.c_uint32 * 130)()
_lib.check(lib.vcr_debug_hit_histogram(buf, 1))
R = E = 0
for cam in cams:
    pkg = render(cam, m, cfg, torch.zeros(3, device=dev), dirs=dirs)
    (pkg["render"].sum() + pkg["depth"].sum() + pkg["normal"].sum()).backward()
    R += pkg["raster"].R
    E += pkg["raster"].emitted
_lib.check(lib.vcr_debug_hit_histogram(buf, 0))
h = list(buf)
print(f"# {wl}: {len(cams)} cameras, tile instances R = {R} (3-sigma), emitted R' = {E} ({E / max(R, 1):.3f} R)")
for name, off in (("forward", 0), ("backward", 65)):
    hh = h[off:off + 65]
    tot = sum(hh)
    mean = sum(k * c for k, c in enumerate(hh)) / max(tot, 1)
    print(f"# {name}: {tot} surviving (quad, Gaussian) pairs, mean hit pixels {mean:.2f} of 64")
    print("# hits  pairs      share   cumulative")
    cum = 0
    for k, c in enumerate(hh):
        cum += c
        if c:
            print(f"{k:5d} {c:10d} {c / tot:8.4f} {cum / tot:8.4f}")
