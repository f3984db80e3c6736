"""A/B of compositing-forward builds: times the stage with HIP events and dumps what the kernel wrote (image, final_T,
n_contrib) so that two builds can be compared BIT FOR BIT (profiles/r6_fwd_cmp.py).
    VCR_LIB=$PWD/vcr_gaus_amd/libX.so python profiles/r6_fwd_ab.py <tag> [workload ...]
Dumps go to /tmp/r6ab/<tag>/ (scratch on the GPU box), one summary line per (workload, camera, list mode) to stdout."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib, synthetic  # noqa: E402
from vcr_gaus_amd.config import make_config  # noqa: E402
from vcr_gaus_amd.gaussian_model import GaussianModel  # noqa: E402
from vcr_gaus_amd.gaussian_renderer import render  # noqa: E402
from vcr_gaus_amd.graphics_utils import get_all_px_dir  # noqa: E402
from vcr_gaus_amd.rasterizer import RasterOptions  # noqa: E402

tag = sys.argv[1]
loads = sys.argv[2:] or ["metric_1m_1080p"]
reps = int(os.environ.get("AB_REPS", "10"))
ncam = int(os.environ.get("AB_CAMS", "4"))
dump = os.environ.get("AB_DUMP", "1") == "1"
dev = torch.device("cuda:0")
outdir = f"/tmp/r6ab/{tag}"
os.makedirs(outdir, exist_ok=True)
for wl in loads:
    n, views, W, H, focal, sem, smult = synthetic.workload(wl)
    raw = synthetic.make_gaussians(n, seed=0, sem_channels=sem)
    if smult != 1.0:
        raw["scaling"] = raw["scaling"] + math.log(smult)
    cams = synthetic.make_cameras(8, W, H, focal, radius=synthetic.camera_radius(wl), device=dev)
    cfg = make_config("tnt")
    m = GaussianModel(cfg.model)
    m.create_from_params(raw, 1.0, device=dev)
    m.active_sh_degree = 3
    m.extent = 3.3
    dirs = get_all_px_dir(cams[0].intr, H, W)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    P = H * W
    mean = {}
    for ql in (False, True):
        tot_f = tot_b = 0.0
        for ci, c in enumerate(cams[:ncam]):
            for rep in range(reps + 1):
                if rep == 1:
                    torch.cuda.synchronize()
                    _lib.profile_enable(True, stages=["composite_fwd", "composite_bwd"])
                    _lib.profile_read()
                pkg = render(c, m, cfg, bg, dirs=dirs, geometry=False, raster_options=RasterOptions(quad_lists=ql))
                out = pkg["render_out"]
                if rep == 0 and dump:
                    st = out.grad_fn.state[_lib.BUF_IMAGE]
                    off = (4 * P + 255) // 256 * 256
                    torch.save({"out": out.detach().cpu(), "final_T": st[:4 * P].view(torch.float32).cpu(),
                                "n_contrib": st[off:off + 4 * P].view(torch.int32).cpu()}, f"{outdir}/{wl}_cam{ci}_ql{int(ql)}.pt")
                out.square().sum().backward()
            torch.cuda.synchronize()
            pr = _lib.profile_read()
            _lib.profile_enable(False)
            f = 1e3 * pr["composite_fwd"][0] / max(pr["composite_fwd"][1], 1)
            b = 1e3 * pr["composite_bwd"][0] / max(pr["composite_bwd"][1], 1)
            tot_f += f / ncam
            tot_b += b / ncam
            print(f"{tag} {wl} ql={int(ql)} cam{ci} R={pkg['raster'].R} E={pkg['raster'].emitted} fwd={f:.1f}us bwd={b:.1f}us", flush=True)
        mean[ql] = (tot_f, tot_b)
    print(f"MEAN {tag} {wl}: per-tile fwd={mean[False][0]:.1f}us bwd={mean[False][1]:.1f}us | per-quad fwd={mean[True][0]:.1f}us bwd={mean[True][1]:.1f}us",
          flush=True)
    del m, raw
    torch.cuda.empty_cache()
