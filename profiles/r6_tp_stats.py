"""Reads the counters of a -DVCR_TPSTATS build of the two-phase compositing forward (csrc/composite.hip): where a wave's
shader-clock cycles go and how much work each part did.
    bash profiles/r6_build_variant.sh tpstats composite.hip "-DVCR_TPSTATS"
    VCR_LIB=$PWD/vcr_gaus_amd/libvcr_raster_tpstats.so python profiles/r6_tp_stats.py [workload ...]"""
import ctypes
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

NAMES = ["waves", "cycles", "cull+stage", "phase1", "phase2", "flushes", "p2_iters", "survivors", "chunks", "candidates", "longest_wave", "hits", "cull", "stage"]
dev = torch.device("cuda:0")
lib = _lib.load()
buf = (ctypes.c_uint32 * 130)()
for wl in sys.argv[1:] or ["metric_1m_1080p"]:
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
    bg = torch.zeros(3, device=dev)
    for ql in (False, True):
        for ci in (0, 1):
            with torch.no_grad():
                render(cams[ci], m, cfg, bg, dirs=dirs, geometry=False, raster_options=RasterOptions(quad_lists=ql))
            torch.cuda.synchronize()
            lib.vcr_debug_hit_histogram(buf, 1)
            _lib.profile_enable(True, stages=["composite_fwd"])
            _lib.profile_read()
            with torch.no_grad():
                render(cams[ci], m, cfg, bg, dirs=dirs, geometry=False, raster_options=RasterOptions(quad_lists=ql))
            torch.cuda.synchronize()
            kern_ms = _lib.profile_read()["composite_fwd"][0]
            _lib.profile_enable(False)
            lib.vcr_debug_hit_histogram(buf, 1)
            v = {k: buf[2 * i] | (buf[2 * i + 1] << 32) for i, k in enumerate(NAMES)}
            c = max(v["cycles"], 1)
            print(f"{wl} ql={int(ql)} cam{ci}: waves {v['waves']} chunks {v['chunks']} survivors {v['survivors']} flushes {v['flushes']} "
                  f"(group {v['survivors'] / max(v['flushes'], 1):.1f}) candidates {v['candidates']} ({v['candidates'] / max(v['survivors'], 1):.1f}/surv) "
                  f"hits {v['hits']} ({v['hits'] / max(v['survivors'], 1):.1f}/surv) phase-2 iterations {v['p2_iters']} ({v['p2_iters'] / max(v['survivors'], 1):.3f}/surv, "
                  f"lane use {v['hits'] / max(64 * v['p2_iters'], 1):.2f})")
            print(f"    wave cycles: total {c:.3e}; cull+stage {v['cull+stage'] / c:.1%} ({v['cull+stage'] / max(v['chunks'], 1):.0f}/chunk), "
                  f"phase 1 {v['phase1'] / c:.1%} ({v['phase1'] / max(v['flushes'], 1):.0f}/flush, {v['phase1'] / max(v['survivors'], 1):.0f}/surv), "
                  f"phase 2 {v['phase2'] / c:.1%} ({v['phase2'] / max(v['p2_iters'], 1):.0f}/iteration, {v['phase2'] / max(v['survivors'], 1):.0f}/surv); "
                  f"of cull+stage: chunk top -> survivor ballot {v['cull'] / c:.1%} ({v['cull'] / max(v['chunks'], 1):.0f}/chunk), staging {v['stage'] / c:.1%} "
                  f"({v['stage'] / max(v['chunks'], 1):.0f}/chunk); longest wave {v['longest_wave']} cycles, kernel {1e3 * kern_ms:.1f} us "
                  f"=> counter runs at {v['longest_wave'] / max(1e3 * kern_ms, 1e-9):.0f} ticks/us or faster", flush=True)
    del m
    torch.cuda.empty_cache()
