"""Round 6, design measurement (CPU, no GPU): what a TWO-PHASE compositing forward would execute on the metric scene.

Today one wave owns an 8x8 quad and shades ONE surviving Gaussian per iteration on all 64 lanes (~10 of them hit).
Two-phase form: phase 1 finds, per group of G survivors, which pixels each survivor can reach (row spans, 8 survivors x 8 rows
per 64-lane step) and hands every pixel lane the bit list of ITS candidates; phase 2 lets every lane walk its OWN list, so an
iteration shades 64 different (pixel, Gaussian) pairs.  Phase 2 runs max_pixel(#own hits in the group) iterations: this script
measures that maximum against the survivor count on sampled tiles of the metric scene (oracle projection, exact per-pixel
hit test, T < 1e-4 stop), for several group sizes.

    python profiles/r6_twophase_sim.py [tile_stride] > profiles/r6_twophase_sim.txt
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_torch as OM  # noqa: E402
from oracle import raster_torch as OR  # noqa: E402
from vcr_gaus_amd import synthetic  # noqa: E402

torch.set_grad_enabled(False)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 11
name = sys.argv[2] if len(sys.argv) > 2 else "metric_1m_1080p"
n, views, W, H, focal, sem, smult = synthetic.workload(name)
raw = synthetic.make_gaussians(n, seed=0)
if smult != 1.0:
    raw["scaling"] = raw["scaling"] + math.log(smult)
for ci in (0, 1):
    cam = synthetic.make_cameras(8, W, H, focal, radius=synthetic.camera_radius(name))[ci]
    s = OR.Settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0, cam.world_view_transform,
                    cam.full_proj_transform, 3, cam.camera_center)
    act = OM.activations(raw)
    pre = OR.preprocess(s, act["xyz"], torch.zeros(n, 3), act["shs"], None, None, None, act["opacity"], act["scaling"],
                        act["rotation"], None)
    owner, beg, end, R = OR.bin_and_sort(pre)
    gx, gy = pre["grid"]
    lens = (end - beg)
    nonempty = torch.nonzero(lens > 0).squeeze(1)
    sample = nonempty[::stride]
    GROUPS = (16, 32, 64, 128, 256)
    tot = dict(surv=0, hits=0, entries=0, quads=0, pix_hits_max=0)
    iters = {G: 0 for G in GROUPS}
    iters_row = {G: 0 for G in GROUPS}      # same, but 16-lane rows progress independently (max over the 16 pixels of a 4x4 block, summed / 4)
    span_cand = 0                           # candidates of the row-span test (bounding interval per row) vs exact hits
    for t in sample.tolist():
        idx = owner[beg[t]:end[t]]
        tx, ty = t % gx, t // gx
        ys, xs = torch.meshgrid(torch.arange(ty * 16, ty * 16 + 16), torch.arange(tx * 16, tx * 16 + 16), indexing="ij")
        xs, ys = xs.reshape(-1).float(), ys.reshape(-1).float()
        inside = (xs < W) & (ys < H)
        dx = pre["px"][idx][None] - xs[:, None]
        dy = pre["py"][idx][None] - ys[:, None]
        con = pre["conic"][idx]
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        alpha = torch.clamp(pre["opacity"][idx][None] * torch.exp(power), max=0.99)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0) & inside[:, None]
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1.0 - a_eff, dim=1)
        stopped = torch.cumsum((valid & (T_incl < 1e-4)).int(), 1) > 0          # this entry or an earlier one stopped the pixel
        hit = valid & ~stopped                                                   # [256, L]
        # row-span candidates: per (Gaussian, pixel row) the interval between the first and last hit of `valid` in that row
        for q in range(4):
            qx, qy = q & 1, q >> 1
            pm = ((xs.long() // 8) % 2 == qx) & ((ys.long() // 8) % 2 == qy)
            hq = hit[pm]                                                         # [64, L], row-major 8x8
            keep = hq.any(0)
            hq = hq[:, keep]
            S = hq.shape[1]
            if S == 0:
                continue
            tot["quads"] += 1
            tot["surv"] += S
            tot["hits"] += int(hq.sum())
            tot["entries"] += int(idx.numel())
            v8 = valid[pm][:, keep].view(8, 8, S)                                # [row, col, S]
            first = torch.where(v8.any(1), v8.int().argmax(1), torch.full((8, S), 9))
            last = torch.where(v8.any(1), 7 - v8.flip(1).int().argmax(1), torch.full((8, S), -1))
            span_cand += int((last - first + 1).clamp_min(0).sum())
            for G in GROUPS:
                for g0 in range(0, S, G):
                    per_pix = hq[:, g0:g0 + G].sum(1)                            # [64]
                    iters[G] += int(per_pix.max())
                    blk = per_pix.view(8, 8).view(2, 4, 2, 4).permute(0, 2, 1, 3).reshape(4, 16).max(1).values
                    iters_row[G] += float(blk.float().mean())
    S = tot["surv"]
    print(f"{name} cam {ci}: tiles sampled {len(sample)} of {len(nonempty)} non-empty; quads {tot['quads']}; survivors (>= 1 hit pixel) {S}; "
          f"hit pairs {tot['hits']} = {tot['hits'] / S:.2f} per survivor; row-span candidates {span_cand / S:.2f} per survivor")
    for G in GROUPS:
        print(f"   group {G:4d}: phase-2 iterations = sum over groups of max_pixel(own hits) = {iters[G]} = {iters[G] / S:.3f} per survivor "
              f"(mean own hits per pixel per group {tot['hits'] / 64 / max(1, sum(1 for _ in range(1))) / 1:.0f} total/64); "
              f"if the four 16-lane rows advanced independently: {iters_row[G] / S:.3f}")
