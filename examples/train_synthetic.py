#!/usr/bin/env python
"""End-to-end training on a synthetic scene with the reference's schedule compressed in time
(densify / prune / opacity reset / SH-degree ramp), reporting PSNR against the held ground truth.
    python examples/train_synthetic.py --gaussians 100000 --width 400 --height 300 --iters 1500
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=400)
    ap.add_argument("--height", type=int, default=300)
    ap.add_argument("--focal", type=float, default=300.0)
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--preset", default="dtu_c3")
    args = ap.parse_args()
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.gaussian_renderer import render
    from vcr_gaus_amd.loss_utils import psnr
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    dev = torch.device("cuda:0")
    raw = synthetic.make_gaussians(args.gaussians, seed=0)
    cams = synthetic.make_cameras(args.views, args.width, args.height, args.focal, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, preset=args.preset, gt_jitter=0.3,
                                optim={"densify_from_iter": 100, "densification_interval": 100, "densify_until_iter": 1000,
                                       "opacity_reset_interval": 600, "prune": {"iterations": [1200]},
                                       "consistent_normal_from_iter": 500})

    def eval_psnr():
        tr.join_side()          # the last SH update may still be pending on / running on the trainer's second stream
        with torch.no_grad():
            vals = [float(psnr(render(c, tr.model, tr.cfg, tr.background, dirs=tr.dirs)["render"].clamp(0, 1),
                               c.original_image).mean()) for c in cams[:8]]
        return sum(vals) / len(vals)

    print(f"iter 0: N={tr.model._xyz.shape[0]} PSNR={eval_psnr():.2f} dB")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, args.iters + 1):
        tr.train_step()
        if it % 250 == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"iter {it}: N={tr.model._xyz.shape[0]} loss={float(tr.losses['total']):.4f} PSNR={eval_psnr():.2f} dB "
                  f"({it / dt:.1f} it/s incl. densify/eval)")
    assert all(torch.isfinite(getattr(tr.model, a)).all() for a in ["_xyz", "_scaling", "_rotation", "_opacity"])


if __name__ == "__main__":
    main()
