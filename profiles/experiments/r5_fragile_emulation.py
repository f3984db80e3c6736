"""CPU experiment (round 5): how much of the error of an fp32 evaluation of the rasterizer's gradients is made of FLIPPED
DECISIONS (alpha >= 1/255, power <= 0, T' < 1e-4, depth order of list neighbours) rather than of arithmetic?
The fp64 oracle marks the Gaussians that sit under a decision whose test quantity is within K unit roundoffs (x the magnitude
of what fp32 rounds) of its threshold (oracle/raster_torch.py::_mark_fragile); printed: tests/util.py::grad_stats of the fp32
oracle against the fp64 oracle over ALL Gaussians and over the non-fragile ones, for K = 4, 16, 64.
    python profiles/experiments/r5_fragile_emulation.py > profiles/r5_fragile_emulation.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import raster_torch as OR  # noqa: E402
from tests import util  # noqa: E402

print("# K | case | fragile / N | tensor | fp32-oracle error on all: maxnorm p99 p99.9 | on the non-fragile Gaussians")
for K in ([float(x) for x in sys.argv[1:]] or [4.0, 16.0, 64.0]):
    OR.FRAGILE_K = K
    for case in [(3000, 96, 64, 80.0, 6.0, 0), (1500, 100, 70, 90.0, 10.0, 2), (10000, 256, 256, 221.7, 3.0, 0)]:
        n, W, H, f, sm, sem = case
        cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
        bg = torch.tensor([0.2, 0.1, 0.4])
        g = torch.Generator().manual_seed(11)
        (ref, _, st), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True, fragile=True)
        fr = st["fragile"]
        wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        (ref * wgt).sum().backward()
        (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True)
        (o32 * wgt.float()).sum().backward()
        for k in ["means3D", "opac", "scales", "rots", "m2", "shs"]:
            a, b = util.grad_stats(l32[k].grad, rl[k].grad), util.grad_stats(l32[k].grad[~fr], rl[k].grad[~fr])
            print(f"{K:4.0f} | {case} | {int(fr.sum())} / {n} | {k:8s} | {a['maxnorm']:.1e} {a['p99']:.1e} {a['p999']:.1e} | "
                  f"{b['maxnorm']:.1e} {b['p99']:.1e} {b['p999']:.1e}", flush=True)
