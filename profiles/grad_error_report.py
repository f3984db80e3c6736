"""Gradient error table of the HIP rasterizer against the fp64 autograd oracle on the parity-test cases: max-norm relative
error and element-wise error quantiles (tests/util.py::grad_stats) per parameter.  Needs a GPU:
    python profiles/grad_error_report.py > profiles/r2_grad_error_table.txt"""
import sys

import torch

sys.path.insert(0, '.')
from tests import util  # noqa: E402
import tests.test_raster_parity_gpu as T  # noqa: E402

device = torch.device('cuda:0')
print("# HIP (fp32) vs oracle fp64, and -- as the yardstick -- the SAME oracle run in fp32 vs itself in fp64:")
print("# case (n, W, H, focal, scale_mult, sem) | param | HIP: maxnorm | p99 | p99.9 | max || oracle-fp32: maxnorm | p99 | p99.9 | max | flipped pixels HIP / oracle-fp32 / budget")
for case in T.CASES:
    n, W, H, f, sm, sem = case
    for nd in (0, 2):
        cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
        bg = torch.tensor([0.2, 0.1, 0.4])
        g = torch.Generator().manual_seed(11)
        (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True, num_dist=nd)
        wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        (ref * wgt).sum().backward()
        (out, _), hl = util.hip_forward(cam, inp, dirs, bg, device, requires_grad=True, num_dist=nd)
        (out * wgt.float().to(device)).sum().backward()
        bad = util.bad_pixels(out, ref)
        (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True, num_dist=nd)
        (o32 * wgt.float()).sum().backward()
        bad32 = util.bad_pixels(o32, ref)
        for k in ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d", "sem"]:
            if rl[k] is None:
                continue
            st = util.grad_stats(hl[k].grad, rl[k].grad)
            s32 = util.grad_stats(l32[k].grad, rl[k].grad)
            print(f"{case} nd={nd} | {k:8s} | {st['maxnorm']:.1e} | {st['p99']:.1e} | {st['p999']:.1e} | {st['max']:.1e} || "
                  f"{s32['maxnorm']:.1e} | {s32['p99']:.1e} | {s32['p999']:.1e} | {s32['max']:.1e} | {bad} / {bad32} / {util.pixel_budget(ref)}",
                  flush=True)
