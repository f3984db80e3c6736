"""Where does the HIP gradient error come from: the compositing backward or the projection backward?  (round 5)

For the small parity cases the fp64 autograd oracle also yields the gradients w.r.t. its SCREEN-SPACE intermediates (pixel
centre, conic, opacity, colour, centre depth, plane offset) -- exactly what the compositing backward accumulates per Gaussian
in its GradRec before the projection backward maps them to the parameters.  Compared, with the figures of
tests/util.py::grad_stats, against the fp64 oracle:
    hip    the GradRec of the HIP backward (vcr_debug_keep_sgrad), constants of GradRec::finish() applied
    yard   the same intermediates' gradients of the oracle evaluated in fp32
and, for the parameter gradients, `hip` / `yard` as in profiles/grad_ratio_table.py.  If the ratio hip / yard is ~1 at the
screen-space stage and > 1 at the parameters, the excess is the projection backward's; if it is > 1 already at the screen-space
stage, it is the compositing backward's (or the forward state both share).
    python profiles/grad_stage_errors.py > profiles/r5_grad_stage_errors.txt        (everything on ONE machine: the activated inputs
    of the oracle and of the HIP path must be bit-identical, and libm differs between hosts in the last bit)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raster_torch as OR  # noqa: E402
from tests import util  # noqa: E402
import tests.test_raster_parity_gpu as T  # noqa: E402
from vcr_gaus_amd import _lib  # noqa: E402

device = torch.device("cuda:0")
lib = _lib.load()
KEEP = {}
_pre = OR.preprocess


def preprocess(*a, **k):
    pre = _pre(*a, **k)
    for key in ("px", "py", "conic", "opacity", "rgb", "depth", "plane"):
        if pre[key].requires_grad:
            pre[key].retain_grad()
    KEEP["pre"] = pre
    return pre


OR.preprocess = preprocess


def screen_grads(pre):
    g = lambda k: pre[k].grad
    vis = pre["vis"]
    out = dict(px=g("px"), py=g("py"), cA=g("conic")[:, 0], cB=g("conic")[:, 1], cC=g("conic")[:, 2], opacity=g("opacity"),
               rgb=g("rgb"), depth=g("depth"), plane=g("plane"))
    return {k: (None if v is None else v.detach()) for k, v in out.items()}, vis


LN2 = 0.6931471805599453
print("# case | stage | quantity | hip maxnorm / p99 / p99.9 | yardstick (oracle fp32) | ratio")
for case in T.CASES:
    n, W, H, f, sm, sem = case
    cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
    bg = torch.tensor([0.2, 0.1, 0.4])
    g = torch.Generator().manual_seed(11)
    (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True)
    wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * wgt).sum().backward()
    s64, vis = screen_grads(KEEP["pre"])
    (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True)
    (o32 * wgt.float()).sum().backward()
    s32, _ = screen_grads(KEEP["pre"])
    _lib.check(lib.vcr_debug_keep_sgrad(1))
    (out, _), hl = util.hip_forward(cam, inp, dirs, bg, device, requires_grad=True)
    (out * wgt.float().to(device)).sum().backward()
    torch.cuda.synchronize()
    rec = torch.empty(n, 16, dtype=torch.float32)
    _lib.check(lib.vcr_debug_read_sgrad(ctypes.c_void_p(rec.data_ptr()), n))
    _lib.check(lib.vcr_debug_keep_sgrad(0))
    op = inp["opac"].reshape(-1).float()
    # GradRec::finish (csrc/vcr_common.h): raw sums -> gradients
    hip = dict(px=rec[:, 0] * LN2, py=rec[:, 1] * LN2, cA=rec[:, 4] * -0.5, cC=rec[:, 5] * -0.5, cB=-rec[:, 6],
               opacity=torch.where(op > 0, rec[:, 7] / op, torch.zeros_like(op)), rgb=rec[:, 8:11], depth=rec[:, 11], plane=rec[:, 12])
    for k in ("px", "py", "cA", "cB", "cC", "opacity", "rgb", "depth", "plane"):
        if s64[k] is None:
            continue
        a64 = s64[k][vis]
        st, sy = util.grad_stats(hip[k][vis], a64), util.grad_stats(s32[k][vis], a64)
        ratio = [st[q] / max(sy[q], fl) for q, fl in zip(("maxnorm", "p99", "p999"), (2e-5, 2e-5, 2e-4))]
        print(f"{case} | screen | {k:8s} | {st['maxnorm']:.1e} {st['p99']:.1e} {st['p999']:.1e} | {sy['maxnorm']:.1e} {sy['p99']:.1e} {sy['p999']:.1e} | "
              f"{ratio[0]:.2f} {ratio[1]:.2f} {ratio[2]:.2f}", flush=True)
    for k in ("means3D", "opac", "scales", "rots", "m2"):
        st, sy = util.grad_stats(hl[k].grad, rl[k].grad), util.grad_stats(l32[k].grad, rl[k].grad)
        ratio = [st[q] / max(sy[q], fl) for q, fl in zip(("maxnorm", "p99", "p999"), (2e-5, 2e-5, 2e-4))]
        print(f"{case} | params | {k:8s} | {st['maxnorm']:.1e} {st['p99']:.1e} {st['p999']:.1e} | {sy['maxnorm']:.1e} {sy['p99']:.1e} {sy['p999']:.1e} | "
              f"{ratio[0]:.2f} {ratio[1]:.2f} {ratio[2]:.2f}", flush=True)
