"""Per-parameter gradient error of the HIP rasterizer against the fp64 autograd oracle on the parity-test cases
(max-norm relative error; the number DESIGN.md section 5 quotes).  Needs a GPU: python profiles/grad_error_report.py"""
import sys, torch
sys.path.insert(0, '.')
from tests import util
import tests.test_raster_parity_gpu as T
device = torch.device('cuda:0')
cases = [c.values if hasattr(c, 'values') else c for c in T.CASES] if hasattr(T, 'CASES') else None
print("cases:", cases)
for case in cases:
    n, W, H, f, sm, sem = case
    cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
    bg = torch.tensor([0.2, 0.1, 0.4])
    g = torch.Generator().manual_seed(11)
    (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True)
    wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * wgt).sum().backward()
    (out, _), hl = util.hip_forward(cam, inp, dirs, bg, device, requires_grad=True)
    (out * wgt.float().to(device)).sum().backward()
    errs = {k: util.rel_err(hl[k].grad, rl[k].grad) for k in ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d", "sem"] if rl[k] is not None}
    print(case, {k: f"{v:.1e}" for k, v in errs.items()})
