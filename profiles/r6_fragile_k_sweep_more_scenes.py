import sys, torch
sys.path.insert(0,'/root/repo')
from oracle import raster_torch as OR
from tests import util
cases=[(4000,160,128,140.0,3.0,3),(3000,96,64,80.0,6.0,7),(2500,120,72,90.0,6.0,44)]
for case in cases:
    n,W,H,f,sm,seed=case
    cam, inp, dirs = util.make_case(n, W, H, f, seed=seed, scale_mult=sm)
    bg = torch.tensor([0.2, 0.1, 0.4])
    (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True)
    wgt = torch.randn(o32.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    (o32 * wgt.float()).sum().backward()
    for K in (2,4,6,8,12,16):
        OR.FRAGILE_K=float(K)
        (ref, _, st), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True)
        fr = rl["fragile"]
        (ref * wgt).sum().backward()
        worst_all = worst_nf = 0.0; wk=None
        for k in ["means3D", "opac", "scales", "rots", "m2", "shs"]:
            worst_all = max(worst_all, util.grad_stats(l32[k].grad, rl[k].grad)["maxnorm"])
            v=util.grad_stats(l32[k].grad[~fr], rl[k].grad[~fr])["maxnorm"]
            if v>worst_nf: worst_nf=v; wk=k
        print(case, "K",K,"fragile %.3f"%float(fr.float().mean()),"worst_all %.2e worst_nf %.2e (%s)"%(worst_all,worst_nf,wk),flush=True)
