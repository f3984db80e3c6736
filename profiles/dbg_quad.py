import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import util
from vcr_gaus_amd.rasterizer import RasterOptions
dev=torch.device("cuda:0")
cam, inp, dirs = util.make_case(3000, 96, 64, 80.0, seed=31, scale_mult=6.0)
bg = torch.tensor([0.3, 0.2, 0.1])
for un in (True, False):
    outs=[]
    for ql in (False, True):
        (out, radii), hl = util.hip_forward(cam, inp, dirs, bg, dev, options=RasterOptions(quad_lists=ql), use_normals=un)
        outs.append(out.detach().cpu())
    d=(outs[0]-outs[1]).abs()
    print("use_normals", un, "split env", os.environ.get("VCR_SPLIT_SLOTS"), "max", float(d.max()), "pixels differing", int((d.amax(0)>0).sum()), [round(float(d[c].max()),8) for c in range(8)])
    ys,xs=torch.nonzero(d.amax(0)>0, as_tuple=True)
    for y,x in list(zip(ys.tolist(), xs.tolist()))[:6]:
        print("   px", y, x, "depth tile/quad", float(outs[0][3,y,x]), float(outs[1][3,y,x]), "alpha", float(outs[0][7,y,x]))
