"""HIP gradient error over the oracle-fp32 yardstick, per tensor, on the small parity cases -- for the library VCR_LIB points
to (default build, or the test-only deterministic-backward build `make -C vcr_gaus_amd/csrc det`).
For every case: the fp64 autograd oracle is the reference; `yard` = error of the SAME oracle evaluated in fp32; `hip` = error
of the HIP path; ratio = hip / yard for the three figures of tests/util.py::grad_stats (max-norm, element-wise p99, p99.9).
Also: are two HIP backward runs bit-identical (they are in the deterministic build, not with free-running fp32 atomics)?
    VCR_LIB=<library> python profiles/grad_ratio_table.py > profiles/r5_grad_ratio_table_<build>.txt"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402
import tests.test_raster_parity_gpu as T  # noqa: E402
from vcr_gaus_amd import _lib  # noqa: E402

device = torch.device("cuda:0")
ORACLE_ONLY = not torch.cuda.is_available()      # on a box without a GPU the run only fills the oracle cache
KEYS = ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d", "sem"]
print(f"# library: {os.path.basename(_lib.LIB_PATH)}")
print("# case | tensor | hip maxnorm / p99 / p99.9 | yardstick (oracle fp32) maxnorm / p99 / p99.9 | ratio | two runs bit-identical | flipped pixels hip / oracle-fp32")
worst = {}
for case in T.CASES:
    n, W, H, f, sm, sem = case
    for nd in (0, 2):
        cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
        bg = torch.tensor([0.2, 0.1, 0.4])
        g = torch.Generator().manual_seed(11)
        # the CPU oracle's results are cached (VCR_ORACLE_CACHE, default build/oracle_cache -- git-ignored, travels to the GPU box) so that several builds of the
        # library can be tabulated in one session without re-running it
        cdir = os.environ.get("VCR_ORACLE_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "oracle_cache"))
        os.makedirs(cdir, exist_ok=True)
        cfile = os.path.join(cdir, "ratio_%s_nd%d.pt" % ("_".join(str(c) for c in case), nd))
        cached = torch.load(cfile) if os.path.exists(cfile) else None
        if cached is None:
            (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True, num_dist=nd)
            wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
            (ref * wgt).sum().backward()
        else:
            ref, wgt = cached["ref"], cached["wgt"]
            rl = {k: (None if v is None else types.SimpleNamespace(grad=v)) for k, v in cached["rl"].items()}
        runs = []
        for rep in range(0 if ORACLE_ONLY else 2):
            (out, _), hl = util.hip_forward(cam, inp, dirs, bg, device, requires_grad=True, num_dist=nd)
            (out * wgt.float().to(device)).sum().backward()
            runs.append(hl)
        if not ORACLE_ONLY:
            hl = runs[0]
            bad = util.bad_pixels(out, ref)
        if cached is None:
            (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True, num_dist=nd)
            (o32 * wgt.float()).sum().backward()
            torch.save(dict(ref=ref.detach(), wgt=wgt, o32=o32.detach(),
                            rl={k: (None if (v is None or not torch.is_tensor(v) or v.grad is None) else v.grad) for k, v in rl.items()},
                            l32={k: (None if (v is None or not torch.is_tensor(v) or v.grad is None) else v.grad) for k, v in l32.items()}), cfile)
        else:
            o32 = cached["o32"]
            l32 = {k: (None if v is None else types.SimpleNamespace(grad=v)) for k, v in cached["l32"].items()}
        bad32 = util.bad_pixels(o32, ref)
        if ORACLE_ONLY:
            print("cached", case, nd, flush=True)
            continue
        for k in KEYS:
            if rl.get(k) is None:
                continue
            st, s32 = util.grad_stats(hl[k].grad, rl[k].grad), util.grad_stats(l32[k].grad, rl[k].grad)
            same = bool(torch.equal(runs[0][k].grad, runs[1][k].grad))
            ratio = [st[q] / max(s32[q], fl) for q, fl in zip(("maxnorm", "p99", "p999"), (2e-5, 2e-5, 2e-4))]
            w = worst.setdefault(k, [0.0, 0.0, 0.0])
            for i in range(3):
                w[i] = max(w[i], ratio[i])
            print(f"{case} nd={nd} | {k:8s} | {st['maxnorm']:.1e} {st['p99']:.1e} {st['p999']:.1e} | {s32['maxnorm']:.1e} {s32['p99']:.1e} "
                  f"{s32['p999']:.1e} | {ratio[0]:.2f} {ratio[1]:.2f} {ratio[2]:.2f} | {'yes' if same else 'no'} | {bad} / {bad32}", flush=True)
print("# worst ratio per tensor (max-norm, p99, p99.9; yardstick floored at 2e-5 / 2e-5 / 2e-4 as in tests/util.py):")
for k, w in worst.items():
    print(f"#   {k:8s} {w[0]:.2f} {w[1]:.2f} {w[2]:.2f}")
