"""CPU experiment (round 5): WHICH part of the compositing backward makes the HIP geometry gradients 2-6x noisier than a plain
fp32 evaluation of the oracle?  The blending of oracle/raster_torch.py::composite_tile is replaced by an explicit autograd
Function (everything else -- projection, conic, SH -- stays torch autograd in the working precision) whose backward is one of

  auto     torch autograd through cumprod (what the yardstick is)
  div      back to front, T_i = T_{i+1} / (1 - alpha_i), suffix sum accumulated back to front   (the HIP kernels of rounds 1-4)
  keep     back to front with the forward's own T_i (kept), suffix accumulated back to front
  fwd      front to back: T by multiplication, suffix = Total - prefix                             (per-splat pipelines)
  chunk    chunks of 64 walked back to front; inside a chunk front to back from the chunk's stored T,
           suffix = (suffix behind the chunk) + (chunk total - prefix inside the chunk)

Each is run in fp32 against the fp64 autograd oracle on the small parity cases; printed are tests/util.py::grad_stats figures
and their ratio to `auto`.
    python profiles/experiments/r5_bwd_algorithm_emulation.py > profiles/r5_bwd_algorithm_emulation.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import raster_torch as OR  # noqa: E402
from tests import util  # noqa: E402

MODE = "auto"


class Blend(torch.autograd.Function):
    """out[P,C] = sum_l w[P,l] F[P,l,C], w = a * T_excl on `contrib`, 0 elsewhere (the stop rule is decided by the caller)."""

    @staticmethod
    def forward(ctx, a, F, contrib, mode):
        om = torch.where(contrib, 1.0 - a, torch.ones_like(a))
        T_incl = torch.cumprod(om, 1)
        T_excl = torch.cat([torch.ones_like(a[:, :1]), T_incl[:, :-1]], 1)
        w = torch.where(contrib, a * T_excl, torch.zeros_like(a))
        out = torch.einsum("pl,plc->pc", w, F)
        ctx.save_for_backward(a, F, contrib, T_excl, T_incl[:, -1].clone(), out)
        ctx.mode = mode
        return out, T_incl[:, -1]

    @staticmethod
    def backward(ctx, g, gT):
        a, F, contrib, T_excl, Tf, out = ctx.saved_tensors
        mode = ctx.mode
        P, L = a.shape
        fg = torch.einsum("plc,pc->pl", F, g)                      # f_i . g per (pixel, entry)
        ae = torch.where(contrib, a, torch.zeros_like(a))
        inv = 1.0 / (1.0 - ae)
        bg_term = Tf * gT                                          # d(out_bg)/dT_final * T_final (the background term)
        da = torch.zeros_like(a)
        w = torch.zeros_like(a)
        if mode in ("div", "keep", "divn", "anchor"):
            T = Tf.clone()
            B = bg_term.clone()
            if mode in ("divn", "anchor"):         # v_rcp_f32: up to 1 ulp off -- modelled as uniform +-1 ulp relative noise on the reciprocal
                gen = torch.Generator().manual_seed(5)
                inv = inv * (1.0 + (torch.rand(inv.shape, generator=gen, dtype=inv.dtype) * 2 - 1) * 1.19e-7)
            for i in range(L - 1, -1, -1):
                if mode in ("div", "divn", "anchor"):
                    T = T * inv[:, i]
                    if mode == "anchor" and i % 64 == 0:
                        T = T_excl[:, i]                # re-anchored on the forward's checkpoint at every chunk boundary
                else:
                    T = T_excl[:, i]
                wi = ae[:, i] * T
                da[:, i] = T * fg[:, i] - B * inv[:, i]
                B = B + wi * fg[:, i]
                w[:, i] = wi
        elif mode == "fwd":
            total = (out * g).sum(1) + bg_term
            T = torch.ones_like(Tf)
            pre = torch.zeros_like(Tf)
            for i in range(L):
                wi = ae[:, i] * T
                pre = pre + wi * fg[:, i]
                da[:, i] = T * fg[:, i] - (total - pre) * inv[:, i]
                w[:, i] = wi
                T = T * (1.0 - ae[:, i])
        elif mode == "chunk":
            CH = 64
            nch = (L + CH - 1) // CH
            B_behind = bg_term.clone()
            for c in range(nch - 1, -1, -1):
                lo, hi = c * CH, min(L, (c + 1) * CH)
                T = T_excl[:, lo].clone()                          # the forward's checkpoint at the chunk boundary
                ws = []
                for i in range(lo, hi):
                    wi = ae[:, i] * T
                    ws.append(wi)
                    w[:, i] = wi
                    T = T * (1.0 - ae[:, i])
                ctot = torch.zeros_like(Tf)
                for k, i in enumerate(range(lo, hi)):
                    ctot = ctot + ws[k] * fg[:, i]
                T = T_excl[:, lo].clone()
                pre = torch.zeros_like(Tf)
                for k, i in enumerate(range(lo, hi)):
                    pre = pre + ws[k] * fg[:, i]
                    da[:, i] = T * fg[:, i] - (B_behind + (ctot - pre)) * inv[:, i]
                    T = T * (1.0 - ae[:, i])
                B_behind = B_behind + ctot
        da = torch.where(contrib, da, torch.zeros_like(da))
        dF = w[:, :, None] * g[:, None, :]
        return da, dF, None, None


_orig = OR.composite_tile


def composite_tile(s, pre, idx, x0, y0, means2D_densify, dirs, num_sem, num_dist=0, fragile=None):
    if MODE == "auto" or idx.numel() == 0:
        return _orig(s, pre, idx, x0, y0, means2D_densify, dirs, num_sem, num_dist, fragile=fragile)
    assert num_dist == 0
    dt = pre["px"].dtype
    H, W = s.image_height, s.image_width
    ys, xs = torch.meshgrid(torch.arange(y0, min(y0 + OR.TILE, H)), torch.arange(x0, min(x0 + OR.TILE, W)), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    npix = xs.numel()
    L = idx.numel()
    xy = torch.stack([pre["px"][idx], pre["py"][idx]], -1)
    holder = means2D_densify[idx, :2] if means2D_densify is not None else torch.zeros(L, 2, dtype=dt)
    xye = OR._AbsGradExpand.apply(xy, holder, npix, 0.5 * W, 0.5 * H)
    dx = xye[:, :, 0] - xs.to(dt)[:, None]
    dy = xye[:, :, 1] - ys.to(dt)[:, None]
    con = pre["conic"][idx]
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    araw = pre["opacity"][idx][None] * torch.exp(torch.clamp(power, max=0.0))
    alpha = araw + (torch.clamp(araw, max=OR.ALPHA_MAX) - araw).detach()
    valid = (power <= 0) & (alpha >= OR.ALPHA_MIN)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    om = 1.0 - a_eff.detach()
    T_incl = torch.cumprod(om, dim=1)
    stop = valid & (T_incl < OR.T_EPS)
    stopped = torch.cumsum(stop.to(torch.int32), 1) > 0
    contrib = valid & ~stopped
    z = pre["depth"][idx]
    if dirs is not None:
        r = dirs.to(dt)[:, ys, xs].t()
        n = pre["normal"][idx]
        den = r @ n.t()
        use = den > OR.PLANE_EPS
        dsafe = torch.where(use, den, torch.ones_like(den))
        dpl = pre["plane"][idx][None] / dsafe * r[:, 2:3]
        dep = torch.where(use, dpl, z[None].expand(npix, -1))
    else:
        dep = z[None].expand(npix, -1)
    feats = [pre["rgb"][idx][None].expand(npix, -1, -1), dep[:, :, None], pre["normal"][idx][None].expand(npix, -1, -1),
             torch.ones(npix, L, 1, dtype=dt)]
    if num_sem:
        feats.append(pre["sem"][idx].to(dt)[None].expand(npix, -1, -1))
    F = torch.cat(feats, 2)
    out, Tfin = Blend.apply(a_eff, F, contrib, MODE)
    return xs, ys, out, Tfin, contrib, None


OR.composite_tile = composite_tile

KEYS = ["means3D", "shs", "normals", "opac", "scales", "rots", "m2", "m2d"]
CASES = [(3000, 96, 64, 80.0, 6.0, 0), (1500, 100, 70, 90.0, 10.0, 2), (10000, 256, 256, 221.7, 3.0, 0)]
if len(sys.argv) > 1:
    CASES = CASES[:int(sys.argv[1])]
print("# case | mode | tensor | maxnorm p99 p99.9 | ratio to auto-fp32")
for case in CASES:
    n, W, H, f, sm, sem = case
    cam, inp, dirs = util.make_case(n, W, H, f, seed=7, scale_mult=sm, sem=sem)
    bg = torch.tensor([0.2, 0.1, 0.4])
    g = torch.Generator().manual_seed(11)
    MODE = "auto"
    (ref, _, _), rl = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float64, requires_grad=True)
    wgt = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * wgt).sum().backward()
    base = {}
    for mode in ("auto", "divn", "anchor"):
        MODE = mode
        (o32, _, _), l32 = util.oracle_forward(cam, inp, dirs, bg, dtype=torch.float32, requires_grad=True)
        (o32 * wgt.float()).sum().backward()
        for k in KEYS:
            if rl[k] is None:
                continue
            st = util.grad_stats(l32[k].grad, rl[k].grad)
            if mode == "auto":
                base[k] = st
            b = base[k]
            ratio = [st[q] / max(b[q], fl) for q, fl in zip(("maxnorm", "p99", "p999"), (2e-5, 2e-5, 2e-4))]
            print(f"{case} | {mode:5s} | {k:8s} | {st['maxnorm']:.1e} {st['p99']:.1e} {st['p999']:.1e} | {ratio[0]:.2f} {ratio[1]:.2f} {ratio[2]:.2f}", flush=True)
