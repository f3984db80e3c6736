"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement (PyTorch, autograd-differentiable, fp32 or fp64) of the tile rasterizer the
reference binds as `diff_gaussian_rasterization` (call sites
`gaussian_renderer/__init__.py:43-59,107-123,332-344,550-562`).

PARITY UNPINNED: the reference's CUDA rasterizer source is not vendored
(`.gitmodules:4-6`, `submodules/diff-gaussian-rasterization/` is empty, no pinned SHA) and the
reference holds no tests / golden vectors for it.  This file therefore states the algorithm by
contract: the public Inria 3DGS rasterizer pipeline, extended with the
channels the reference's call sites consume (depth, camera-space normal, alpha, semantics), and
the build decisions U1..U8 listed in DESIGN.md.  What IS pinned against the reference:
  * the colour rule `clamp_min(eval_sh + 0.5, 0)` (`gaussian_renderer/__init__.py:83-87`,
    `tools/sh_utils.py:57-112`) -- golden vectors in tests/golden/sh_*.npz,
  * covariance from scale/rotation (`tools/general_utils.py:98-130`, `scene/gaussian_model.py:38-42`),
  * camera matrix conventions (`scene/cameras.py:62-73`) -- golden vectors in tests/golden/cam_*.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Gradients come from torch autograd through the vectorised forward, i.e. they are independent of
the hand-derived adjoints in the HIP kernels.
"""
import math
from typing import NamedTuple, Optional

import torch

TILE = 16
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4
NEAR_CULL = 0.2
LOWPASS = 0.3
PLANE_EPS = 1e-4  # U5: ray/plane denominators below this fall back to the centre depth

# tools/sh_utils.py:24-52
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class Settings(NamedTuple):
    """Mirror of GaussianRasterizationSettings (`gaussian_renderer/__init__.py:43-57`)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False
    f_count: int = 0


def eval_sh(deg, shs, dirs):
    """shs [N,K,3] (coefficient-major, `scene/gaussian_model.py:139-142`), dirs [N,3] unit.
    Polynomial basis of `tools/sh_utils.py:57-112` (deg <= 3)."""
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                   + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9]
                       + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return res


def quat_to_rotmat(q):
    """`tools/general_utils.py:105-118` on an already-normalised (w,x,y,z) quaternion."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales, mod, q):
    """Sigma = (R S)(R S)^T, `tools/general_utils.py:121-130`, `scene/gaussian_model.py:38-42`."""
    R = quat_to_rotmat(q)
    L = R * (scales * mod)[:, None, :]
    return L @ L.transpose(1, 2)


def cov3d_from_packed(c6):
    """upper-triangular (xx,xy,xz,yy,yz,zz), `tools/general_utils.py:84-93`."""
    xx, xy, xz, yy, yz, zz = c6.unbind(-1)
    return torch.stack([xx, xy, xz, xy, yy, yz, xz, yz, zz], dim=-1).view(-1, 3, 3)


class _AbsGradExpand(torch.autograd.Function):
    """xy [L,2] -> [P,L,2].  Backward sends the plain sum to `xy` and the sum of absolute
    per-pixel gradients to the densification holder (build decision U3)."""

    @staticmethod
    def forward(ctx, xy, holder, npix, sx, sy):
        ctx.sc = (sx, sy)
        return xy.unsqueeze(0).expand(npix, -1, -1).clone()

    @staticmethod
    def backward(ctx, g):
        sc = torch.tensor(ctx.sc, dtype=g.dtype)     # pixel -> NDC units (0.5*W, 0.5*H)
        return g.sum(0), g.abs().sum(0) * sc, None, None, None


def preprocess(s: Settings, means3D, means2D, shs, colors_precomp, normals_precomp,
               semantics_precomp, opacities, scales, rotations, cov3D_precomp):
    """Per-Gaussian stage (K1).  Returns dict of per-Gaussian screen-space quantities."""
    dt = means3D.dtype
    H, W = s.image_height, s.image_width
    N = means3D.shape[0]
    V = s.viewmatrix.to(dt)
    P = s.projmatrix.to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ V                      # row-vector convention, scene/cameras.py:68-70
    t = pv[:, :3]
    pc = ph @ P
    pw = 1.0 / (pc[:, 3:4] + 1e-7)
    ndc = pc[:, :3] * pw
    front = t[:, 2] > NEAR_CULL

    if cov3D_precomp is not None:
        S3 = cov3d_from_packed(cov3D_precomp.to(dt))
    else:
        S3 = cov3d_from_scale_rot(scales, s.scale_modifier, rotations)

    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    tz = torch.where(front, t[:, 2], torch.ones_like(t[:, 2]))
    u = torch.clamp(t[:, 0] / tz, -1.3 * s.tanfovx, 1.3 * s.tanfovx)
    v = torch.clamp(t[:, 1] / tz, -1.3 * s.tanfovy, 1.3 * s.tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * u / tz,
                     zero, fy / tz, -fy * v / tz], -1).view(-1, 2, 3)
    Rv = V[:3, :3].t()               # world -> view rotation (column-vector form)
    M = J @ Rv                       # [N,2,3]
    c2 = M @ S3 @ M.transpose(1, 2)
    a = c2[:, 0, 0] + LOWPASS
    b = c2[:, 0, 1]
    c = c2[:, 1, 1] + LOWPASS
    det = a * c - b * b
    ok = front & (det != 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / dets, -b / dets, a / dets], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach().clamp(min=0)))
    radius = torch.where(ok, radius, torch.zeros_like(radius))

    xy_ndc = ndc[:, :2] + means2D[:, :2]      # means2D is the zero grad-holder
    px = ((xy_ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((xy_ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pxd, pyd = px.detach(), py.detach()
    xmin = torch.clamp(torch.floor((pxd - radius) / TILE), 0, gx).long()
    xmax = torch.clamp(torch.floor((pxd + radius + TILE - 1) / TILE), 0, gx).long()
    ymin = torch.clamp(torch.floor((pyd - radius) / TILE), 0, gy).long()
    ymax = torch.clamp(torch.floor((pyd + radius + TILE - 1) / TILE), 0, gy).long()
    tiles = (xmax - xmin) * (ymax - ymin)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    vis = tiles > 0
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp.to(dt)
    else:
        d = means3D - s.campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(s.sh_degree, shs, d) + 0.5, 0.0)

    if normals_precomp is not None:
        nrm = normals_precomp.to(dt)
        plane = (nrm * t).sum(-1)
    else:
        nrm = torch.zeros(N, 3, dtype=dt)
        plane = torch.zeros(N, dtype=dt)

    return dict(px=px, py=py, depth=t[:, 2], conic=conic, opacity=opacities.reshape(-1),
                rgb=rgb, normal=nrm, plane=plane, sem=semantics_precomp, radii=radii,
                xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, tiles=tiles, vis=vis,
                grid=(gx, gy))


def bin_and_sort(pre):
    """K2..K5: (tile, depth, index)-ordered instance list + per-tile ranges."""
    gx, gy = pre["grid"]
    ids = torch.nonzero(pre["vis"]).squeeze(1)
    counts = pre["tiles"][ids]
    R = int(counts.sum())
    owner = torch.repeat_interleave(ids, counts)
    start = torch.cumsum(counts, 0) - counts
    local = torch.arange(R) - torch.repeat_interleave(start, counts)
    w = (pre["xmax"] - pre["xmin"])[owner]
    tx = pre["xmin"][owner] + local % w
    ty = pre["ymin"][owner] + local // w
    tile = ty * gx + tx
    o1 = torch.sort(pre["depth"].detach()[owner], stable=True).indices   # ties -> lower index first
    owner, tile = owner[o1], tile[o1]
    o2 = torch.sort(tile, stable=True).indices
    owner, tile = owner[o2], tile[o2]
    nt = gx * gy
    cnt = torch.bincount(tile, minlength=nt)
    end = torch.cumsum(cnt, 0)
    beg = end - cnt
    return owner, beg, end, R


# multiples of the fp32 unit roundoff (2^-24) x the magnitude of what is being rounded, see _mark_fragile.  Round 6 sweeps (fp32 oracle
# against fp64 oracle): K = 8 is the smallest value at which the error on the UNMARKED Gaussians has plateaued on every swept scene
# -- c1 plateaus at K = 4 (rotations 2.5e-5 at K = 1, 2; 1.6e-5 from 4 on: profiles/r6_fragile_k_sweep.txt), a 4 000-Gaussian scene
# at 160 x 128 only at K = 8 (5.3e-4 up to K = 6, 7.3e-5 from 8 on: profiles/r6_fragile_k_sweep_more_scenes.txt) -- and the marked
# share of c1 falls from the 19 % of round 5's K = 16 to 10 %.  VCR_TEST_FRAGILE_K overrides it for such sweeps (test infrastructure).
import os as _os
FRAGILE_K = float(_os.environ.get("VCR_TEST_FRAGILE_K", "8"))


@torch.no_grad()
def _mark_fragile(fragile, idx, con, dx, dy, power, opac, alpha, valid, a_eff, T_incl, stopped, contrib, z):
    """Which Gaussians of this tile sit on a DISCONTINUITY of the algorithm that an fp32 evaluation may resolve the other way?
    The rasterizer decides per (pixel, entry) -- power <= 0, alpha >= 1/255, T' < 1e-4 (stop) -- and per pair of list neighbours
    (depth order); each decision moves the result by a finite amount, so two correct fp32 implementations that round
    differently agree only up to those flips, however accurate their arithmetic is.  A decision is FRAGILE when its test
    quantity lies within the rounding-error bound of an fp32 evaluation of it:
        exponent   |ln(alpha / (1/255))| or |power| <  K u (|A dx^2|/2 + |C dy^2|/2 + |B dx dy| + |ln o| + 1)
        stop       |ln(T' / 1e-4)|                  <  K u sum_{j<=k} 1 / (1 - alpha_j)          (u = 2^-24)
        order      |z_k+1 - z_k|                    <  K u |z|
    A fragile alpha / power test at (pixel, k) marks k and everything that contributes behind it at that pixel (their
    transmittance changes by the factor 1 - alpha_k); a fragile stop marks the entry it decides about; a fragile order marks
    both neighbours.  `fragile` [N] bool is OR-ed in place.  Test infrastructure: lets the parity tests separate "rounds
    differently at a threshold" from "computes the gradient wrong" (tests/util.py)."""
    u = 2.0 ** -24
    K = FRAGILE_K
    mag = 0.5 * (con[None, :, 0].abs() * dx * dx + con[None, :, 2].abs() * dy * dy) + (con[None, :, 1] * dx * dy).abs()
    bound = K * u * (mag + opac.clamp_min(1e-30).log().abs()[None] + 1.0)
    lna = opac.clamp_min(1e-30).log()[None] + torch.clamp(power, max=0.0) - math.log(ALPHA_MIN)
    near_alpha = (lna.abs() < bound) & (power <= bound)
    near_pow = (power.abs() < bound) & (lna >= -bound)
    dec = (near_alpha | near_pow) & ~stopped                       # (a decision behind the stop changes nothing)
    om_inv = 1.0 / (1.0 - a_eff)
    bound_T = K * u * torch.cumsum(om_inv, 1)
    stop_here = valid & ~(torch.cumsum((valid & (T_incl < T_EPS)).to(torch.int32), 1) - (valid & (T_incl < T_EPS)).to(torch.int32) > 0)
    near_stop = stop_here & ((T_incl.clamp_min(1e-300) / T_EPS).log().abs() < bound_T)
    mark = near_stop.any(0)
    if bool(dec.any()):
        behind = (torch.cumsum(dec.to(torch.int32), 1) > 0) & (contrib | dec)      # the fragile entry and what contributes behind it
        mark = mark | behind.any(0)
    if z.numel() > 1:
        tie = (z[1:] - z[:-1]).abs() < K * u * z[1:].abs()
        if bool(tie.any()):
            both = torch.zeros_like(mark)
            both[1:] |= tie
            both[:-1] |= tie
            mark = mark | (both & (valid | contrib).any(0))
    fragile[idx[mark]] = True


@torch.no_grad()
def _fragile_footprint(s, pre):
    """Per-Gaussian decisions of the projection that an fp32 evaluation may round the other way: the radius ceil(3 sqrt(lambda))
    and the four floor()s of the tile rectangle.  A rectangle that differs by one tile changes which pixels the Gaussian is
    rendered at (it is composited wherever alpha >= 1/255 INSIDE its rectangle), a discrete difference like the ones of
    _mark_fragile.  -> bool [N]."""
    u, K = 2.0 ** -24, FRAGILE_K
    H, W = s.image_height, s.image_width
    r = pre["radii"].to(pre["px"].dtype)
    px, py = pre["px"].detach(), pre["py"].detach()
    vis = pre["vis"]
    out = torch.zeros_like(vis)
    # 3 sqrt(lambda) within rounding distance of an integer (the stored radius is its ceil): recompute lambda's root from the conic
    con = pre["conic"].detach()
    det_c = con[:, 0] * con[:, 2] - con[:, 1] * con[:, 1]                      # det(conic) = 1 / det(cov2D)
    ok = det_c > 0
    a, c = con[:, 2] / det_c.clamp_min(1e-300), con[:, 0] / det_c.clamp_min(1e-300)   # cov2D diagonal
    b = -con[:, 1] / det_c.clamp_min(1e-300)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - (a * c - b * b), min=0.1))
    q = 3.0 * torch.sqrt(lam.clamp_min(0))
    # (the fp32 error of lambda is amplified by the cancellation in a c - b^2 for needle-shaped footprints: scale the bound by it)
    amp = ((a * c).abs() + b * b) / (a * c - b * b).abs().clamp_min(1e-300)
    out |= ok & ((q - torch.round(q)).abs() < K * u * q * amp.clamp(1.0, 1e6))
    for centre, lim in ((px, W), (py, H)):
        for qq in ((centre - r) / TILE, (centre + r + TILE - 1) / TILE):
            inside = (qq > -1) & (qq < (lim + TILE - 1) // TILE + 1)               # (clamped away otherwise: no decision)
            out |= inside & ((qq - torch.round(qq)).abs() < K * u * (centre.abs() + r + TILE) / TILE)
    return out & vis


def composite_tile(s, pre, idx, x0, y0, means2D_densify, dirs, num_sem, num_dist=0, fragile=None):
    """K6 for one 16x16 tile, vectorised over [pixels, list].  `fragile`: optional [N] bool collector, see _mark_fragile."""
    dt = pre["px"].dtype
    H, W = s.image_height, s.image_width
    ys, xs = torch.meshgrid(torch.arange(y0, min(y0 + TILE, H)), torch.arange(x0, min(x0 + TILE, W)),
                            indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    npix = xs.numel()
    L = idx.numel()
    C = 8 + num_sem + num_dist
    if L == 0:
        out = torch.zeros(npix, C, dtype=dt)
        Tfin = torch.ones(npix, dtype=dt)
        return xs, ys, out, Tfin, None, None
    xy = torch.stack([pre["px"][idx], pre["py"][idx]], -1)
    holder = means2D_densify[idx, :2] if means2D_densify is not None else torch.zeros(L, 2, dtype=dt)
    xye = _AbsGradExpand.apply(xy, holder, npix, 0.5 * W, 0.5 * H)            # [P,L,2]
    dx = xye[:, :, 0] - xs.to(dt)[:, None]
    dy = xye[:, :, 1] - ys.to(dt)[:, None]
    con = pre["conic"][idx]
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    araw = pre["opacity"][idx][None] * torch.exp(torch.clamp(power, max=0.0))
    # min(0.99, .) with a straight-through gradient: the public 3DGS backward differentiates
    # alpha = o*G without regard to the clamp [UPSTREAM]; kept so gradients match that family.
    alpha = araw + (torch.clamp(araw, max=ALPHA_MAX) - araw).detach()
    valid = (power <= 0) & (alpha >= ALPHA_MIN)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    om = 1.0 - a_eff
    T_incl = torch.cumprod(om, dim=1)
    T_excl = torch.cat([torch.ones(npix, 1, dtype=dt), T_incl[:, :-1]], 1)
    stop = valid & (T_incl.detach() < T_EPS)
    stopped = torch.cumsum(stop.to(torch.int32), 1) > 0
    contrib = valid & ~stopped
    wgt = torch.where(contrib, a_eff * T_excl, torch.zeros_like(alpha))     # [P,L]
    Tfin = torch.where(contrib, om, torch.ones_like(om)).prod(1)
    if fragile is not None:
        _mark_fragile(fragile, idx, con, dx, dy, power, pre["opacity"][idx], alpha, valid, a_eff, T_incl, stopped, contrib,
                      pre["depth"][idx])

    z = pre["depth"][idx]
    if dirs is not None:
        r = dirs.to(dt)[:, ys, xs].t()                     # [P,3] unit rays
        n = pre["normal"][idx]                             # [L,3]
        den = r @ n.t()                                    # [P,L]
        use = den > PLANE_EPS
        dsafe = torch.where(use, den, torch.ones_like(den))
        dpl = pre["plane"][idx][None] / dsafe * r[:, 2:3]
        dep = torch.where(use, dpl, z[None].expand(npix, -1))
    else:
        dep = z[None].expand(npix, -1)

    feats = [pre["rgb"][idx], None, pre["normal"][idx], torch.ones(L, 1, dtype=dt)]
    out_rgb = wgt @ feats[0]
    out_d = (wgt * dep).sum(1, keepdim=True)
    out_n = wgt @ feats[2]
    out_a = wgt.sum(1, keepdim=True)
    outs = [out_rgb, out_d, out_n, out_a]
    if num_sem:
        outs.append(wgt @ pre["sem"][idx].to(dt))
    if num_dist == 2:                                  # depth moments (gaussian_renderer/__init__.py:155-157)
        outs += [out_d, (wgt * dep * dep).sum(1, keepdim=True)]
    if num_dist == 1:                                  # U2: 2DGS-form distortion of the mapped depth (near .01, far 100)
        md = -(100.0 / (100.0 - 0.01)) * 0.01 / dep          # mapped depth minus its constant term (shift-invariant)
        m1, m2 = (wgt * md).sum(1, keepdim=True), (wgt * md * md).sum(1, keepdim=True)
        outs.append(out_a * m2 - m1 * m1)
    out = torch.cat(outs, 1)
    return xs, ys, out, Tfin, contrib, wgt


def rasterize(s: Settings, means3D, means2D=None, means2D_densify=None, shs=None,
              colors_precomp=None, normals_precomp=None, semantics_precomp=None, opacities=None,
              scales=None, rotations=None, cov3D_precomp=None, dirs=None, inside=None, tile_stride=1,
              timings=None, num_dist=0, fragile=False):
    """Full forward (tile_stride>1 composites only every k-th tile: bounded CPU-baseline sample).  f_count==0: (out[C,H,W], radii).  f_count==1/2: (count, score, image, radii).
    f_count==3: (count, radii).  C = 8 + S (colour3, depth1, normal3, alpha1, sem S)."""
    dt = means3D.dtype
    N = means3D.shape[0]
    H, W = s.image_height, s.image_width
    if means2D is None:
        means2D = torch.zeros(N, 3, dtype=dt)
    num_sem = 0 if semantics_precomp is None else semantics_precomp.shape[1]
    import time as _time
    _t0 = _time.perf_counter()
    pre = preprocess(s, means3D, means2D, shs, colors_precomp, normals_precomp, semantics_precomp,
                     opacities, scales, rotations, cov3D_precomp)
    owner, beg, end, R = bin_and_sort(pre)
    _t1 = _time.perf_counter()
    gx, gy = pre["grid"]
    C = 8 + num_sem + num_dist
    img = torch.zeros(H, W, C, dtype=dt)
    Tmap = torch.ones(H, W, dtype=dt)
    count = torch.zeros(N, dtype=torch.int32)
    score = torch.zeros(N, dtype=dt)
    rows, cols, vals, tvals = [], [], [], []
    frag = torch.zeros(N, dtype=torch.bool) if fragile else None
    if fragile:
        frag |= _fragile_footprint(s, pre)
    for ty in range(gy):
        for tx in range(gx):
            t = ty * gx + tx
            if t % tile_stride:
                continue
            idx = owner[beg[t]:end[t]]
            xs, ys, out, Tfin, contrib, wgt = composite_tile(
                s, pre, idx, tx * TILE, ty * TILE, means2D_densify, dirs, num_sem, num_dist, fragile=frag)
            rows.append(ys); cols.append(xs); vals.append(out); tvals.append(Tfin)
            if s.f_count and contrib is not None:
                count.index_add_(0, idx, contrib.sum(0).to(torch.int32))
                score.index_add_(0, idx, wgt.detach().sum(0))
    rows, cols = torch.cat(rows), torch.cat(cols)
    img = img.index_put((rows, cols), torch.cat(vals))
    Tmap = Tmap.index_put((rows, cols), torch.cat(tvals))
    bg = s.bg.to(dt)
    rgb = img[:, :, :3] + Tmap[:, :, None] * bg[None, None]
    out = torch.cat([rgb, img[:, :, 3:]], -1).permute(2, 0, 1).contiguous()
    stats = dict(R=R, V=int(pre["vis"].sum()), final_T=Tmap, fragile=frag)
    if timings is not None:
        timings["pre_bin_s"] = _t1 - _t0
        timings["tiles_s"] = _time.perf_counter() - _t1
        timings["tiles_done"] = len(range(0, gx * gy, tile_stride))
        timings["tiles_total"] = gx * gy
    if s.f_count == 0:
        return out, pre["radii"], stats
    if s.f_count in (1, 2):
        return count, score, out[:3], pre["radii"], stats
    return count, pre["radii"], stats
