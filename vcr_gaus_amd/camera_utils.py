"""Virtual visibility cameras for `densify_large` (`tools/camera_utils.py:315-401,404-481`, used by
`Trainer.sample_cameras` / `get_visi_mask_acc`, `trainer.py:357-370,621-634,688-702`): camera centres on or inside the
normalised bounding box `pts_norm = (pts - trans) / scale` (or `trans` a 4x4 world -> box transform), every one looking
at a target, rendered as 1500 x 1500 / FoV 2.5 rad `SampleCam`s.

`bb_camera` covers every placement the reference offers:
  sample_mode 'random'  centres uniform in the box (what the training loop requests),
              'grid'    a regular lattice on the top face (`up`) and / or on the four side walls (`around`), the number
                        of points per face proportional to its area; `bidirect` builds the walls twice, offset by half a
                        cell, the second half looking at the mirrored target;
  look_mode   'target'  all cameras look at one point (default: one unit below the box centre),
              'direction' every camera looks at its own mirror point across the box.
The arithmetic is pinned against the reference's function by the fixture tests/golden/g11_bb_camera.npz.
"""
import math

import torch

from .cameras import SampleCam

# world axes of a COLMAP scene (`tools/camera_utils.py:124-142`): up = -y, front = +z, right = +x
_WORLD_AXES = {"up": (0.0, -1.0, 0.0), "front": (0.0, 0.0, 1.0), "right": (1.0, 0.0, 0.0)}


def _normalize(v):
    return v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=1e-20))


def look_at_w2c(campos, target, opengl=False):
    """Rows = camera right / up / forward axes in world coordinates (`tools/camera_utils.py:182-199`)."""
    up = torch.tensor([0.0, 1.0, 0.0], device=campos.device).expand_as(campos)
    if not opengl:
        fwd = _normalize(target - campos)
        right = _normalize(torch.cross(fwd, up, dim=-1))
        up2 = _normalize(torch.cross(right, fwd, dim=-1))
    else:
        fwd = _normalize(campos - target)
        right = _normalize(torch.cross(up, fwd, dim=-1))
        up2 = _normalize(torch.cross(fwd, right, dim=-1))
    return torch.stack([right, up2, fwd], dim=1)


def _box_axis(rot, name):
    """Which box axis (index, sign) the world's `name` axis maps to under the world -> box rotation `rot`."""
    c = rot @ torch.tensor(_WORLD_AXES[name], dtype=torch.float32, device=rot.device)
    k = int(torch.argmax(c.abs()))
    return k, float(torch.sign(c[k]))


def _from_box(pts, trans, scale):
    """`inv_normalize_pts` (`tools/math_utils.py:61-67`)."""
    if trans.ndim == 1:
        return pts * scale + trans
    return (pts * scale[None] - trans[:3, 3:].T) @ trans[:3, :3]


def _top_lattice(n, scale, ra, ua, fa):
    """Regular lattice on the box's top face (`up_grid_posi`, `tools/camera_utils.py:465-481`): counts per side
    proportional to the side lengths, end points included."""
    l, w = scale[ra], scale[fa]
    ratio = math.sqrt(n / (l * w))               # (fp32 area, fp64 root: the rounding below is sensitive to it)
    nl, nw = int(torch.round(l * ratio)), int(torch.round(w * ratio))
    gl, gw = torch.meshgrid([torch.linspace(-1, 1, nl), torch.linspace(-1, 1, nw)], indexing="xy")
    pts = torch.ones(gl.numel(), 3)
    pts[:, ra], pts[:, fa] = gl.flatten(), gw.flatten()
    return pts


def _wall_lattice(n, scale, ra, ua, fa, sign=1, up_sign=1.0):
    """Regular lattice on the four side walls (`around_grid_posi`, `tools/camera_utils.py:404-462`), walked front
    (+front axis), back, right (+right axis), left; `sign = -1` shifts every row by one cell (second half of `bidirect`).
    Half-open ranges: every wall contributes its start corner, not its end corner."""
    dev = scale.device
    h, l, w = scale[ua], scale[ra], scale[fa]
    ratio = (n / (2 * (l * h + h * w))).sqrt()
    nh, nl, nw = (torch.round(v * ratio).int() for v in (h, l, w))
    hc = torch.arange(start=-1, end=1, step=2 / nh, device=dev) * up_sign
    out = []

    def wall(fixed_axis, fixed_val, run_axis, start, end, step, run_first):
        first = start if sign == 1 else start + step
        rc = torch.arange(start=first, end=end, step=step, device=dev)
        if run_first:      # the running coordinate varies fastest
            g_run, g_h = torch.meshgrid([rc, hc], indexing="xy")
        else:              # the height varies fastest
            g_h, g_run = torch.meshgrid([hc, rc], indexing="xy")
        p = torch.full((g_run.numel(), 3), float(fixed_val), dtype=torch.float32, device=dev)
        p[:, run_axis], p[:, ua] = g_run.flatten(), g_h.flatten()
        out.append(p)

    wall(fa, 1, ra, -1, 1, 2 / nl, True)
    wall(fa, -1, ra, 1, -1, -2 / nl, True)
    wall(ra, 1, fa, 1, -1, -2 / nw, False)
    wall(ra, -1, fa, -1, 1, 2 / nw, False)
    return torch.cat(out, 0)


def bb_camera(n, trans, scale, height=None, target=None, opengl=False, up=True, around=True, look_mode="target",
              sample_mode="grid", boundary=0.9, bidirect=False, generator=None):
    """World-to-camera matrices [m,4,4] (`tools/camera_utils.py:315-401`).  `generator`: optional torch.Generator for
    the 'random' placement (default: the global RNG, like the reference)."""
    trans = torch.as_tensor(trans, dtype=torch.float32)
    scale = torch.as_tensor(scale, dtype=torch.float32)
    dev = trans.device
    if scale.ndim == 0:
        scale = torch.ones(3, device=dev) * scale
    rot = trans[:3, :3] if trans.ndim == 2 else torch.eye(3, device=dev)
    ua, us = _box_axis(rot, "up")
    grid = sample_mode == "grid"
    if grid or (up and around):
        ra, _ = _box_axis(rot, "right")
        fa, _ = _box_axis(rot, "front")
    side_axes = [i for i in (0, 1, 2) if i != ua]
    n_up = n_around = n
    if up and around:            # share of the top face in the surface that carries cameras
        h, l, w = scale[ua], scale[ra], scale[fa]
        n_up = int(n * (l * w) / (2 * (l * h + h * w) + l * w))
    per_camera_targets = look_mode == "direction"
    tgt_list = []
    if target is None:
        if not per_camera_targets:
            target = torch.zeros(1, 3, device=dev)
            target[:, ua] = -us
    else:
        target = torch.as_tensor(target, dtype=torch.float32)
    rnd = lambda k: (torch.rand(k, 3, generator=generator).to(dev) * 2 - 1)
    pos = []
    top = None
    if up:
        if grid:
            top = _top_lattice(n_up, scale, ra, ua, fa).to(dev)
            n_around = n - n_up
        else:
            top = rnd(n_up)
        top[:, ua] = us
        pos.append(top)
        if per_camera_targets:
            t = top.clone()
            t[:, ua] *= -1
            tgt_list.append(t)
    if around:
        if not grid:
            walls = rnd(n_around)
        elif not bidirect:
            walls = _wall_lattice(n_around, scale, ra, ua, fa, up_sign=us).to(dev)
        else:
            first = _wall_lattice(n_around // 2, scale, ra, ua, fa, sign=1, up_sign=us).to(dev)
            second = _wall_lattice(n_around - first.shape[0], scale, ra, ua, fa, sign=-1, up_sign=us).to(dev)
            walls = torch.cat([first, second], 0)
            total = walls.shape[0] + (top.shape[0] if up else 0)
            target = target.repeat(total, 1)
            target[-second.shape[0]:, ua] *= -1           # the second half looks at the mirrored target
        walls[:, ua] = walls[:, ua] * boundary + (1 - boundary) * us
        pos.append(walls)
        if per_camera_targets:
            t = walls.clone()
            for i in side_axes:                            # (the reference's strided mirror pattern, `:384-386`)
                t[i - 1::2, i] *= -1
            tgt_list.append(t)
    xyz = torch.cat(pos, 0)
    if per_camera_targets:
        target = torch.cat(tgt_list, 0)
    xyz = _from_box(xyz, trans, scale)
    target = _from_box(target, trans, scale)
    R = look_at_w2c(xyz, target.expand_as(xyz) if target.shape[0] == 1 else target, opengl)
    T = torch.zeros(xyz.shape[0], 4, 4, device=dev)
    T[:, :3, :3] = R
    T[:, :3, 3] = -(R @ xyz[..., None]).squeeze(-1)
    T[:, 3, 3] = 1
    return T


def bb_camera_random(n, trans, scale, up=False, around=True, boundary=0.9, generator=None):
    """The form the training loop requests (`trainer.py:363-366`): `sample_mode='random'`, one shared target."""
    return bb_camera(n, trans.detach().float().cpu(), scale.detach().float().cpu(), up=up, around=around,
                     sample_mode="random", boundary=boundary, generator=generator)


def sample_cameras(n, trans, scale, up=False, around=True, look_mode="target", sample_mode="random", bidirect=True,
                   device="cuda", generator=None, size=1500, fov=2.5):
    """`Trainer.sample_cameras` (`trainer.py:621-634`)."""
    return SampleCam.batch_from_host(sample_cameras_host(n, trans, scale, up=up, around=around, look_mode=look_mode,
                                                         sample_mode=sample_mode, bidirect=bidirect, generator=generator, size=size,
                                                         fov=fov), device)


def sample_cameras_host(n, trans, scale, up=False, around=True, look_mode="target", sample_mode="random", bidirect=True,
                        generator=None, size=1500, fov=2.5):
    """The host half of `sample_cameras` (placement + camera matrices, no device work): what a trainer runs ahead of a
    densification on a worker thread; `SampleCam.batch_from_host` finishes it."""
    w2cs = bb_camera(n, trans.detach().float().cpu(), scale.detach().float().cpu(), up=up, around=around,
                     look_mode=look_mode, sample_mode=sample_mode, bidirect=bidirect, generator=generator)
    return SampleCam.batch_host(w2cs, size, size, fov, fov)
