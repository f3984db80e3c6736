"""Deterministic synthetic scenes + orbit cameras.

There is no dataset access in this environment, so every workload BASELINE.json names is
rebuilt as a seeded synthetic scene of the same shape: surface-biased Gaussian centres inside the
unit bounding box, flattened anisotropic scales, random rotations / opacities / SH, and COLMAP
convention cameras on orbit rings looking at the origin.  Generation happens on the CPU with an
explicit generator so that every rank of a data-parallel job builds bit-identical replicas.
"""
import math

import numpy as np
import torch

from .cameras import Camera
from .graphics_utils import focal2fov
from .sh_utils import RGB2SH

# name -> (num_gaussians, num_views, width, height, focal, sem_channels)
WORKLOADS = {
    "c1_10k_256": (10_000, 4, 256, 256, 221.7, 0),
    "c2_dtu_300k_800x600": (300_000, 49, 800, 600, 597.0, 0),
    "c4_tnt_2m_1080p": (2_000_000, 300, 1920, 1080, 1165.0, 2),
    "c5_360_5m_1600x1200": (5_000_000, 100, 1600, 1200, 1250.0, 0),
    "metric_1m_1080p": (1_000_000, 8, 1920, 1080, 1165.0, 0),
}
# Denser variants of a workload (same scene, every scale multiplied): the Appendix-B recipe gives R/N ~ 3 tile instances per
# Gaussian and leaves 84 % of the 1080p tiles empty; real trained scenes sit at R/N ~ 10-20.  name -> (base, scale multiplier)
DENSE_VARIANTS = {"dense_1m_1080p": ("metric_1m_1080p", 3.5)}
# Full-frame variant: the same 1 M Gaussians seen from INSIDE the orbit (radius 1.3 instead of 3: the scene subtends more than
# the field of view) with every scale x 1.6, so that nearly every pixel is covered and R/N ~ 10 as in a trained scene -- the
# Appendix-B orbit leaves 81 % of the 1080p frame empty.  name -> (base workload, camera orbit radius, scale multiplier)
CLOSE_VARIANTS = {"fullframe_1m_1080p": ("metric_1m_1080p", 1.3, 1.6)}


def workload(name):
    """-> (n, views, width, height, focal, sem_channels, scale_mult) for a name of WORKLOADS, DENSE_VARIANTS or CLOSE_VARIANTS."""
    if name in DENSE_VARIANTS:
        base, mult = DENSE_VARIANTS[name]
        return WORKLOADS[base] + (mult,)
    if name in CLOSE_VARIANTS:
        return WORKLOADS[CLOSE_VARIANTS[name][0]] + (CLOSE_VARIANTS[name][2],)
    return WORKLOADS[name] + (1.0,)


def camera_radius(name):
    """Orbit radius of the workload's cameras (3 = SURVEY Appendix B)."""
    return CLOSE_VARIANTS[name][1] if name in CLOSE_VARIANTS else 3.0


def _surface_points(n, g):
    """Points on a sphere (r=.55), a ground disc and a box shell, jittered off-surface."""
    k = torch.randint(0, 3, (n,), generator=g)
    u = torch.rand(n, 3, generator=g)
    # sphere
    v = torch.randn(n, 3, generator=g)
    sph = 0.55 * v / v.norm(dim=1, keepdim=True) + torch.tensor([0.0, -0.1, 0.0])
    # ground disc (y is down in COLMAP axes -> floor at y=+0.5)
    rad = 0.95 * torch.sqrt(u[:, 0])
    ang = 2 * math.pi * u[:, 1]
    disc = torch.stack([rad * torch.cos(ang), torch.full((n,), 0.5), rad * torch.sin(ang)], 1)
    # box shell
    face = torch.randint(0, 6, (n,), generator=g)
    box = (u * 2 - 1) * 0.3
    ax = face % 3
    sgn = (face // 3).float() * 2 - 1
    box[torch.arange(n), ax] = 0.3 * sgn
    box = box + torch.tensor([0.55, 0.2, 0.45])
    pts = torch.where((k == 0)[:, None], sph, torch.where((k == 1)[:, None], disc, box))
    return pts + 0.004 * torch.randn(n, 3, generator=g)


def make_gaussians(n, seed=0, sh_degree=3, sem_channels=0, device="cpu"):
    """Raw (pre-activation) parameters in the reference's storage layout
    (`scene/gaussian_model.py:219-229`): _xyz[N,3] _features_dc[N,1,3] _features_rest[N,K-1,3]
    _scaling[N,3] (log) _rotation[N,4] _opacity[N,1] (logit) (+ _objects_dc[N,1,S])."""
    g = torch.Generator().manual_seed(seed)
    xyz = _surface_points(n, g)
    area = 4 * math.pi * 0.55 ** 2 + math.pi * 0.95 ** 2 + 6 * 0.6 ** 2
    dbar = math.sqrt(area / n)
    lo, hi = math.log(0.3 * dbar), math.log(3.0 * dbar)
    log_s = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    flat = torch.randint(0, 3, (n,), generator=g)
    log_s[torch.arange(n), flat] += math.log(0.05)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    op = 1.5 * torch.randn(n, 1, generator=g)
    K = (sh_degree + 1) ** 2
    f_dc = RGB2SH(torch.rand(n, 1, 3, generator=g))
    f_rest = 0.05 * torch.randn(n, K - 1, 3, generator=g)
    out = dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, scaling=log_s, rotation=q, opacity=op)
    if sem_channels:
        out["obj_dc"] = RGB2SH(torch.rand(n, 1, sem_channels, generator=g))
    return {k: v.float().contiguous().to(device) for k, v in out.items()}


def look_at_colmap(eye, target=(0.0, 0.0, 0.0)):
    """Camera-to-world rotation R (columns = camera x right, y down, z forward in world) and the
    world-to-camera translation T, as `scene/cameras.py:28-29` expects them."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    up_world = np.array([0.0, -1.0, 0.0])           # COLMAP: +y is down
    right = np.cross(fwd, up_world)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 1)             # c2w
    T = -R.T @ eye
    return R, T


def orbit_eyes(n_views, radius=3.0, elevations_deg=(-15.0, 15.0, 40.0)):
    eyes = []
    for i in range(n_views):
        el = math.radians(elevations_deg[i % len(elevations_deg)])
        az = 2 * math.pi * (i / n_views) + 0.1
        eyes.append((radius * math.cos(el) * math.sin(az), -radius * math.sin(el),
                     radius * math.cos(el) * math.cos(az)))
    return eyes


def make_cameras(n_views, width, height, focal, radius=3.0, device="cpu"):
    fovx, fovy = focal2fov(focal, width), focal2fov(focal, height)
    cams = []
    for i, eye in enumerate(orbit_eyes(n_views, radius)):
        R, T = look_at_colmap(eye)
        cams.append(Camera(i, R, T, fovx, fovy, width=width, height=height, device=device))
    return cams


def cameras_extent(cams):
    """1.1 x max camera distance from the mean camera centre (`scene/dataset_readers.py:57-78`)."""
    c = torch.stack([cam.camera_center.cpu() for cam in cams])
    return float(1.1 * (c - c.mean(0, keepdim=True)).norm(dim=1).max())


def make_workload(name, seed=0, device="cpu", max_views=None):
    n, views, w, h, f, sem = WORKLOADS[name]
    if max_views is not None:
        views = min(views, max_views)
    params = make_gaussians(n, seed=seed, sem_channels=sem, device=device)
    cams = make_cameras(views, w, h, f, device=device)
    return params, cams
