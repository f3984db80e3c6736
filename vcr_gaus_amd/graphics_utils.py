"""Camera / projection conventions of the reference (`tools/graphics_utils.py`), device-agnostic.

Row-vector convention downstream: `p_view = [p,1] @ world_view_transform`
(`scene/cameras.py:68-70`).
"""
import math

import numpy as np
import torch


def fov2focal(fov, pixels):
    """`tools/graphics_utils.py:104-105`."""
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal, pixels):
    """`tools/graphics_utils.py:107-108`."""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World->camera 4x4 with an optional recentring of the camera position
    (`tools/graphics_utils.py:38-49`).  R is the camera-to-world rotation, t the w2c translation."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = np.asarray(t)
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """OpenGL-style perspective with z in [0,1] and w = +z (`tools/graphics_utils.py:63-86`)."""
    tx, ty = math.tan(fovX / 2.0), math.tan(fovY / 2.0)
    right, top = tx * znear, ty * znear
    P = torch.zeros(4, 4)
    P[0, 0] = znear / right
    P[1, 1] = znear / top
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def getIntrinsic(fovX, fovY, h, w):
    """K with the principal point forced to the image centre (`tools/graphics_utils.py:89-101`)."""
    K = torch.eye(3, dtype=torch.float32)
    K[0, 0] = fov2focal(fovX, w)
    K[1, 1] = fov2focal(fovY, h)
    K[0, 2] = w / 2
    K[1, 2] = h / 2
    return K


@torch.no_grad()
def get_all_px_dir(intrinsics, height, width):
    """Unit ray through every pixel centre, [3,H,W] (`tools/graphics_utils.py:143-155`):
    normalize(K^-1 [u+.5, v+.5, 1])."""
    dev = intrinsics.device
    u = torch.arange(width, dtype=torch.float32, device=dev) + 0.5
    v = torch.arange(height, dtype=torch.float32, device=dev) + 0.5
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    pix = torch.stack([uu, vv, torch.ones_like(uu)], -1)
    rays = pix @ torch.inverse(intrinsics.t())
    return torch.nn.functional.normalize(rays, dim=-1).permute(2, 0, 1).contiguous()
