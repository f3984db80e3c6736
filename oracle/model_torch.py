"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Torch restatements of the per-Gaussian glue around the rasterizer call:
  * `GaussianModel.get_normal` (`scene/gaussian_model.py:168-192`, `tools/general_utils.py:98-119`)
  * normal orientation + rotation to camera space (`gaussian_renderer/__init__.py:95-101`)
  * activations (`scene/gaussian_model.py:125-162`)
Device-agnostic and autograd-differentiable, so they serve as the checker for the fused HIP versions.
"""
import torch


def build_rotation(r):
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def get_normal(rotation_activated, scaling_activated):
    """Column of R(q) belonging to the smallest scale (`scene/gaussian_model.py:182-186`)."""
    rots = build_rotation(rotation_activated)
    axis = torch.argmin(scaling_activated, dim=-1)
    return rots.gather(2, axis[:, None, None].expand(-1, 3, -1)).squeeze(-1)


def camera_normals(normal_world, means3D, camera_center, R_w2c):
    """Flip to face away from the camera, rotate to camera space
    (`gaussian_renderer/__init__.py:97-101`; R_w2c = cam.R.T)."""
    view_dir = means3D - camera_center
    sign = (((view_dir * normal_world).sum(-1) > 0) * 1 - 0.5) * 2
    n = normal_world * sign[..., None]
    return n @ R_w2c.to(n.dtype).t()


def activations(raw):
    """raw dict (synthetic.make_gaussians layout) -> activated inputs of the rasterizer."""
    return dict(
        xyz=raw["xyz"],
        opacity=torch.sigmoid(raw["opacity"]),
        scaling=torch.exp(raw["scaling"]),
        rotation=torch.nn.functional.normalize(raw["rotation"]),
        shs=torch.cat([raw["f_dc"], raw["f_rest"]], 1),
    )
