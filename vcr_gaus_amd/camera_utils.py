"""Virtual visibility cameras for `densify_large` (`tools/camera_utils.py:315-401` in its `sample_mode='random'`
form, which is what `Trainer.get_visi_mask_acc` requests, `trainer.py:363-366,621-634`): camera centres are drawn
uniformly inside the normalised bounding box (squeezed towards the up side by `boundary`), all looking at the
point one unit below the box centre, rendered as 1500x1500 / FoV 2.5 rad `SampleCam`s."""
import torch

from .cameras import SampleCam


def _normalize(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-20)


def look_at_w2c(campos, target):
    """Rows = camera right / up / forward axes in world coordinates, forward = target - campos
    (`tools/camera_utils.py:182-199`, opengl=False branch)."""
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(campos)
    fwd = _normalize(target - campos)
    right = _normalize(torch.cross(fwd, up, dim=-1))
    up2 = _normalize(torch.cross(right, fwd, dim=-1))
    return torch.stack([right, up2, fwd], dim=1)


def bb_camera_random(n, trans, scale, up=False, around=True, boundary=0.9, generator=None):
    """World-to-camera matrices [m,4,4] for the box `pts_norm = (pts - trans) / scale` (vector `trans`)."""
    trans, scale = trans.detach().float().cpu(), scale.detach().float().cpu()
    up_axis, up_sign = 1, -1.0                  # COLMAP world: up = (0,-1,0) (`tools/camera_utils.py:124-142`)
    xyz = []
    if up:
        p = torch.rand(n, 3, generator=generator) * 2 - 1
        p[:, up_axis] = up_sign
        xyz.append(p)
    if around:
        p = torch.rand(n, 3, generator=generator) * 2 - 1
        p[:, up_axis] = p[:, up_axis] * boundary + (1 - boundary) * up_sign
        xyz.append(p)
    xyz = torch.cat(xyz, 0) * scale + trans      # inv_normalize_pts (`tools/math_utils.py:61-67`)
    target = torch.zeros(1, 3)
    target[:, up_axis] = -up_sign
    target = (target * scale + trans).expand_as(xyz)
    R = look_at_w2c(xyz, target)
    T = torch.zeros(xyz.shape[0], 4, 4)
    T[:, :3, :3] = R
    T[:, :3, 3] = -(R @ xyz[..., None]).squeeze(-1)
    T[:, 3, 3] = 1
    return T


def sample_cameras(n, trans, scale, up=False, around=True, device="cuda", generator=None, size=1500, fov=2.5):
    """`Trainer.sample_cameras` (`trainer.py:621-634`)."""
    w2cs = bb_camera_random(n, trans, scale, up=up, around=around, generator=generator)
    return [SampleCam(w2cs[i], size, size, fov, fov, device=device) for i in range(w2cs.shape[0])]
