"""Camera containers with the reference's matrix conventions (`scene/cameras.py:19-113`)."""
import numpy as np
import torch

from .graphics_utils import getIntrinsic, getProjectionMatrix, getWorld2View2


class Camera:
    """`scene/cameras.py:19-73`: znear .01 / zfar 100; world_view_transform = W2C^T;
    full_proj_transform = W2C^T P^T; camera_center = inv(view)[3,:3]; intr from FoV."""

    def __init__(self, uid, R, T, FoVx, FoVy, image=None, normal=None, mask=None, width=None,
                 height=None, trans=np.array([0.0, 0.0, 0.0]), scale=1.0, device="cuda"):
        self.uid = self.idx = uid
        self.R, self.T = np.asarray(R), np.asarray(T)
        self.FoVx, self.FoVy = FoVx, FoVy
        self.device = torch.device(device)
        if image is not None:
            self.original_image = image.clamp(0.0, 1.0).to(self.device)
            height, width = self.original_image.shape[1:]
        self.image_width, self.image_height = int(width), int(height)
        self.normal = normal.to(self.device) if normal is not None else None
        if mask is not None:
            self.mask = mask.to(self.device)
        self.zfar, self.znear = 100.0, 0.01
        self.world_view_transform = torch.tensor(getWorld2View2(R, T, trans, scale)).t().contiguous().to(self.device)
        self.projection_matrix = getProjectionMatrix(self.znear, self.zfar, FoVx, FoVy).t().contiguous().to(self.device)
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).contiguous()
        self.camera_center = torch.inverse(self.world_view_transform.cpu())[3, :3].contiguous().to(self.device)
        intr = getIntrinsic(FoVx, FoVy, self.image_height, self.image_width)
        self.intr_scalars = (float(intr[0, 0]), float(intr[1, 1]), float(intr[0, 2]), float(intr[1, 2]))
        self.intr = intr.to(self.device)
        # camera rotation world->camera as a device tensor, built once (the reference re-uploads
        # `R.T` on every render, `gaussian_renderer/__init__.py:100`)
        self.R_w2c = torch.tensor(self.R.T, dtype=torch.float32).contiguous().to(self.device)


class SampleCam:
    """Virtual visibility camera (`scene/cameras.py:90-113`)."""

    def __init__(self, w2c, width, height, FoVx, FoVy, device="cuda"):
        self.FoVx, self.FoVy = FoVx, FoVy
        self.image_width, self.image_height = int(width), int(height)
        self.zfar, self.znear = 100.0, 0.01
        self.device = torch.device(device)
        w2c = w2c.to(torch.float32)
        self.R = w2c[:3, :3].t().cpu().numpy()
        self.world_view_transform = w2c.t().contiguous().to(self.device)
        self.projection_matrix = getProjectionMatrix(self.znear, self.zfar, FoVx, FoVy).t().contiguous().to(self.device)
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).contiguous()
        self.camera_center = torch.inverse(self.world_view_transform.cpu())[3, :3].contiguous().to(self.device)
        self.R_w2c = w2c[:3, :3].contiguous().to(self.device)
