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

    @classmethod
    def batch(cls, w2cs, width, height, FoVx, FoVy, device="cuda"):
        """The cameras of a whole [B,4,4] stack of world-to-camera matrices: the same fields as B constructor calls, formed
        by batched host arithmetic and FOUR host-to-device copies for all of them (the constructor costs four copies, a
        device matmul and a device-to-host round trip per camera: 0.2 ms each, 40 ms for the 200 cameras of one visibility
        pass -- as long as the renders themselves).  Every camera's tensors are views into the stacked ones, which
        `gaussian_renderer.visibility_counts` recognises (`_stack`) and hands to the library without re-stacking."""
        return cls.batch_from_host(cls.batch_host(w2cs, width, height, FoVx, FoVy), device)

    @staticmethod
    def batch_host(w2cs, width, height, FoVx, FoVy):
        """The host arithmetic of `batch` (no device work): -> a bundle `batch_from_host` turns into cameras.  Split off so that a
        trainer can form the 200 cameras of the NEXT densification on a worker thread while the GPU runs training steps (round 6:
        8 ms of the densification event were this arithmetic)."""
        w2cs = w2cs.to(torch.float32).cpu()
        B = w2cs.shape[0]
        wv = w2cs.transpose(1, 2).contiguous()
        proj = getProjectionMatrix(0.01, 100.0, FoVx, FoVy).t().contiguous()
        full = torch.matmul(wv, proj).contiguous()
        # (matrix by matrix: the batched CPU inverse takes 60 ms for 200 matrices here, 200 single ones 5 ms, and this is the
        #  constructor's arithmetic to the bit)
        centre = torch.stack([torch.inverse(wv[i])[3, :3] for i in range(B)]) if B else wv.new_zeros(0, 3)
        rot = w2cs[:, :3, :3].contiguous()
        return {"stack": (wv, full, centre, rot), "proj": proj, "R": rot.transpose(1, 2).numpy(), "size": (int(width), int(height)),
                "fov": (FoVx, FoVy)}

    @classmethod
    def batch_from_host(cls, host, device="cuda"):
        dev = torch.device(device)
        stack = tuple(t.to(dev) for t in host["stack"])
        proj_d = host["proj"].to(dev)
        R_np = host["R"]
        (width, height), (FoVx, FoVy) = host["size"], host["fov"]
        cams = []
        for i in range(stack[0].shape[0]):
            c = cls.__new__(cls)
            c.FoVx, c.FoVy = FoVx, FoVy
            c.image_width, c.image_height = int(width), int(height)
            c.zfar, c.znear = 100.0, 0.01
            c.device = dev
            c.R = R_np[i]
            c.world_view_transform, c.full_proj_transform, c.camera_center, c.R_w2c = (t[i] for t in stack)
            c.projection_matrix = proj_d
            c._stack, c._stack_index = stack, i
            cams.append(c)
        return cams
