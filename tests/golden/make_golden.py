"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own Python
functions from /root/reference (possible only in the build container; the reference cannot travel).
Fixtures hold inputs and the reference's outputs only.  Run:  python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

# tools.general_utils imports torchvision / PIL at module level; they are not needed by the functions used
for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "PIL"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
sys.modules["PIL"].ImageFile = types.SimpleNamespace(LOAD_TRUNCATED_IMAGES=False)

from tools import graphics_utils as RG  # noqa: E402
from tools import loss_utils as RL  # noqa: E402
from tools import normal_utils as RN  # noqa: E402
from tools import sh_utils as RS  # noqa: E402
from tools import math_utils as RM  # noqa: E402
from tools import image_utils as RI  # noqa: E402
from tools import general_utils as RGU  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name)


def depth_cases():
    g = torch.Generator().manual_seed(0)
    out = {}
    for tag, (H, W) in {"a": (48, 64), "b": (37, 53)}.items():
        K = RG.getIntrinsic(1.1, 0.9, H, W)
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        plane = 2.0 + 0.01 * xx + 0.02 * yy
        sphere = 3.0 - torch.sqrt(torch.clamp(1.0 - ((xx - W / 2) / W) ** 2 - ((yy - H / 2) / H) ** 2, min=0.05))
        rnd = 1.5 + torch.rand(H, W, generator=g)
        for dn, d in {"plane": plane, "sphere": sphere, "rand": rnd}.items():
            d = d[None].clone().requires_grad_(True)
            n = RN.compute_normals(d, K)
            gt = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
            rn = torch.nn.functional.normalize(gt + 0.3 * torch.randn(H, W, 3, generator=g), dim=-1)
            mask = torch.rand(H, W, generator=g) > 0.3
            w = RL.cos_weight(rn, gt, 0.01)
            loss = RL.monosdf_normal_loss(n[mask], gt[mask], w[mask])
            loss.backward()
            plain = RL.monosdf_normal_loss(n.detach(), gt)
            out.update({f"{tag}_{dn}_depth": d.detach(), f"{tag}_{dn}_K": K, f"{tag}_{dn}_normal": n.detach(),
                        f"{tag}_{dn}_gt": gt, f"{tag}_{dn}_rn": rn, f"{tag}_{dn}_mask": mask, f"{tag}_{dn}_w": w,
                        f"{tag}_{dn}_loss": loss.detach(), f"{tag}_{dn}_plain": plain,
                        f"{tag}_{dn}_ddepth": d.grad})
    save("g1_depth_normal.npz", **out)


def image_cases():
    g = torch.Generator().manual_seed(1)
    out = {}
    for tag, (H, W) in {"a": (40, 56), "b": (33, 70)}.items():
        a = torch.rand(3, H, W, generator=g).requires_grad_(True)
        b = (a.detach() + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
        l1 = RL.l1_loss(a, b)
        s = RL.ssim(a, b)
        (0.8 * l1 + 0.2 * (1 - s)).backward()
        out.update({f"{tag}_a": a.detach(), f"{tag}_b": b, f"{tag}_l1": l1.detach(), f"{tag}_ssim": s.detach(),
                    f"{tag}_grad": a.grad, f"{tag}_psnr": RI.psnr(a.detach(), b)})
    save("g2_l1_ssim.npz", **out)


def sh_cases():
    g = torch.Generator().manual_seed(2)
    N = 257
    sh = torch.randn(N, 3, 16, generator=g)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    out = {"sh": sh, "dirs": d}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = torch.clamp_min(RS.eval_sh(deg, sh, d) + 0.5, 0.0)   # gaussian_renderer/__init__.py:86-87
    out["rgb2sh"] = RS.RGB2SH(torch.linspace(0, 1, 11))
    save("g3_sh.npz", **out)


def camera_cases():
    out = {}
    g = np.random.RandomState(3)
    for i in range(3):
        A = g.randn(3, 3)
        R, _ = np.linalg.qr(A)
        if np.linalg.det(R) < 0:
            R[:, 0] *= -1
        T = g.randn(3)
        fovx, fovy = 0.6 + 0.3 * i, 0.5 + 0.2 * i
        H, W = 60 + 7 * i, 80 + 5 * i
        w2c = RG.getWorld2View2(R, T, np.array([0.1 * i, 0.0, -0.05]), 1.0 + 0.5 * i)
        view = torch.tensor(w2c).transpose(0, 1)
        proj = RG.getProjectionMatrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
        full = view.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        K = RG.getIntrinsic(fovx, fovy, H, W)
        # get_all_px_dir hard-codes .cuda(); same arithmetic on the CPU (tools/graphics_utils.py:143-155)
        _, ray = RG.depth2point_cam(torch.ones(1, 1, 1, H, W), K[None])
        dirs = torch.nn.functional.normalize(ray.squeeze(), dim=-1).permute(2, 0, 1)
        out.update({f"c{i}_R": R, f"c{i}_T": T, f"c{i}_trans": np.array([0.1 * i, 0.0, -0.05]), f"c{i}_scale": 1.0 + 0.5 * i,
                    f"c{i}_fov": np.array([fovx, fovy]), f"c{i}_hw": np.array([H, W]), f"c{i}_view": view, f"c{i}_proj": proj,
                    f"c{i}_full": full, f"c{i}_center": view.inverse()[3, :3], f"c{i}_K": K, f"c{i}_dirs": dirs,
                    f"c{i}_focal": np.array([RG.fov2focal(fovx, W), RG.focal2fov(RG.fov2focal(fovx, W), W)])})
    save("g4_cameras.npz", **out)


def misc_cases():
    g = torch.Generator().manual_seed(4)
    img = torch.rand(3, 30, 41, generator=g)
    dist = torch.rand(1, 30, 41, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(30, 41, 3, generator=g), dim=-1)
    mask = (torch.rand(30, 41, 1, generator=g) > 0.2).float()
    op = torch.rand(100, 1, generator=g)
    pts = torch.randn(200, 3, generator=g)
    trans, scale = torch.tensor([0.1, -0.2, 0.3]), torch.tensor([1.5, 1.0, 0.8])
    inside, npts = RM.get_inside_normalized(pts, trans, scale)
    lr = RGU.get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    save("g5_misc.npz", img=img, dist=dist, edge=RN.get_edge_aware_distortion_map(img, dist), nrm=nrm, mask=mask,
         curv=RL.normal2curv(nrm, mask), op=op, entropy=RL.entropy_loss(op), pts=pts, trans=trans, scale=scale,
         inside=inside, npts=npts, lr_steps=np.array([0, 1, 100, 15000, 30000]),
         lr_vals=np.array([lr(s) for s in [0, 1, 100, 15000, 30000]]))


if __name__ == "__main__":
    depth_cases(); image_cases(); sh_cases(); camera_cases(); misc_cases()
