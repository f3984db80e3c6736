"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Torch restatements of the reference's image-space loss chain, pinned by golden vectors captured
from the reference's own functions (tests/golden/g1_depth_normal.npz, g2_l1_ssim.npz):
  compute_normals      tools/normal_utils.py:24-41, tools/graphics_utils.py:111-131
  cos_weight           tools/loss_utils.py:135-143
  monosdf_normal_loss  tools/loss_utils.py:122-132
  l1_loss / ssim       tools/loss_utils.py:36,49-92
Used as the fp64 checker for the HIP kernels at sizes the fixtures do not cover.
"""
import math

import torch
import torch.nn.functional as F


def compute_normals(depth, K):
    """depth [1,H,W] or [H,W]; K [3,3].  X = K^-1 [(u+.5) z, (v+.5) z, z]; n = normalize(dX/du x dX/dv)."""
    d = depth.reshape(depth.shape[-2], depth.shape[-1])
    H, W = d.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=d.dtype) + 0.5, torch.arange(W, dtype=d.dtype) + 0.5, indexing="ij")
    fx, fy, cx, cy = K[0, 0].to(d.dtype), K[1, 1].to(d.dtype), K[0, 2].to(d.dtype), K[1, 2].to(d.dtype)
    X = torch.stack([(u - cx) * d / fx, (v - cy) * d / fy, d], -1)
    dv = torch.gradient(X, dim=0)[0]
    du = torch.gradient(X, dim=1)[0]
    return F.normalize(torch.cross(du, dv, dim=-1), p=2, dim=-1)


def cos_weight(render_normal, gt_normal, exp_t):
    c = (render_normal * gt_normal).sum(-1)
    return (torch.exp((c - 1) / exp_t) if exp_t > 0 else torch.ones_like(c)).detach()


def monosdf_normal_loss(pred, gt, weight=None):
    w = 1.0 if weight is None else weight
    return (w * (pred - gt).abs().sum(-1)).mean() + (w * (1.0 - (pred * gt).sum(-1))).mean()


def masked_weighted_normal_loss(pred, gt, wsrc=None, exp_t=0.0, mask=None):
    """What vcr_normal_loss_forward computes: the trainer.py:266-280 chain."""
    w = cos_weight(wsrc, gt, exp_t) if wsrc is not None else None
    if mask is not None:
        pred, gt = pred[mask], gt[mask]
        w = None if w is None else w[mask]
    return monosdf_normal_loss(pred, gt, w)


def l1_loss(a, b):
    return (a - b).abs().mean()


def ssim(img1, img2):
    C = img1.shape[-3]
    g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=img1.dtype)
    g = g / g.sum()
    win = (g[:, None] @ g[None, :]).expand(C, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t[None] if t.dim() == 3 else t, win, padding=5, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s11 = conv(img1 * img1) - mu1 * mu1
    s22 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean()


def entropy_loss(opacity):
    """`tools/loss_utils.py:30-33` (pinned by g5 / g7)."""
    return (-opacity * torch.log(opacity + 1e-6) - (1 - opacity) * torch.log(1 - opacity + 1e-6)).mean()
