"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

One training iteration of the reference restated on the CPU in torch (fp64 by default), assembled from the other oracle
modules, to check `vcr_gaus_amd.trainer.Trainer.train_step` as a whole (BASELINE config 3: "grad parity"):

  render            gaussian_renderer/__init__.py:22-164   (activations, shortest-axis normal flipped to the camera and
                                                           rotated, rasterizer, mask, F.normalize, compute_normals)
  _compute_loss     trainer.py:233-308                     (l1, 1-ssim, l1_scale, mono_normal, depth_normal with the
                                                           cos_weight confidence and the render mask, curv,
                                                           consistent_normal, distortion / depth_var, entropy)
  _get_total_loss   trainer.py:310-321                     (weighted sum over the configured weights)
  optimizer         scene/gaussian_model.py:232-270        (torch.optim.Adam(lr=0, eps=1e-15), per-group learning rates,
                                                           xyz lr from get_expon_lr_func, tools/general_utils.py:49-82)

Parity status: the rasterizer part is PARITY UNPINNED (see raster_torch.py); the loss functions are pinned by the
reference-generated fixtures g1 / g2 / g5 / g7 (tests/test_oracle_cpu.py).  Only tests may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import losses_torch as OL
from . import model_torch as OM
from . import raster_torch as OR

GROUPS = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """`tools/general_utils.py:49-82`."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return float(delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


def edge_aware_map(gt_image, dmap):
    """`tools/normal_utils.py:57-66`."""
    c = gt_image[:, 1:-1, 1:-1]
    g = torch.stack([(c - gt_image[:, 1:-1, :-2]).abs().mean(0), (c - gt_image[:, 1:-1, 2:]).abs().mean(0),
                     (c - gt_image[:, :-2, 1:-1]).abs().mean(0), (c - gt_image[:, 2:, 1:-1]).abs().mean(0)], -1)
    return dmap * F.pad(torch.exp(-g.max(-1)[0]), (1, 1, 1, 1))


def normal2curv(normal, mask):
    """`tools/loss_utils.py:287-300`."""
    n = F.pad(normal[None], [0, 0, 1, 1, 1, 1], mode="replicate")
    m = F.pad(mask[None].to(normal.dtype), [0, 0, 1, 1, 1, 1], mode="replicate").to(torch.bool)
    c = n[:, 1:-1, 1:-1] * m[:, 1:-1, 1:-1]
    tot = ((n[:, :-2, 1:-1] - c) * m[:, :-2, 1:-1] + (n[:, 1:-1, :-2] - c) * m[:, 1:-1, :-2]
           + (n[:, 2:, 1:-1] - c) * m[:, 2:, 1:-1] + (n[:, 1:-1, 2:] - c) * m[:, 1:-1, 2:])
    return (tot[0] * mask).norm(1, -1, True)


def render(raw, cam, cfg, extent, bg, dirs, sh_degree, num_dist=0):
    """`gaussian_renderer/__init__.py:22-164` on raw (pre-activation) parameters; camera tensors on the CPU."""
    dt = raw["xyz"].dtype
    act = OM.activations(raw)
    nw = OM.get_normal(act["rotation"], act["scaling"])
    ncam = OM.camera_normals(nw, act["xyz"], cam.camera_center.cpu().to(dt), cam.R_w2c.cpu().to(dt))
    N = raw["xyz"].shape[0]
    s = OR.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg.cpu().to(dt),
                    1.0, cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), sh_degree, cam.camera_center.cpu())
    m2 = torch.zeros(N, 3, dtype=dt, requires_grad=True)
    m2d = torch.zeros(N, 3, dtype=dt, requires_grad=True)
    sem = raw["obj_dc"].squeeze(1) if "obj_dc" in raw and cfg.optim.loss_weight.semantic > 0 else None
    out, radii, st = OR.rasterize(s, act["xyz"], m2, m2d, act["shs"], None, ncam, sem, act["opacity"], act["scaling"],
                                  act["rotation"], None, None if dirs is None else dirs.cpu(), num_dist=num_dist,
                                  fragile=dt == torch.float64)      # (the fp64 reference also reports its fragile decisions)
    image, depth, normal, alpha = out[:8].split([3, 1, 3, 1], dim=0)
    with torch.no_grad():
        mask = cam.mask.cpu().bool() if hasattr(cam, "mask") else torch.ones_like(depth, dtype=torch.bool).squeeze(0)
        if cfg.optim.mask_depth_thr > 0:
            mask = mask & (depth < extent * cfg.optim.mask_depth_thr).squeeze(0)
    normal = F.normalize(normal.permute(1, 2, 0), dim=-1)
    est = OL.compute_normals(depth, cam.intr.cpu())
    return dict(render=image, depth=depth, normal=normal, est_normal=est, alpha=alpha, mask=mask, radii=radii, out=out,
                viewspace_points=m2, viewspace_points_densify=m2d, stats=st)


def losses(data, raw, cam, cfg, it, trans, scale, classifier=None):
    """`trainer.py:233-321` -> (dict of losses, weighted total).  `classifier`: (weight [K,S,1,1], bias [K]) of the 1x1
    convolution over the rendered semantic planes (`gaussian_renderer/__init__.py:149-152`, `trainer.py:304-307`)."""
    w = {k: v for k, v in cfg.optim.loss_weight.items() if v}
    dt = raw["xyz"].dtype
    gt_image = cam.original_image.cpu().to(dt)
    L = {"l1": OL.l1_loss(data["render"], gt_image), "ssim": 1.0 - OL.ssim(data["render"], gt_image)}
    pts = (raw["xyz"] - trans.cpu().to(dt)) / scale.cpu().to(dt)
    inside = torch.all(pts.detach().abs() < 1, dim=-1)
    if "l1_scale" in w:
        sc = torch.exp(raw["scaling"])[inside].min(-1)[0]
        L["l1_scale"] = sc.abs().mean()
    if "entropy" in w:
        L["entropy"] = OL.entropy_loss(torch.sigmoid(raw["opacity"])[inside])
    gt_normal = cam.normal.cpu().to(dt) if getattr(cam, "normal", None) is not None else None
    if "mono_normal" in w and it > cfg.optim.normal_from_iter:
        L["mono_normal"] = OL.monosdf_normal_loss(data["normal"], gt_normal)
    if "depth_normal" in w and it > cfg.optim.dnormal_from_iter:
        m = data["mask"]
        wgt = OL.cos_weight(data["normal"].detach(), gt_normal, cfg.optim.exp_t)
        if m.sum() != 0:
            L["depth_normal"] = OL.monosdf_normal_loss(data["est_normal"][m], gt_normal[m], wgt[m])
        else:
            L["depth_normal"] = torch.zeros((), dtype=dt)
        if "curv" in w and it > getattr(cfg.optim, "curv_from_iter", 0):
            L["curv"] = normal2curv(data["est_normal"], data["mask"][..., None].to(dt)).abs().mean()
    if "consistent_normal" in w and it > cfg.optim.consistent_normal_from_iter:
        L["consistent_normal"] = OL.monosdf_normal_loss(data["est_normal"], data["normal"])
    if "distortion" in w and it > cfg.optim.close_depth_from_iter and "distortion" in data:
        L["distortion"] = edge_aware_map(gt_image, data["distortion"]).mean()
    if "depth_var" in w and it > cfg.optim.close_depth_from_iter and "depth_var" in data:
        L["depth_var"] = edge_aware_map(gt_image, data["depth_var"]).mean()
    if "semantic" in w and classifier is not None and data["out"].shape[0] > 8:
        cw, cb = classifier
        S = cw.shape[1]
        logits = F.conv2d(data["out"][8:8 + S][None], cw.to(dt), cb.to(dt))[0].permute(1, 2, 0)          # [H, W, cls]
        L["semantic"] = F.cross_entropy(logits.reshape(-1, cw.shape[0]), cam.mask.cpu().reshape(-1).long()) / math.log(cw.shape[0])
    total = sum(L[k] * w[k] for k in w if k in L)
    return L, total


def step(raw, cam, cfg, extent, bg, dirs, it, sh_degree, trans, scale, spatial_lr_scale, adam_state=None, dtype=torch.float64):
    """One iteration on a copy of `raw` (dict of CPU tensors, reference storage layout).  Returns dict(losses, total,
    grads, params (after Adam), radii, densify_grad).  `adam_state`: {name: (step, exp_avg, exp_avg_sq)} or None (fresh)."""
    leaf = {k: v.detach().cpu().to(dtype).clone().requires_grad_(True) for k, v in raw.items() if k in GROUPS + ["obj_dc"]}
    lw = cfg.optim.loss_weight
    nd = 2 if getattr(lw, "depth_var", 0) > 0 else (1 if getattr(lw, "distortion", 0) > 0 else 0)
    data = render(leaf, cam, cfg, extent, bg, dirs, sh_degree, num_dist=nd)
    if nd == 2:
        d1, d2 = data["out"][-2:-1], data["out"][-1:]
        data["depth_var"] = d2 / data["alpha"] - (d1 / data["alpha"]) ** 2
    if nd == 1:
        data["distortion"] = data["out"][-1:]
    L, total = losses(data, leaf, cam, cfg, it, trans, scale)
    total.backward()
    o = cfg.optim
    lrs = {"xyz": expon_lr(it, o.position_lr_init * spatial_lr_scale, o.position_lr_final * spatial_lr_scale,
                           lr_delay_mult=o.position_lr_delay_mult, max_steps=o.position_lr_max_steps),
           "f_dc": o.feature_lr, "f_rest": o.feature_lr / 20.0, "opacity": o.opacity_lr, "scaling": o.scaling_lr,
           "rotation": o.rotation_lr}
    grads = {k: (leaf[k].grad.clone() if leaf[k].grad is not None else torch.zeros_like(leaf[k])) for k in GROUPS}
    opt = torch.optim.Adam([{"params": [leaf[k]], "lr": lrs[k], "name": k} for k in GROUPS], lr=0.0, eps=1e-15)
    if adam_state:
        for k in GROUPS:
            if k in adam_state:
                st, m, v = adam_state[k]
                opt.state[leaf[k]] = dict(step=torch.tensor(float(st)), exp_avg=m.cpu().to(dtype).clone(),
                                          exp_avg_sq=v.cpu().to(dtype).clone())
    opt.step()
    return dict(losses={k: float(v) for k, v in L.items()}, total=float(total), grads=grads,
                params={k: leaf[k].detach().clone() for k in GROUPS}, radii=data["radii"], lrs=lrs,
                densify_grad=data["viewspace_points_densify"].grad, stats=data["stats"])
