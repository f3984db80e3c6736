"""`render()` and the forward-only raster modes: same signature, dictionary keys and shapes as the
reference's `gaussian_renderer/__init__.py` (`render :22-164`, `count_render :250-355`,
`visi_acc_render :467-571`), with every tensor op on the path executed by a HIP kernel:
fused activation/normal kernel -> rasterizer -> normal normalisation -> depth-to-normal."""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

from .gaussian_model import fused_activate
from .normal_utils import compute_normals, normalize_rendered_normal


_ZEROS = {}


def _zero_holder(like):
    """An [N,3] zero tensor that is never written (the rasterizer only reads the holders' shape), cached per size."""
    key = (like.shape[0], like.device)
    z = _ZEROS.get(key)
    if z is None:
        _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros_like(like)
    return z


def _settings(cam, pc, bg_color, scaling_modifier, debug, f_count):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=debug, f_count=f_count)


def _cam_rotation(cam, device):
    R = getattr(cam, "R_w2c", None)
    if R is None:   # reference cameras only carry the numpy c2w rotation (`scene/cameras.py:28`)
        R = torch.tensor(cam.R.T, dtype=torch.float32)
    return R.to(device)


class _RenderOut(dict):
    """The reference's return dictionary; with `lazy_mask` the [N] `visibility_filter` (radii > 0) is only materialised
    when somebody reads it (the fused trainer hands `radii` to the densification-statistics kernel instead)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lazy = {}                      # key -> thunk: entries materialised on first read (e.g. render_sem)

    def __missing__(self, key):
        if key == "visibility_filter":
            v = self["radii"] > 0
            self[key] = v
            return v
        if key in self.lazy:
            v = self.lazy.pop(key)()
            self[key] = v
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return key == "visibility_filter" or key in self.lazy or dict.__contains__(self, key)


def render(viewpoint_camera, pc, cfg, bg_color, scaling_modifier=1.0, override_color=None, return_normal=True,
           is_all=True, dirs=None, mask_depth_thr=0.8, lazy_mask=False, geometry=True, raster_options=None, dist_channels=True):
    """Background tensor (bg_color) must be on the GPU.  Returns the reference's dict:
    render[3,H,W] depth[1,H,W] normal[H,W,3] est_normal[H,W,3] alpha[1,H,W] viewspace_points[N,3]
    viewspace_points_densify[N,3] visibility_filter[N] mask[H,W] radii[N] (+render_sem); beyond the reference: "raster", the
    call's `RasterRecord` (V, R; the SH-gradient factors after an `sh_grad="rgb"` backward).  `raster_options`: `RasterOptions`.
    `dist_channels=False`: skip the distortion / depth-variance channels the configuration asks for (the trainer does while
    their losses are not active yet)."""
    dev = pc.get_xyz.device
    # gradient holders for the 2D means (`:31-37`); leaves, so `.grad` is populated without the reference's `+ 0` copies
    grad_on = torch.is_grad_enabled()
    z = _zero_holder(pc.get_xyz)        # shared all-zero storage; fresh leaf views so each call gets its own .grad
    screenspace_points = z.detach().requires_grad_(grad_on)
    screenspace_points_densify = z.detach().requires_grad_(grad_on)

    rs = _settings(viewpoint_camera, pc, bg_color, scaling_modifier, cfg.pipline.debug, 0)
    lw = cfg.optim.loss_weight
    want_var = dist_channels and getattr(lw, "depth_var", 0) > 0
    want_dist = dist_channels and getattr(lw, "distortion", 0) > 0 and not want_var
    rasterizer = GaussianRasterizer(raster_settings=rs, num_dist=2 if want_var else (1 if want_dist else None),
                                    options=raster_options)

    act = fused_activate(pc, viewpoint_camera.camera_center, _cam_rotation(viewpoint_camera, dev), return_normal)
    scales, rotations, opacity = act[:3]
    normals_precomp = act[3] if return_normal else None
    cov3D_precomp = None
    if cfg.pipline.compute_cov3D_python:
        cov3D_precomp, scales, rotations = pc.get_covariance(scaling_modifier), None, None

    # SH coefficients go to the kernel in the model's split storage (no torch.cat of get_features)
    shs, shs_rest, colors_precomp = (pc._features_dc, pc._features_rest, None) if override_color is None \
        else (None, None, override_color)
    if override_color is None and cfg.pipline.convert_SHs_python:        # `gaussian_renderer/__init__.py:81-87`
        from .sh_utils import eval_sh
        shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
        sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True))
        shs, shs_rest, colors_precomp = None, None, torch.clamp_min(sh2rgb + 0.5, 0.0)
    with_sem = cfg.optim.loss_weight.semantic > 0 and getattr(pc, "enable_semantic", False) and pc.get_objects.numel() > 0
    sem_feats = pc.get_objects.squeeze(1) if with_sem else None

    rendered_out, radii = rasterizer(
        means3D=pc.get_xyz, means2D=screenspace_points, means2D_densify=screenspace_points_densify, shs=shs,
        colors_precomp=colors_precomp, normals_precomp=normals_precomp, semantics_precomp=sem_feats,
        opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, dirs=dirs, inside=None,
        shs_rest=shs_rest)

    rendered_image, rendered_depth, rendered_normal, rendered_alpha = rendered_out[:8].split([3, 1, 3, 1], dim=0)
    cam_mask = viewpoint_camera.mask.bool() if hasattr(viewpoint_camera, "mask") else None
    mask = None
    if not lazy_mask:      # reference behaviour (:125-131); the fused trainer passes depth + threshold to the loss kernel
        with torch.no_grad():
            mask = cam_mask
            if cfg.optim.mask_depth_thr > 0:
                m1 = (rendered_depth < (pc.extent * cfg.optim.mask_depth_thr)).squeeze(0)
                mask = m1 if mask is None else (mask & m1)
            if mask is None:
                mask = torch.ones(rendered_depth.shape[1:], dtype=torch.bool, device=dev)

    normal = est_normal = None
    if geometry:       # the fused loss node (fused_losses.py) derives both itself from `render_out`
        normal = normalize_rendered_normal(rendered_normal)
        est_normal = compute_normals(rendered_depth, viewpoint_camera.intr,
                                     getattr(viewpoint_camera, "intr_scalars", None))
    out = _RenderOut({"render": rendered_image, "depth": rendered_depth, "normal": normal, "est_normal": est_normal,
                      "alpha": rendered_alpha, "viewspace_points": screenspace_points,
                      "viewspace_points_densify": screenspace_points_densify, "mask": mask,
                      "mask_static": cam_mask, "radii": radii, "render_out": rendered_out, "raster": rasterizer.record})
    if not lazy_mask:
        out["visibility_filter"] = radii > 0
    if with_sem:
        sem = rendered_out[8:8 + cfg.model.ch_sem_feat]
        out["sem_planes"] = sem                          # the trainer's fused semantic loss consumes these directly
        if lazy_mask:
            out.lazy["render_sem"] = lambda: pc.classifier(sem[None])[0].permute(1, 2, 0)
        else:
            out["render_sem"] = pc.classifier(sem[None])[0].permute(1, 2, 0)
    if want_var:                                    # gaussian_renderer/__init__.py:154-158
        d1, d2 = rendered_out[-2:-1], rendered_out[-1:]
        out["depth_var"] = d2 / rendered_alpha - (d1 / rendered_alpha) ** 2
    if want_dist:                                   # gaussian_renderer/__init__.py:160-162
        out["distortion"] = rendered_out[-1:]
    return out


def _forward_only(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, f_count):
    rs = _settings(viewpoint_camera, pc, bg_color, scaling_modifier, pipe.debug, f_count)
    rasterizer = GaussianRasterizer(raster_settings=rs)
    dev = pc.get_xyz.device
    with torch.no_grad():
        scales, rotations, opacity = fused_activate(pc, viewpoint_camera.camera_center,
                                                    _cam_rotation(viewpoint_camera, dev), False)
    shs, shs_rest, colors = (pc._features_dc, pc._features_rest, None) if override_color is None \
        else (None, None, override_color)
    screenspace_points = torch.zeros_like(pc.get_xyz)
    res = rasterizer(means3D=pc.get_xyz, means2D=screenspace_points, means2D_densify=None, shs=shs,
                     colors_precomp=colors, normals_precomp=None, semantics_precomp=None, opacities=opacity,
                     scales=scales, rotations=rotations, cov3D_precomp=None, shs_rest=shs_rest)
    return res, screenspace_points


def count_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """f_count=1 (`gaussian_renderer/__init__.py:250-355`): per-Gaussian hit count and sum of alpha*T."""
    (count, score, image, radii), sp = _forward_only(viewpoint_camera, pc, pipe, bg_color, scaling_modifier,
                                                     override_color, 1)
    return {"render": image, "viewspace_points": sp, "visibility_filter": radii > 0, "radii": radii,
            "gaussians_count": count, "important_score": score}


def visi_acc_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """f_count=3 (`gaussian_renderer/__init__.py:467-571`): per-Gaussian visibility count only."""
    (count, radii), sp = _forward_only(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, 3)
    return {"viewspace_points": sp, "visibility_filter": radii > 0, "radii": radii, "countlist": count}


@torch.no_grad()
def visibility_counts(cameras, pc, pipe, scaling_modifier=1.0, flags_only=False, count=None, inflight=0):
    """`get_visi_list` (`tools/prune.py:51-69`: one `visi_acc_render` per camera, `countlist`s summed) for a whole list of
    cameras in one batched library call per image size.  -> int32 [N]: the summed `countlist` (or, with `flags_only`, 1
    where it is > 0 -- the only thing `get_visi_list` derives from it)."""
    from .rasterizer import visibility_batch
    dev = pc.get_xyz.device
    n = pc.get_xyz.shape[0]
    if count is None:
        count = torch.zeros(n, dtype=torch.int32, device=dev)
    if not cameras or n == 0:
        return count
    # activations do not depend on the camera when no normals are asked for
    scales, rotations, opacity = fused_activate(pc, cameras[0].camera_center, _cam_rotation(cameras[0], dev), False)
    cov = None
    if pipe.compute_cov3D_python:
        cov, scales, rotations = pc.get_covariance(scaling_modifier), None, None
    groups = {}
    for cam in cameras:
        groups.setdefault((int(cam.image_height), int(cam.image_width)), []).append(cam)
    for (h, w), cams in groups.items():
        st = getattr(cams[0], "_stack", None)
        if st is not None and st[0].device == dev and st[0].shape[0] == len(cams) and \
                all(getattr(c, "_stack", None) is st and c._stack_index == i for i, c in enumerate(cams)):
            vm, pm, cc = st[0], st[1], st[2]                  # (`SampleCam.batch`: the cameras ARE the rows of one stack)
        else:
            vm = torch.stack([c.world_view_transform.to(dev) for c in cams])
            pm = torch.stack([c.full_proj_transform.to(dev) for c in cams])
            cc = torch.stack([c.camera_center.to(dev) for c in cams])
        visibility_batch(vm, pm, cc, [math.tan(c.FoVx * 0.5) for c in cams], [math.tan(c.FoVy * 0.5) for c in cams], h, w,
                         pc.get_xyz, opacity, scales, rotations, cov, scaling_modifier, flags_only, count, inflight)
    return count


def visi_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """f_count=2 (`gaussian_renderer/__init__.py:358-464`): per-Gaussian visibility count + importance + image."""
    (count, score, image, radii), sp = _forward_only(viewpoint_camera, pc, pipe, bg_color, scaling_modifier,
                                                     override_color, 2)
    return {"render": image, "viewspace_points": sp, "visibility_filter": radii > 0, "radii": radii,
            "countlist": count, "important_score": score}


def render_fast(viewpoint_camera, pc, cfg, bg_color, scaling_modifier=1.0, override_color=None):
    """Colour-only differentiable render (`gaussian_renderer/__init__.py:167-247`): no normals, no per-pixel ray
    directions, settings built without `f_count`; returns the first three output channels."""
    grad_on = torch.is_grad_enabled()
    screenspace_points = _zero_holder(pc.get_xyz).detach().requires_grad_(grad_on)
    rs = _settings(viewpoint_camera, pc, bg_color, scaling_modifier, cfg.pipline.debug, 0)
    rasterizer = GaussianRasterizer(raster_settings=rs, num_dist=0)
    scales, rotations, opacity = fused_activate(pc, viewpoint_camera.camera_center,
                                                _cam_rotation(viewpoint_camera, pc.get_xyz.device), False)
    cov3D_precomp = None
    if cfg.pipline.compute_cov3D_python:
        cov3D_precomp, scales, rotations = pc.get_covariance(scaling_modifier), None, None
    shs, shs_rest, colors_precomp = (pc._features_dc, pc._features_rest, None) if override_color is None \
        else (None, None, override_color)
    if override_color is None and cfg.pipline.convert_SHs_python:
        from .sh_utils import eval_sh
        shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
        sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True))
        shs, shs_rest, colors_precomp = None, None, torch.clamp_min(sh2rgb + 0.5, 0.0)
    out, radii = rasterizer(means3D=pc.get_xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                            opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                            shs_rest=shs_rest)
    return {"render": out[:3], "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
