"""Depth -> TSDF input path (`tools/depth2mesh.py:22-102`, `tools/graphics_utils.py:134-141`): what happens to a rendered
depth map between `render()` and the TSDF integrator -- back-projection, alpha / mask thresholds, bounding-box test -- as
one HIP kernel.  The voxel-block TSDF itself stays with Open3D, as in the reference (out of scope)."""
import ctypes as C

import torch

from . import _lib


def _c2w_host(extrinsic_w2c):
    m = torch.inverse(extrinsic_w2c.detach().cpu().double()).float().contiguous().reshape(-1)
    return (C.c_float * 16)(*m.tolist())


def _call(depth, intr_scalars, extrinsic_w2c, trans=None, scale=None, alpha=None, alpha_thres=0.0, gt_alpha=None,
          want_depth=True, want_points=False):
    lib = _lib.load()
    d = depth.detach().reshape(depth.shape[-2:]).contiguous().float()
    H, W = d.shape
    dev = d.device
    p = lambda t: None if t is None else t.data_ptr()
    cf = lambda t: None if t is None else t.detach().reshape(-1).contiguous().float().to(dev)
    a, ga, tr, sc = cf(alpha), cf(gt_alpha), cf(trans), cf(scale)
    out = torch.empty_like(d) if want_depth else None
    cam = torch.empty(H, W, 3, device=dev) if want_points else None
    wld = torch.empty(H, W, 3, device=dev) if want_points else None
    fx, fy, cx, cy = intr_scalars
    _lib.check(lib.vcr_tsdf_depth_input(H, W, fx, fy, cx, cy, _c2w_host(extrinsic_w2c), p(tr), p(sc), d.data_ptr(), p(a),
                                        float(alpha_thres), p(ga), p(out), p(cam), p(wld), _lib.stream_of(d)))
    return out, cam, wld


def depth2point(depth_image, intrinsic_matrix, extrinsic_matrix, intr_scalars=None):
    """`tools/graphics_utils.py:134-141`: depth [H,W] -> (xyz_cam [H,W,3], xyz_world [H,W,3]); extrinsic = world-to-camera."""
    if intr_scalars is None:
        k = intrinsic_matrix.detach().cpu()
        intr_scalars = (float(k[0, 0]), float(k[1, 1]), float(k[0, 2]), float(k[1, 2]))
    _, cam, wld = _call(depth_image, intr_scalars, extrinsic_matrix, want_depth=False, want_points=True)
    return cam, wld


@torch.no_grad()
def tsdf_depth_input(render_pkg, view, model, alpha_thres=0.5):
    """The depth image `tsdf_fusion` hands to the integrator (`tools/depth2mesh.py:37-52`): rendered depth with
    alpha < alpha_thres, gt_alpha_mask < 0.5 and points outside the model's bounding box zeroed.  Returns [1,H,W]."""
    out, _, _ = _call(render_pkg["depth"], view.intr_scalars, view.world_view_transform.t(), model.trans, model.scale,
                      alpha=render_pkg["alpha"], alpha_thres=alpha_thres, gt_alpha=getattr(view, "gt_alpha_mask", None))
    return out[None]
