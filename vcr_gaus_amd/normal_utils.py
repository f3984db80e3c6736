"""Depth -> normal ("D-Normal") operators, HIP-backed.  Same names / argument meaning as the
reference's `tools/normal_utils.py`."""
import torch

from . import _lib


class _DepthToNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, fx, fy, cx, cy):
        lib = _lib.load()
        d = depth.detach().contiguous().float()
        H, W = d.shape[-2:]
        out = torch.empty(H, W, 3, dtype=torch.float32, device=d.device)
        _lib.check(lib.vcr_depth_to_normal_forward(H, W, fx, fy, cx, cy, d.data_ptr(), out.data_ptr(), _lib.stream_of(d)))
        ctx.save_for_backward(d)
        ctx.k = (fx, fy, cx, cy)
        ctx.shape = depth.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (d,) = ctx.saved_tensors
        H, W = d.shape[-2:]
        g = g.contiguous().float()
        scratch = torch.empty(H * W * 6, dtype=torch.float32, device=d.device)
        dd = torch.empty(H, W, dtype=torch.float32, device=d.device)
        _lib.check(lib.vcr_depth_to_normal_backward(H, W, *ctx.k, d.data_ptr(), g.data_ptr(), scratch.data_ptr(),
                                                    dd.data_ptr(), _lib.stream_of(d)))
        return dd.view(ctx.shape), None, None, None, None


def _intr_scalars(K):
    """(fx, fy, cx, cy) as Python floats.  Cameras cache them (`Camera.intr_scalars`) so that no
    device->host read happens on the hot path."""
    k = K.detach().cpu()
    return float(k[0, 0]), float(k[1, 1]), float(k[0, 2]), float(k[1, 2])


def compute_normals(depth_map, K, intr_scalars=None):
    """`tools/normal_utils.py:30-41`: back-project every pixel centre with K^-1, take the
    torch.gradient-style finite differences along columns / rows, normalise their cross product.
    depth_map: [1,H,W] or [H,W]; returns [H,W,3]."""
    fx, fy, cx, cy = intr_scalars if intr_scalars is not None else _intr_scalars(K)
    return _DepthToNormal.apply(depth_map, fx, fy, cx, cy)


class _NormalizeCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.detach().contiguous().float()
        _, H, W = x.shape
        out = torch.empty(H, W, 3, dtype=torch.float32, device=x.device)
        _lib.check(lib.vcr_normalize_chw_forward(H * W, x.data_ptr(), out.data_ptr(), _lib.stream_of(x)))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        _lib.check(lib.vcr_normalize_chw_backward(x.shape[1] * x.shape[2], x.data_ptr(), g.data_ptr(), dx.data_ptr(),
                                                  _lib.stream_of(x)))
        return dx


def normalize_rendered_normal(normal_chw):
    """[3,H,W] blended normal -> unit [H,W,3] (`gaussian_renderer/__init__.py:133-134`)."""
    return _NormalizeCHW.apply(normal_chw)


def get_edge_aware_distortion_map(gt_image, distortion_map):
    """`tools/normal_utils.py:57-66` (image-sized elementwise glue; off in the TNT/360 configs)."""
    c = gt_image[:, 1:-1, 1:-1]
    g = torch.stack([(c - gt_image[:, 1:-1, :-2]).abs().mean(0), (c - gt_image[:, 1:-1, 2:]).abs().mean(0),
                     (c - gt_image[:, :-2, 1:-1]).abs().mean(0), (c - gt_image[:, 2:, 1:-1]).abs().mean(0)], -1)
    w = torch.nn.functional.pad(torch.exp(-g.max(-1)[0]), (1, 1, 1, 1))
    return distortion_map * w
