"""Training losses, HIP-backed.  Function names follow the reference's `tools/loss_utils.py`."""
import weakref

import torch

from . import _lib


class _L1SSIM(torch.autograd.Function):
    """One pass over both images: returns (mean |a-b|, mean ssim_map)."""

    @staticmethod
    def forward(ctx, img1, img2):
        lib = _lib.load()
        a = img1.detach().contiguous().float()
        b = img2.detach().contiguous().float()
        C, H, W = a.shape
        assert C == 3, "l1_ssim expects [3,H,W] images"
        sums = torch.empty(lib.vcr_sums_elems(2), dtype=torch.float64, device=a.device)
        res = torch.empty(2, dtype=torch.float32, device=a.device)
        need = img1.requires_grad
        part = torch.empty(9, H, W, dtype=torch.float32, device=a.device) if need else None
        _lib.check(lib.vcr_l1_ssim_forward(H, W, a.data_ptr(), b.data_ptr(), sums.data_ptr(), res.data_ptr(),
                                           part.data_ptr() if need else None, 0, _lib.stream_of(a)))
        ctx.save_for_backward(a, b, part)
        return res[0], res[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        lib = _lib.load()
        a, b, part = ctx.saved_tensors
        _, H, W = a.shape
        d = torch.empty_like(a)
        gl = g_l1.contiguous().float().reshape(1)
        gs = g_ssim.contiguous().float().reshape(1)
        _lib.check(lib.vcr_l1_ssim_backward(H, W, a.data_ptr(), b.data_ptr(), part.data_ptr(), gl.data_ptr(),
                                            gs.data_ptr(), d.data_ptr(), _lib.stream_of(a)))
        return d, None


def l1_ssim(network_output, gt):
    """Fused (l1_loss, ssim) pair (`tools/loss_utils.py:36,61-92`)."""
    return _L1SSIM.apply(network_output, gt)


def l1_loss(network_output, gt):
    """`tools/loss_utils.py:36-37` for [3,H,W] images (shares the fused kernel)."""
    if network_output.dim() == 3 and network_output.shape[0] == 3:
        return _L1SSIM.apply(network_output, gt)[0]
    return torch.abs(network_output - gt).mean()     # small per-Gaussian vectors (l1_scale)


def ssim(img1, img2, window_size=11, size_average=True):
    """`tools/loss_utils.py:61-92` (11x11, sigma 1.5, zero padding, C1=.01^2, C2=.03^2, global mean)."""
    assert window_size == 11 and size_average
    return _L1SSIM.apply(img1, img2)[1]


class _NormalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, wsrc, exp_t, mask, gt_grad, depth, depth_max):
        lib = _lib.load()
        p = pred.detach().contiguous().float().view(-1, 3)
        g = gt.detach().contiguous().float().view(-1, 3)
        w = None if wsrc is None else wsrc.detach().contiguous().float().view(-1, 3)
        m = None if mask is None else mask.detach().contiguous().view(-1).to(torch.uint8)
        d = None if depth is None else depth.detach().contiguous().float().view(-1)
        P = p.shape[0]
        sums = torch.empty(lib.vcr_sums_elems(3), dtype=torch.float64, device=p.device)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        _lib.check(lib.vcr_normal_loss_forward(P, p.data_ptr(), g.data_ptr(), None if w is None else w.data_ptr(),
                                               float(exp_t), None if m is None else m.data_ptr(),
                                               None if d is None else d.data_ptr(), float(depth_max), sums.data_ptr(),
                                               loss.data_ptr(), 0, _lib.stream_of(p)))
        ctx.save_for_backward(p, g, w, m, d, sums)
        ctx.exp_t, ctx.shape, ctx.gt_grad, ctx.depth_max = float(exp_t), pred.shape, bool(gt_grad), float(depth_max)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        p, g, w, m, d, sums = ctx.saved_tensors
        go = gout.contiguous().float().reshape(1)
        dp = torch.empty_like(p)
        dg = torch.empty_like(p) if ctx.gt_grad else None
        _lib.check(lib.vcr_normal_loss_backward(p.shape[0], p.data_ptr(), g.data_ptr(), None if w is None else w.data_ptr(),
                                                ctx.exp_t, None if m is None else m.data_ptr(),
                                                None if d is None else d.data_ptr(), ctx.depth_max, sums.data_ptr(),
                                                go.data_ptr(), dp.data_ptr(), None if dg is None else dg.data_ptr(), 0,
                                                _lib.stream_of(p)))
        return dp.view(ctx.shape), (dg.view(ctx.shape) if dg is not None else None), None, None, None, None, None, None


def normal_loss(normal_pred, normal_gt, weight_src=None, exp_t=0.0, mask=None, depth=None, depth_max=0.0):
    """Fused form of the reference's D-Normal chain (`trainer.py:266-280`):
        w = cos_weight(weight_src.detach(), gt, exp_t);  monosdf_normal_loss(pred[mask], gt[mask], w[mask])
    i.e. mean_mask(w |p-g|_1) + mean_mask(w (1 - p.g)) with w = exp((<weight_src,g> - 1)/exp_t)
    (`tools/loss_utils.py:122-143`).  Gradients flow to `normal_pred` and, when it requires grad
    (normal-consistency loss, `trainer.py:289-293`), to `normal_gt`.  `depth`/`depth_max` fuse the
    `rendered_depth < extent * mask_depth_thr` part of the mask (`gaussian_renderer/__init__.py:128-131`)."""
    return _NormalLoss.apply(normal_pred, normal_gt, weight_src, exp_t, mask, normal_gt.requires_grad, depth, depth_max)


class _ScaleReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling_raw, xyz, trans, scale):
        lib = _lib.load()
        s = scaling_raw.detach().contiguous()
        x = xyz.detach().contiguous()
        t, sc = trans.detach().contiguous().float(), scale.detach().contiguous().float()
        sums = torch.empty(lib.vcr_sums_elems(3), dtype=torch.float64, device=s.device)
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        _lib.check(lib.vcr_scale_reg_forward(s.shape[0], s.data_ptr(), x.data_ptr(), t.data_ptr(), sc.data_ptr(),
                                             sums.data_ptr(), loss.data_ptr(), 0, _lib.stream_of(s)))
        ctx.save_for_backward(s, x, t, sc, sums)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        s, x, t, sc, sums = ctx.saved_tensors
        go = gout.contiguous().float().reshape(1)
        d = torch.empty_like(s)
        _lib.check(lib.vcr_scale_reg_backward(s.shape[0], s.data_ptr(), x.data_ptr(), t.data_ptr(), sc.data_ptr(),
                                              sums.data_ptr(), go.data_ptr(), d.data_ptr(), _lib.stream_of(s)))
        return d, None, None, None


def scale_regulariser(scaling_raw, xyz, trans, scale):
    """l1_scale (`trainer.py:243-245`): mean of the smallest activated scale over the Gaussians inside the
    normalised bounding box (`tools/math_utils.py:50-74`), one kernel each way."""
    return _ScaleReg.apply(scaling_raw, xyz, trans, scale)


def monosdf_normal_loss(normal_pred, normal_gt, weight=None):
    """`tools/loss_utils.py:122-132`.  Unweighted calls run on the fused HIP kernel; an explicit
    per-pixel weight tensor (only produced by `cos_weight`, which `normal_loss` fuses) falls back to
    the two-line torch expression for API compatibility."""
    if weight is None:
        return normal_loss(normal_pred, normal_gt)
    l1 = (weight * torch.abs(normal_pred - normal_gt).sum(dim=-1)).mean()
    cos = (weight * (1.0 - torch.sum(normal_pred * normal_gt, dim=-1))).mean()
    return l1 + cos


def cos_weight(render_normal, gt_normal, exp_t=1.0):
    """`tools/loss_utils.py:135-143` (kept for API compatibility; `normal_loss` computes it in-kernel)."""
    cos = torch.sum(render_normal * gt_normal, dim=-1)
    cos = torch.exp((cos - 1) / exp_t) if exp_t > 0 else torch.ones_like(cos)
    return cos.detach()


class _Entropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity_raw, xyz, trans, scale):
        lib = _lib.load()
        o = opacity_raw.detach().contiguous().float()
        x = None if xyz is None else xyz.detach().contiguous().float()
        t = None if xyz is None else trans.detach().contiguous().float()
        sc = None if xyz is None else scale.detach().contiguous().float()
        sums = torch.empty(lib.vcr_sums_elems(3), dtype=torch.float64, device=o.device)
        loss = torch.empty(1, dtype=torch.float32, device=o.device)
        p = lambda v: None if v is None else v.data_ptr()
        _lib.check(lib.vcr_entropy_forward(o.numel(), o.data_ptr(), p(x), p(t), p(sc), sums.data_ptr(), loss.data_ptr(),
                                           _lib.stream_of(o)))
        ctx.save_for_backward(o, x, t, sc, sums)
        ctx.shape = opacity_raw.shape
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        o, x, t, sc, sums = ctx.saved_tensors
        d = torch.empty_like(o)
        go = gout.contiguous().float().reshape(1)
        p = lambda v: None if v is None else v.data_ptr()
        _lib.check(lib.vcr_entropy_backward(o.numel(), o.data_ptr(), p(x), p(t), p(sc), sums.data_ptr(), go.data_ptr(),
                                            d.data_ptr(), _lib.stream_of(o)))
        return d.view(ctx.shape), None, None, None


def entropy_regulariser(opacity_raw, xyz=None, trans=None, scale=None):
    """`entropy_loss(get_opacity[inside])` of `trainer.py:247-249` on the RAW opacities, the sigmoid and the bounding-box
    mask (`tools/math_utils.py:70-74`) fused: one kernel each way."""
    return _Entropy.apply(opacity_raw, xyz, trans, scale)


def entropy_loss(opacity):
    """`tools/loss_utils.py:30-33` on activated opacities (API compatibility; the trainer uses `entropy_regulariser`)."""
    return (-opacity * torch.log(opacity + 1e-6) - (1 - opacity) * torch.log(1 - opacity + 1e-6)).mean()


class _Curv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normal, mask):
        lib = _lib.load()
        n = normal.detach().contiguous().float()
        H, W = n.shape[:2]
        m = mask.detach().reshape(H, W).contiguous().to(torch.uint8)
        sums = torch.empty(lib.vcr_sums_elems(1), dtype=torch.float64, device=n.device)
        loss = torch.empty(1, dtype=torch.float32, device=n.device)
        _lib.check(lib.vcr_curv_forward(H, W, n.data_ptr(), m.data_ptr(), sums.data_ptr(), loss.data_ptr(), _lib.stream_of(n)))
        ctx.save_for_backward(n, m)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        n, m = ctx.saved_tensors
        H, W = n.shape[:2]
        d = torch.empty_like(n)
        go = gout.contiguous().float().reshape(1)
        _lib.check(lib.vcr_curv_backward(H, W, n.data_ptr(), m.data_ptr(), go.data_ptr(), d.data_ptr(), _lib.stream_of(n)))
        return d, None


def curv_loss(normal, mask):
    """`l1_loss(normal2curv(normal, mask), 0)` (`tools/loss_utils.py:287-300`, `trainer.py:282-287`): normal [H,W,3],
    mask [H,W] or [H,W,1] (bool / 0-1), fused forward and backward."""
    return _Curv.apply(normal, mask)


def normal2curv(normal, mask=None):
    """`tools/loss_utils.py:287-300` (the map itself, for visualisation; the loss runs on `curv_loss`)."""
    pad = torch.nn.functional.pad
    n = pad(normal[None], [0, 0, 1, 1, 1, 1], mode="replicate")
    m = pad(mask[None].to(torch.float32), [0, 0, 1, 1, 1, 1], mode="replicate").to(torch.bool)
    c = n[:, 1:-1, 1:-1] * m[:, 1:-1, 1:-1]
    tot = sum((n[:, ys, xs] - c) * m[:, ys, xs] for ys, xs in ((slice(None, -2), slice(1, -1)), (slice(1, -1), slice(None, -2)),
                                                             (slice(2, None), slice(1, -1)), (slice(1, -1), slice(2, None))))
    return (tot[0] * mask).norm(1, -1, True)


class _EdgeAwareMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt_image, dmap):
        lib = _lib.load()
        g = gt_image.detach().contiguous().float()
        d = dmap.detach().contiguous().float()
        H, W = g.shape[-2:]
        sums = torch.empty(lib.vcr_sums_elems(1), dtype=torch.float64, device=g.device)
        loss = torch.empty(1, dtype=torch.float32, device=g.device)
        _lib.check(lib.vcr_edge_aware_forward(H, W, g.data_ptr(), d.data_ptr(), sums.data_ptr(), loss.data_ptr(), _lib.stream_of(g)))
        ctx.save_for_backward(g)
        ctx.shape = dmap.shape
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        (g,) = ctx.saved_tensors
        H, W = g.shape[-2:]
        d = torch.empty(H, W, dtype=torch.float32, device=g.device)
        go = gout.contiguous().float().reshape(1)
        _lib.check(lib.vcr_edge_aware_backward(H, W, g.data_ptr(), go.data_ptr(), d.data_ptr(), _lib.stream_of(g)))
        return None, d.view(ctx.shape)


def edge_aware_mean(gt_image, distortion_map):
    """`get_edge_aware_distortion_map(gt_image, map).mean()` (`tools/normal_utils.py:57-66`, `trainer.py:295-303`),
    forward and backward in one kernel each."""
    return _EdgeAwareMean.apply(gt_image, distortion_map)


def psnr(img1, img2):
    """`tools/image_utils.py:17-19`."""
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


class _SemanticCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sem_planes, weight, bias, labels):
        lib = _lib.load()
        sem = sem_planes.detach().contiguous().float()
        S, H, W = sem.shape
        K = weight.shape[0]
        w = weight.detach().reshape(K, S).contiguous().float()
        b = bias.detach().contiguous().float()
        lab = labels.detach().reshape(-1).contiguous().long()
        sums = torch.empty(lib.vcr_sums_elems(1), dtype=torch.float64, device=sem.device)
        loss = torch.empty(1, dtype=torch.float32, device=sem.device)
        _lib.check(lib.vcr_semantic_ce_forward(H * W, S, K, sem.data_ptr(), w.data_ptr(), b.data_ptr(), lab.data_ptr(),
                                               sums.data_ptr(), loss.data_ptr(), _lib.stream_of(sem)))
        ctx.save_for_backward(sem, lab, w, b)
        ctx.shapes = (weight.shape, bias.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        sem, lab, w, b = ctx.saved_tensors
        S, H, W = sem.shape
        K = w.shape[0]
        dsem = torch.empty_like(sem)
        dW = torch.empty(K, S, dtype=torch.float32, device=sem.device)
        db = torch.empty(K, dtype=torch.float32, device=sem.device)
        go = gout.contiguous().float().reshape(1)
        _lib.check(lib.vcr_semantic_ce_backward(H * W, S, K, sem.data_ptr(), w.data_ptr(), b.data_ptr(), lab.data_ptr(),
                                                go.data_ptr(), dsem.data_ptr(), dW.data_ptr(), db.data_ptr(), _lib.stream_of(sem)))
        return dsem, dW.view(ctx.shapes[0]), db.view(ctx.shapes[1]), None


_CHECKED_LABELS = {}


def _check_label_range(labels, K):
    """`F.cross_entropy` refuses a target outside [0, K) (device-side assert); the fused kernel would silently count such a pixel
    as zero loss while still dividing by all pixels.  Label images are per-camera constants, so the range is checked ONCE per
    tensor (one min / max read-back), not per step.  `ignore_index` is not supported (the reference does not pass one)."""
    key = (labels.data_ptr(), labels.numel(), int(labels._version), int(K))
    seen = _CHECKED_LABELS.get(id(labels))
    if seen is not None and seen[0] == key and seen[1]() is labels:          # (the weak reference guards against a recycled id)
        return
    if labels.numel():
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= K:
            raise ValueError(f"semantic_loss: labels span [{lo}, {hi}] but the classifier has {K} classes "
                             "(F.cross_entropy would refuse them too)")
    if len(_CHECKED_LABELS) > 4096:
        _CHECKED_LABELS.clear()
    _CHECKED_LABELS[id(labels)] = (key, weakref.ref(labels))


def semantic_loss(sem_planes, classifier, labels):
    """`F.cross_entropy(classifier(sem)[0].permute(1, 2, 0).view(-1, K), labels.view(-1)) / log(K)`
    (`gaussian_renderer/__init__.py:146-148`, `trainer.py:304-307`) with the 1x1-conv classifier, the log-softmax and the
    NLL fused: sem_planes [S,H,W] = rows 8..8+S of the rasterizer output, classifier = the model's Conv2d(S, K, 1)."""
    _check_label_range(labels, classifier.weight.shape[0])
    return _SemanticCE.apply(sem_planes, classifier.weight, classifier.bias, labels)
