"""The whole image-space loss of `Trainer._compute_loss` (`trainer.py:233-321`) as ONE autograd node.

Same HIP kernels as `loss_utils` / `normal_utils`, but driven from a single forward and a single backward: the
rasterizer output [C,H,W] goes in, the weighted total comes out, and backward writes dL/d(out) plane by plane into one
buffer (colour <- L1+SSIM, depth <- depth-to-normal adjoint, normal <- normalisation adjoint) -- no per-loss autograd
nodes, no `split`/`cat` of channel gradients, ~12 Python-level operator calls less per step."""
import torch

from . import _lib

NAMES = ["l1", "ssim", "l1_scale", "mono_normal", "depth_normal", "consistent_normal"]
_ONES = {}


def unit_seed(device):
    """A cached scalar 1.0 to pass to `loss.backward(...)`: spares autograd's ones_like fill, and the loss node skips
    the multiplication of its weight vector by it."""
    k = str(device)
    if k not in _ONES:
        _ONES[k] = torch.ones((), device=device)
    return _ONES[k]


class _FusedLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, scaling_raw, xyz, gt_image, gt_normal, mask, intr, wvec, active, exp_t, depth_max, trans, scale,
                sink=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)          # (no zero-filled gradient for the non-differentiable `res` output)
        o = out.detach().contiguous()
        C, H, W = o.shape
        P = H * W
        dev = o.device
        st = _lib.stream_of(o)
        base = o.data_ptr()
        n2, n3, n9 = lib.vcr_sums_elems(2), lib.vcr_sums_elems(3), lib.vcr_sums_elems(9)
        # ONE buffer for all reductions.  It is reused
        # from step to step: the finalize kernel leaves the slots zeroed again (a fresh zero-filled one only if the previous
        # forward's backward has not run yet).
        cache = sink.sums if (sink is not None and sink.sums is not None) else None      # the caller's cache, or none
        ent = cache.get(dev) if cache is not None else None
        if ent is not None and not ent[1] and ent[0].numel() == n2 + n3 + n9:
            sums = ent[0]
        else:
            sums = torch.zeros(n2 + n3 + n9, dtype=torch.float64, device=dev)
            ent = [sums, False]
            if cache is not None:
                cache[dev] = ent
        ent[1] = torch.is_grad_enabled() and ent[0] is sums
        ctx.sums_entry = ent if ent[0] is sums else None
        res8 = torch.empty(8, dtype=torch.float32, device=dev)      # (fresh per call: the returned loss values must not alias the reused buffer)
        res, total = res8[:6], torch.empty((), device=dev)       # (not a view: a view output costs a select_backward)
        s_ssim, s_scale, s_nrm = sums.data_ptr(), sums.data_ptr() + 8 * n2, sums.data_ptr() + 8 * (n2 + n3)
        rp = lambda k: res.data_ptr() + 4 * k
        gi = gt_image.detach().contiguous()
        part = torch.empty(9, H, W, device=dev)
        sr, xz = scaling_raw.detach().contiguous(), xyz.detach().contiguous()
        gn = None if gt_normal is None else gt_normal.detach().contiguous()
        m = None if mask is None else mask.detach().contiguous().view(-1).to(torch.uint8)
        nbits = (1 if active[3] else 0) | (2 if active[4] else 0) | (4 if active[5] else 0)
        try:
            _lib.check(lib.vcr_l1_ssim_forward(H, W, base, gi.data_ptr(), s_ssim, rp(0), part.data_ptr(), 3, st))
            if active[2]:
                _lib.check(lib.vcr_scale_reg_forward(sr.shape[0], sr.data_ptr(), xz.data_ptr(), trans.data_ptr(), scale.data_ptr(),
                                                     s_scale, rp(2), 3, st))
            if nbits:      # mono_normal, depth_normal, consistent_normal: one kernel (+ one finalize) for all three
                _lib.check(lib.vcr_normal_losses_forward(H, W, *intr, base + 3 * P * 4, base + 4 * P * 4,
                                                         None if gn is None else gn.data_ptr(), None if m is None else m.data_ptr(),
                                                         float(depth_max), float(exp_t), nbits, s_nrm, rp(3), 3, st))
            # ONE finalize for all reductions; total = sum_k w_k L_k with the ssim entry meaning (1 - ssim): wvec[1] = -w_ssim,
            # the constant +w_ssim is added in-kernel
            _lib.check(lib.vcr_finalize_losses(H, W, s_ssim, s_scale if active[2] else None, s_nrm if nbits else None, res.data_ptr(),
                                               wvec.data_ptr(), 1, total.data_ptr(), st))
        except Exception:
            # a kernel failed between the first accumulation and the finalize that re-zeroes the slots: the reused buffer may
            # hold partial sums -- drop it, the next forward starts from a fresh zero-filled one
            if cache is not None and cache.get(dev) is ent:
                del cache[dev]
            raise
        ctx.save_for_backward(o, gi, part, sums, sr, xz, gn, m, wvec, trans, scale)
        ctx.meta = (H, W, C, tuple(intr), tuple(active), float(exp_t), float(depth_max), n2, n3, nbits)
        ctx.sink = sink
        ctx.mark_non_differentiable(res)
        return total, res

    @staticmethod
    def backward(ctx, g_total, _g_res):
        if ctx.sums_entry is not None:
            ctx.sums_entry[1] = False               # (the reduction results are read below, on this stream, before any reuse)
        if g_total is None:
            return (None,) * 14
        lib = _lib.load()
        o, gi, part, sums, sr, xz, gn, m, wvec, trans, scale = ctx.saved_tensors
        H, W, C, intr, active, exp_t, depth_max, n2, n3, nbits = ctx.meta
        P = H * W
        dev = o.device
        st = _lib.stream_of(o)
        one = _ONES.get(str(dev))
        seeds = wvec if (one is not None and g_total.data_ptr() == one.data_ptr()) else (g_total * wvec).contiguous()
        gp = lambda k: seeds.data_ptr() + 4 * k
        s_scale, s_nrm = sums.data_ptr() + 8 * n2, sums.data_ptr() + 8 * (n2 + n3)
        dout = torch.empty_like(o)
        dbase = dout.data_ptr()
        if C > 8 or (C > 7 and not nbits):
            dout[7:].zero_()                                       # (the alpha plane is zeroed by the normal-loss backward; the
                                                                   #  semantic planes get their gradient from semantic_loss)
        _lib.check(lib.vcr_l1_ssim_backward(H, W, o.data_ptr(), gi.data_ptr(), part.data_ptr(), gp(0), gp(1), dbase, st))
        base = o.data_ptr()
        if nbits:
            scratch = torch.empty(P * 6, device=dev)
            _lib.check(lib.vcr_normal_losses_backward(H, W, *intr, base + 3 * P * 4, base + 4 * P * 4,
                                                      None if gn is None else gn.data_ptr(), None if m is None else m.data_ptr(),
                                                      float(depth_max), float(exp_t), nbits, s_nrm, gp(3), scratch.data_ptr(),
                                                      dbase + 3 * P * 4, dbase + 4 * P * 4, st))
        else:
            dout[3:7].zero_()
        d_sc = None
        if active[2] and ctx.sink is not None and ctx.sink.armed and ctx.sink.scale_reg is None:
            # fused static tail: the geometry step forms the l1_scale gradient itself from these factors
            ctx.sink.scale_reg = dict(gout=gp(2), sums=s_scale, trans=trans, scale=scale, keep=(seeds, sums, sr, xz))
        elif active[2]:
            d_sc = torch.empty_like(sr)
            _lib.check(lib.vcr_scale_reg_backward(sr.shape[0], sr.data_ptr(), xz.data_ptr(), trans.data_ptr(), scale.data_ptr(),
                                                  s_scale, gp(2), d_sc.data_ptr(), st))
        if d_sc is not None and ctx.sink is not None and ctx.sink.defer_scale_grad:
            ctx.sink.scale_grad = d_sc                      # added by the activation backward of the same graph (runs later)
            d_sc = None
        return (dout, d_sc) + (None,) * 12


class _LossVals(dict):
    """Loss dictionary whose "ssim" entry (1 - SSIM index, `trainer.py:238`) is formed when it is read."""
    ssim_index = None

    def __missing__(self, key):
        if key == "ssim" and self.ssim_index is not None:
            v = 1.0 - self.ssim_index
            self[key] = v
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return (key == "ssim" and self.ssim_index is not None) or dict.__contains__(self, key)

    def items(self):
        self["ssim"]
        return dict.items(self)

    def keys(self):
        self["ssim"]
        return dict.keys(self)

    def __iter__(self):
        self["ssim"]
        return dict.__iter__(self)


def scale_grad_from_factors(f):
    """The l1_scale gradient [N,3] from the factors a `GeometrySink` holds (the path taken when no geometry step consumed them)."""
    _, _, sr, xz = f["keep"]
    d_sc = torch.empty_like(sr)
    _lib.check(_lib.load().vcr_scale_reg_backward(sr.shape[0], sr.data_ptr(), xz.data_ptr(), f["trans"].data_ptr(), f["scale"].data_ptr(),
                                                 f["sums"], f["gout"], d_sc.data_ptr(), _lib.stream_of(sr)))
    return d_sc


def fused_losses(out, model, cam, weights, it, optim_cfg, extent, mask=None):
    """-> (total, {name: value}) for the losses of `weights` that this node covers (`NAMES`)."""
    w = [float(weights.get(n, 0.0)) for n in NAMES]
    has_n = getattr(cam, "normal", None) is not None
    active = [True, True, w[2] != 0,
              w[3] != 0 and has_n and it > optim_cfg.normal_from_iter,
              w[4] != 0 and has_n and it > optim_cfg.dnormal_from_iter,
              w[5] != 0 and it > optim_cfg.consistent_normal_from_iter]
    wv = [w[0], -w[1]] + [w[k] if active[k] else 0.0 for k in range(2, 6)]
    key = tuple(wv)
    cache = fused_losses.__dict__.setdefault("_wcache", {})
    if key not in cache:
        cache[key] = torch.tensor(wv, device=out.device)
    depth_max = extent * optim_cfg.mask_depth_thr if optim_cfg.mask_depth_thr > 0 else 0.0
    total, res = _FusedLosses.apply(out, model._scaling, model._xyz, cam.original_image, getattr(cam, "normal", None), mask,
                                    cam.intr_scalars, cache[key], tuple(active), optim_cfg.exp_t, depth_max, model.trans,
                                    model.scale, getattr(model, "_geom_sink", None))
    vals = _LossVals({"l1": res[0]})
    vals.ssim_index = res[1]
    for k in range(2, 6):
        if active[k]:
            vals[NAMES[k]] = res[k]
    return total, vals
