"""Training step of the reference (`trainer.py:233-392`) on the HIP hot path, plus view-parallel
data parallelism (new capability: the reference trains on one GPU).

Per iteration (same order as `Trainer.train_step`, `trainer.py:323-392`):
  camera pick -> render() -> loss dictionary (`_compute_loss`, :233-308) -> weighted sum (:310-321)
  -> backward -> [DP: sum all-reduce of the per-Gaussian gradients over RCCL] -> densification
  statistics / densify_and_prune / opacity reset on the reference's schedule -> Adam step.

Data parallelism: one process per GPU; every rank holds a full replica (parameters, Adam moments,
densification statistics), renders ITS OWN camera of the step's batch of `world` cameras, and the
gradients are summed across ranks and scaled by 1/world inside the fused Adam kernel.  The densification
statistics accumulate per rank and are reduced where they are read (`sync_densify_stats`).  All schedule
decisions depend only on reduced quantities, so replicas stay in lock-step without broadcasts.
"""
import random

import torch
import torch.distributed as dist

from .gaussian_model import GaussianModel
from .gaussian_renderer import count_render, render, visi_acc_render, visibility_counts
from .loss_utils import (curv_loss, edge_aware_mean, entropy_regulariser, l1_ssim, normal_loss, scale_regulariser,
                         semantic_loss)


def load_capture(path):
    """`torch.load(path, weights_only=True)` of a `(GaussianModel.capture(), iteration)` file.  The reference's own files hold a
    few numpy scalars beside the tensors (`spatial_lr_scale`, learning rates from `get_expon_lr_func`), so numpy's scalar
    reconstructor and dtype classes are allow-listed -- data only, no code."""
    import numpy as np
    from torch.serialization import safe_globals
    # torch's weights-only unpickler matches a global by the "module.name" STRING in the file.  numpy >= 2 pickles its scalar
    # reconstructor as `numpy._core.multiarray.scalar`, numpy 1.x (what the reference's pytorch-2.0.1 environment has) as
    # `numpy.core.multiarray.scalar`: both spellings are allow-listed for the one function this numpy provides.
    core = getattr(np, "_core", None) or np.core
    scalar = core.multiarray.scalar
    allow = [(scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar"), np.dtype]
    allow += [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64, np.int32)]
    with safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


class Trainer:
    def __init__(self, cfg, model, cameras, extent, device, world=1, rank=0, dirs=None, seed=0, force_factorised=False,
                 overlap_sh=None, overlap_min_gaussians=400_000, exchange="allreduce", side_cus=0):
        self.cfg, self.model, self.cameras = cfg, model, cameras
        self.device, self.world, self.rank = device, world, rank
        # Collective for the geometry bucket (44 B / Gaussian): "allreduce" = ONE all-reduce, the algorithm is RCCL's choice
        # (a ring is bound by one xGMI link: 2 (n-1)/n S / B); "rs_ag" = reduce-scatter + all-gather of the same bucket,
        # which a fully connected xGMI node can run over all seven links at once (2 S / (n B)).  Same sums,
        # same replicas; which one is faster on hardware is what the first SCALE run decides.
        if exchange not in ("allreduce", "rs_ag"):
            raise ValueError("exchange must be 'allreduce' or 'rs_ag'")
        self.exchange_algo = exchange
        self.extent = extent
        self.model.extent = extent
        self.dirs = dirs
        self.weights = {k: v for k, v in cfg.optim.loss_weight.items() if v}
        self.losses = {}
        self._loss_sums = {}                  # reduction-buffer cache of the fused loss node (GeometrySink.sums)
        self.current_iteration = 0
        self.background = torch.tensor([1.0, 1.0, 1.0] if cfg.model.white_background else [0.0, 0.0, 0.0], device=device)
        self.rng = random.Random(seed)            # identical on every rank
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.prefetch_visibility_cameras = True    # the next densification's virtual cameras on a worker thread (`_visibility_cameras`)
        self._vis_ahead = None
        # random backgrounds (`trainer.py:334`) are drawn once on the host and kept on the device: a per-step H2D copy
        # of a pageable tensor is a stream synchronisation that stops the host from running ahead of the GPU
        self.bg_table = torch.rand(4096, 3, generator=self.gen).to(device)
        self.view_order = []
        self.visi_list = None
        self._stats_delta, self._visi_delta, self._stats_dirty = None, None, False     # data parallel: see _densify_stats
        self.last_stats = {}
        self._picked = []
        # DP: exchange dL/drgb (12 B/Gaussian/view, all-gather) instead of all-reducing the 192 B/Gaussian SH gradients
        self.factorised_sh = world > 1 or force_factorised       # (forcing it at world 1 exercises the path in tests)
        # Single GPU: the SH colour path runs on a second stream.  The backward leaves dL/drgb + view directions
        # (24 B/Gaussian) instead of the 192 B/Gaussian SH gradient; `vcr_sh_adam_from_rgb` applies Adam to the SH
        # coefficients on the side stream while the main stream already runs the next iteration's geometry Adam,
        # activation, projection and the latency-bound sort chain; that iteration's SH -> RGB evaluation follows on the
        # side stream and is joined before compositing.  Same arithmetic and ordering of updates as the serial loop.
        # Data parallel: the same second stream additionally waits for the all-gather of dL/drgb, so that exchange (the
        # larger of the two: 12 B/Gaussian/view) and the SH update run beside the next iteration's geometry all-reduce
        # wait, geometry Adam, projection and sort chain instead of in front of them.
        if overlap_sh is None:
            overlap_sh = not force_factorised and str(device).startswith("cuda")
        self.overlap_sh = bool(overlap_sh)
        # below ~400 k Gaussians the step is launch-bound and the second stream's events / extra launches cost more than
        # the overlap returns (100 k Gaussians at 400x300: 580 vs 810 it/s): the two-stream form is used per step, by size
        self.overlap_min_gaussians = int(overlap_min_gaussians)
        self._factorised_base = self.factorised_sh
        self.side = None
        self._pending_sh = None          # (drgb, view_dirs, sh_degree) of the last backward, not yet applied
        self._zero_campos = None
        # Fused static tail (round 3): on iterations without densify / prune / reset surgery a single process runs the adjoint
        # of the fused activation, the l1_scale gradient, the densification statistics and Adam on xyz / scaling / rotation /
        # opacity as ONE kernel (`FusedAdam.geometry_step`) instead of five.  `fuse_geometry = False` keeps the modular form (for
        # losses built on the plain getters); the three attributes are settings of the object, not of the environment -- each
        # combination the tests switch to is checked against the oracle as a whole step (tests/test_train_step_gpu.py).
        self.fuse_geometry = True
        self.fuse_raster_tail = True       # (the tail inside the rasterizer's backward)
        # ... which also evaluates the activations for the NEXT iteration's camera while the updated parameters are in registers
        self.prefetch_activation = True
        self._prefetched = None          # cameras of the next iteration, drawn ahead by `_peek_next_camera`
        # third stream: depth keys + depth sort beside the projection (two-stream form only).  Measured: neutral at 1-2 M
        # Gaussians (1.61 vs 1.61, 2.50-2.58 vs 2.49-2.58 ms/step), -4 % at 5 M (4.53 vs 4.74), where the 8 sort launches
        # over 5 M keys are long enough to matter: used from 3 M Gaussians on.
        # Binning granularity of the training render (`RasterOptions.quad_lists`), chosen per step from the previous render's
        # footprint statistic R / V (3-sigma tiles per visible Gaussian; a property of the scene that changes slowly): per
        # 8x8 quad below `quad_lists_below`, per 16x16 tile above.  Measured (profiles/r4_quad_threshold.txt, step in ms, per tile
        # -> per quad): R / V = 1.92 (5 M Gaussians, 1600 x 1200) 4.32 -> 4.04, 2.24 (2 M, 1080p, semantics) 2.635 -> 2.571, 2.64
        # (300 k, 800 x 600) 0.812 -> 0.788, 2.76 (metric scene) 1.354 -> 1.364, 10.5 (dense variant) 1.655 -> 2.10: the sort grows
        # with the quads a footprint covers (1.37x ... 4x the entries), the compositing gain does not.
        # The statistic is smoothed over ~10 renders and the switch has a hysteresis band (on below `quad_lists_below`, off above
        # it + 0.2): the cameras of one scene differ by +-0.2, and a form that flips from view to view costs more (two sets of
        # buffer sizes in the allocator, 0.7 % at the metric scene) than either form.
        self.quad_lists_below, self._tiles_per_visible, self._quad_on = 2.7, None, False
        self.sort_stream, self.sort_stream_min_gaussians = None, 3_000_000
        if self.overlap_sh:
            # `side_cus` > 0: the side stream is confined to that many compute units (spread over the XCDs), so that the streaming
            # SH update cannot occupy the CUs the sort chain of the main stream needs (experiment)
            if side_cus:
                from . import _lib
                self.side = _lib.cu_masked_stream(int(side_cus), device)
            else:
                self.side = torch.cuda.Stream(device=device)
            self.sort_stream = torch.cuda.Stream(device=device)

    def reserve_arena(self, factor=8.0, min_gb=4.0, max_fraction=0.25):
        """Take ONE large block from the device through torch's caching allocator and hand it straight back to the cache:
        later requests of sizes the cache has not seen (every densification changes N and with it the size of every
        per-Gaussian array and N-sized scratch buffer) are split off it instead of going to `hipMalloc`.  `factor` x the
        model's state (parameters + both Adam moments), at least `min_gb`, at most `max_fraction` of the free memory.
        (Measured, profiles/r4_diag_alloc.json: this removes the large `hipMalloc`s of a densification, 14 -> 0, but NOT the
        ~0.3-0.5 s the FIRST densification of a process takes -- that is the lazy loading of the code objects of the dozen
        torch kernels the selection logic uses for the first time; the second event takes 4 ms either way -- and the steady
        step is ~7 us slower out of the arena's addresses (1.354 against 1.347 ms).  Hence opt-in, not a default.)
        Returns the bytes reserved (0 on host tensors)."""
        m = self.model
        if not m._xyz.is_cuda:
            return 0
        n = m._xyz.shape[0]
        row = sum(g["params"][0].numel() // max(n, 1) for g in m.optimizer.param_groups if not g.get("aux")) * 4
        want = max(int(min_gb * 2 ** 30), int(factor * 3 * row * n))
        free, _total = torch.cuda.mem_get_info(self.device)
        want = min(want, int(max_fraction * free))
        if want <= 0:
            return 0
        block = torch.empty(want, dtype=torch.uint8, device=self.device)
        del block
        return want

    def _launch_pending_sh(self):
        """Enqueue the deferred SH Adam update on the side stream as its own kernel (`join_side`: a pending update that no
        forward is going to consume -- before evaluation renders, checkpoints, row surgery)."""
        if self._pending_sh is None:
            return
        pend, self._pending_sh = self._pending_sh, None
        if pend[0] == "views":           # data parallel: (tag, all-gather work, drgb_all, xyz snapshot, campos_all, degree)
            _, gather, drgb_all, xyz0, campos_all, deg = pend
            self._side_wait(gather, (drgb_all, xyz0, campos_all))
            self.model.optimizer.step_sh_from_rgb_views(drgb_all, xyz0, campos_all, deg, stream=self.side)
            return
        drgb, vdirs, deg = pend
        self._side_wait(None, (drgb, vdirs))
        self.model.optimizer.step_sh_from_rgb(drgb, vdirs, deg, stream=self.side)

    def _side_wait(self, work, tensors):
        """Make the SIDE stream (not the main one) wait for an async collective and keep `tensors` alive for it.
        Without a side stream (CPU / gloo tests) the wait is a plain blocking wait."""
        if self.side is None:
            if work is not None:
                work.wait()
            return
        if work is not None:
            with torch.cuda.stream(self.side):
                work.wait()
        for t in tensors:
            t.record_stream(self.side)

    def _pending_sh_update(self):
        """Provider for the rasterizer's `COLOUR_SH_UPDATE`: hands the deferred SH update to the forward call, which applies
        it on the second stream fused with the SH -> RGB evaluation (one pass over the coefficients instead of two)."""
        if self._pending_sh is None:
            return None
        pend, self._pending_sh = self._pending_sh, None
        opt = self.model.optimizer
        if pend[0] == "views":
            _, gather, drgb_all, xyz0, campos_all, deg = pend
            self._side_wait(gather, (drgb_all, xyz0, campos_all))
            return opt.make_sh_update(drgb_all, deg, xyz=xyz0, campos_all=campos_all)
        drgb, vdirs, deg = pend
        self._side_wait(None, (drgb, vdirs))
        return opt.make_sh_update(drgb, deg, view_dirs=vdirs)

    def join_side(self):
        """Apply a still-pending SH update and make the current stream wait for the side stream: call before anything
        that reads or replaces the SH coefficients outside `train_step`'s render (evaluation renders, saving, surgery)."""
        if self._pending_sh is not None:
            if self.side is not None:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
            self._launch_pending_sh()
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)

    # ---- camera batch: `world` cameras per step, rank r takes the r-th (`trainer.py:326-328`) -------
    def _draw_cameras(self):
        picked = []
        for _ in range(self.world):
            if not self.view_order:
                self.view_order = list(range(len(self.cameras)))
            picked.append(self.view_order.pop(self.rng.randint(0, len(self.view_order) - 1)))
        return picked

    def _next_cameras(self):
        picked, self._prefetched = (self._prefetched if self._prefetched is not None else self._draw_cameras()), None
        self._picked = picked
        return picked

    def _peek_next_camera(self):
        """The camera of the NEXT iteration, drawn now (same draws, same order: nothing else uses the generator between the
        static tail of an iteration without surgery and the start of the next one) so that the tail can already evaluate the
        activations for it (`VcrGeometryStep.next_*`).  -> (camera_center, R_w2c) of this rank's next camera, or None."""
        if not self.prefetch_activation:
            return None
        if self._prefetched is None:
            self._prefetched = self._draw_cameras()
        cam = self.cameras[self._prefetched[self.rank]]
        R = getattr(cam, "R_w2c", None)
        if R is None or not torch.is_tensor(cam.camera_center):
            return None
        return cam.camera_center, R

    # ---- losses (`trainer.py:233-321`) -------------------------------------------------------------------
    def active_extra_losses(self, it):
        """The losses OUTSIDE the fused loss node that contribute at iteration `it` (a weight alone does not make one active:
        the reference's `dtu` configuration carries distortion = 1000 from the start but applies it after
        `close_depth_from_iter`, `trainer.py:295-303`).  While the list is empty the step takes the fused node and the
        rasterizer skips the distortion / depth-variance channels."""
        cfg, w = self.cfg, self.weights
        act = []
        if it > cfg.optim.close_depth_from_iter:
            act += [k for k in ("distortion", "depth_var") if k in w]
        act += [k for k in ("entropy", "mono_depth") if k in w]
        if "curv" in w and "depth_normal" in w and it > cfg.optim.dnormal_from_iter and it > getattr(cfg.optim, "curv_from_iter", 0):
            act.append("curv")
        return act

    def _compute_loss(self, data, cam):
        extra = self.active_extra_losses(self.current_iteration)
        if not extra and "render_out" in data and getattr(self, "use_fused_losses", True):
            from .fused_losses import fused_losses          # one autograd node for the whole image-space loss
            total, vals = fused_losses(data["render_out"], self.model, cam, self.weights, self.current_iteration,
                                       self.cfg.optim, self.extent, mask=data.get("mask_static"))
            if "semantic" in self.weights and "sem_planes" in data:          # `trainer.py:304-307`, its own fused kernel
                vals["semantic"] = semantic_loss(data["sem_planes"], self.model.classifier, cam.mask)
                total = torch.add(total, vals["semantic"], alpha=float(self.weights["semantic"]))
            vals["total"] = total
            self.losses = vals
            return total
        cfg, it, L = self.cfg, self.current_iteration, {}
        gt_image = cam.original_image
        l1, ssim_v = l1_ssim(data["render"], gt_image)
        L["l1"], L["ssim"] = l1, 1.0 - ssim_v
        if "l1_scale" in self.weights:
            L["l1_scale"] = scale_regulariser(self.model._scaling, self.model._xyz, self.model.trans, self.model.scale)
        if "entropy" in self.weights:                                        # `trainer.py:247-249`
            L["entropy"] = entropy_regulariser(self.model._opacity, self.model._xyz, self.model.trans, self.model.scale)
        gt_normal = getattr(cam, "normal", None)
        if "mono_normal" in self.weights and it > cfg.optim.normal_from_iter:
            L["mono_normal"] = normal_loss(data["normal"], gt_normal)
        if "depth_normal" in self.weights and it > cfg.optim.dnormal_from_iter:
            L["depth_normal"] = normal_loss(data["est_normal"], gt_normal, weight_src=data["normal"].detach(),
                                            exp_t=cfg.optim.exp_t, mask=data.get("mask_static"),
                                            depth=data["depth"] if cfg.optim.mask_depth_thr > 0 else None,
                                            depth_max=self.extent * cfg.optim.mask_depth_thr)
            if "curv" in self.weights and it > getattr(cfg.optim, "curv_from_iter", 0):      # `trainer.py:282-287`
                L["curv"] = curv_loss(data["est_normal"], self._render_mask(data))
        if "consistent_normal" in self.weights and it > cfg.optim.consistent_normal_from_iter:
            L["consistent_normal"] = normal_loss(data["est_normal"], data["normal"])
        if "distortion" in self.weights and it > cfg.optim.close_depth_from_iter and "distortion" in data:
            L["distortion"] = edge_aware_mean(gt_image, data["distortion"])
        if "depth_var" in self.weights and it > cfg.optim.close_depth_from_iter and "depth_var" in data:
            L["depth_var"] = edge_aware_mean(gt_image, data["depth_var"])
        if "semantic" in self.weights and "sem_planes" in data:            # `trainer.py:304-307`, classifier + CE in one kernel
            L["semantic"] = semantic_loss(data["sem_planes"], self.model.classifier, cam.mask)
        self.losses = L
        names = [k for k in self.weights if k in L]
        wvec = self._weight_vector(names)
        total = torch.dot(torch.stack([L[k] for k in names]), wvec)      # one weighted sum instead of 2 ops per loss
        L["total"] = total
        return total

    def _render_mask(self, data):
        """`data["mask"]` of the reference's render (camera mask AND depth < extent * mask_depth_thr,
        `gaussian_renderer/__init__.py:125-131`); with `lazy_mask` renders it is formed here, on demand."""
        m = data.get("mask")
        if m is None:
            with torch.no_grad():
                m = data.get("mask_static")
                thr = self.cfg.optim.mask_depth_thr
                if thr > 0:
                    m1 = (data["depth"] < self.extent * thr).squeeze(0)
                    m = m1 if m is None else (m & m1)
                if m is None:
                    m = torch.ones(data["depth"].shape[1:], dtype=torch.bool, device=data["depth"].device)
        return m

    def _weight_vector(self, names):
        key = tuple(names)
        if getattr(self, "_wkey", None) != key:
            self._wkey = key
            self._wvec = torch.tensor([float(self.weights[k]) for k in names], device=self.device)
        return self._wvec

    # ---- gradient exchange ------------------------------------------------------------------------------------
    def _allreduce_grads(self, early_feature_step=False, defer_sh=False, rec=None, sink=None):
        """Sum the per-Gaussian gradients of all ranks (RCCL over xGMI).  One collective per parameter
        tensor, all in flight together; the 1/world scale is folded into the Adam kernel.  With the factorised SH
        exchange the all-gather of dL/drgb is awaited first, the SH gradients are rebuilt and (on iterations without
        densify / prune / opacity-reset surgery) their Adam update runs while the all-reduce of the remaining
        44 B/Gaussian is still in flight.  `rec`: the `RasterRecord` of THIS step's render (its backward left dL/drgb there).
        `sink` (round 5, the one-kernel tail under data parallelism): the bucket then carries the gradients of the four geometry
        groups in ACTIVATED space -- dL/d(mean, activated scales, unit quaternion, opacity, world-space axis column), 56 B per
        Gaussian, what the rasterizer's backward returned -- instead of the 44 B of raw-parameter gradients: the activation
        adjoint is linear in them and the parameters are replicated, so the tail can run ONCE on the sum."""
        if self.world == 1 and not self.factorised_sh:
            self.model.optimizer.grad_scale = 1.0
            return
        if self.world == 1 and not getattr(self, "force_collectives", False):   # single-rank factorised path: no collectives
            # the view directions are those the backward stored with dL/drgb, not re-derived from the positions: the static
            # tail may already have moved them (it runs inside the rasterizer's backward).  A unit vector minus a zero
            # "camera centre" is that direction again.
            drgb, dirs = rec.take_sh_factors()
            if self._zero_campos is None or self._zero_campos.device != drgb.device:
                self._zero_campos = torch.zeros(1, 3, device=drgb.device)
            self.model._features_dc.grad, self.model._features_rest.grad = \
                self._sh_grads_from_rgb(drgb[None].contiguous(), self._zero_campos, xyz=dirs.contiguous())
            self.model.optimizer.grad_scale = 1.0
            return
        works = []
        gather = None
        drgb_all = None
        timing = getattr(self, "exchange_timing", None)     # bench.py / tests: HIP events around the collectives of this step
        if timing is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.model.optimizer.grad_scale = 1.0 / self.world
        if self.factorised_sh:
            drgb = rec.take_sh_factors()[0].contiguous()
            flat = torch.empty((self.world * drgb.shape[0], 3), dtype=drgb.dtype, device=drgb.device)
            gather = dist.all_gather_into_tensor(flat, drgb, async_op=True)           # concatenated along dim 0
            drgb_all = flat.view(self.world, drgb.shape[0], 3)
        # ONE bucket for the remaining gradients (xyz, opacity, scaling, rotation, ...): a ring all-reduce pays 2(n-1) link
        # latencies per call, so four small collectives cost four times the latency of one (xGMI is point-to-point)
        geo = ("xyz", "scaling", "rotation", "opacity")
        params = [g["params"][0] for g in self.model.optimizer.param_groups
                  if not (self.factorised_sh and g["name"] in ("f_dc", "f_rest")) and not (sink is not None and g["name"] in geo)]
        grads = [p.grad for p in params]
        if sink is not None:          # activated-space gradients first: [d mean | d scales | d quaternion | d opacity | d axis column]
            act = [self.model._xyz.grad] + list(sink.grads)
            keep = [k for k, t in enumerate(act) if t is not None]
            grads = [act[k] for k in keep] + grads
        else:
            act = keep = None
        shapes = [None if t is None else t.shape for t in grads]
        numels = [p.numel() for p in params] if sink is None else [act[k].numel() for k in keep] + [p.numel() for p in params]
        sizes = [(n + 3) // 4 * 4 for n in numels]                            # 16-byte aligned segments for the Adam kernel
        ref = params[0] if params else self.model._xyz
        if all(t is not None and n == t.numel() for t, n in zip(grads, sizes)):
            flat = torch.cat([t.reshape(-1) for t in grads])                  # one kernel
        else:
            flat = torch.zeros(sum(sizes), dtype=ref.dtype, device=ref.device)
            off = 0
            for t, n in zip(grads, sizes):
                if t is not None:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += n
        if self.exchange_algo == "rs_ag" and self.world > 1:
            # reduce-scatter + all-gather on the bucket padded to a multiple of the ranks: rank r sums slice r of every
            # rank's bucket, then the summed slices are gathered back in place (both queued on the collective stream in order)
            pad = (-flat.numel()) % self.world
            if pad:
                flat = torch.cat([flat, flat.new_zeros(pad)])
            shard = torch.empty(flat.numel() // self.world, dtype=flat.dtype, device=flat.device)
            # (wait(): RCCL -- this stream waits, the host does not block; gloo's asynchronous operations are NOT ordered among
            #  themselves, so the gather must not start before the scatter has finished)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True).wait()
            works.append(dist.all_gather_into_tensor(flat, shard, async_op=True))
            self._rs_ag_keep = shard                                          # (alive until the waits below)
        else:
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        off = 0                                                                 # views of the bucket: no copy back
        if sink is not None:
            for j, k in enumerate(keep):
                act[k] = flat[off:off + numels[j]].view(shapes[j])
                off += sizes[j]
            self.model._xyz.grad, sink.grads = act[0], act[1:]
        for p, n in zip(params, sizes[(len(keep) if sink is not None else 0):]):
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += n
        if gather is not None:
            campos_all = torch.stack([self.cameras[i].camera_center for i in self._picked]).float().contiguous()
            if defer_sh:
                # nobody on this stream waits for the all-gather: the SH update (gradient formed on the fly from all views)
                # is applied on the second stream from the next forward's hook, beside its sort chain.  xyz is snapshotted
                # because the geometry update below runs first.
                self._pending_sh = ("views", gather, drgb_all, self.model._xyz.detach().clone(), campos_all,
                                    int(self.model.active_sh_degree))
            else:
                gather.wait()
                self.model._features_dc.grad, self.model._features_rest.grad = self._sh_grads_from_rgb(drgb_all, campos_all)
                if early_feature_step:
                    self.model.optimizer.step(only={"f_dc", "f_rest"})
        for w in works:
            w.wait()
        if timing is not None:
            # launch of the first collective -> the stream has waited for the bucket: the part of the exchange the step cannot hide
            # (the deferred all-gather of dL/drgb is not waited for here)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            timing["events"].append((ev0, ev1))
            timing["bucket_bytes"] = int(flat.numel() * flat.element_size())
            timing["gather_bytes"] = int(drgb_all.numel() * drgb_all.element_size()) if drgb_all is not None else 0
            timing["collectives_per_step"] = len(works) + (1 if gather is not None else 0) + (1 if self.exchange_algo == "rs_ag" and self.world > 1 else 0)
            timing["algorithm"] = self.exchange_algo
            timing["ranks"] = int(dist.get_world_size()) if dist.is_initialized() else 1

    def exchange_report(self):
        """-> what the timed exchanges of this trainer did (after `exchange_timing = {"events": []}` was set and the stream has been
        synchronised): mean / max milliseconds between the launch of a step's collectives and the point where the training stream
        has waited for the bucket, bytes per collective, collectives per step, the algorithm and how many ranks took part."""
        t = getattr(self, "exchange_timing", None)
        if not t or not t.get("events"):
            return None
        ms = [a.elapsed_time(b) for a, b in t["events"]]
        return {"exchange_ms_exposed": sum(ms) / len(ms), "exchange_ms_exposed_max": max(ms), "timed_steps": len(ms),
                "bucket_bytes": t.get("bucket_bytes"), "gather_bytes": t.get("gather_bytes"),
                "collectives_per_step": t.get("collectives_per_step"), "algorithm": t.get("algorithm"), "rccl_ranks_seen": t.get("ranks")}

    def _exchange_grads(self, overlap, surgery, rec=None, sink=None):
        """What happens between backward and the optimizer step.  Two-stream form (`overlap`, no surgery this iteration):
        single GPU -> the SH update is only stashed (applied from the next forward's colour stream); data parallel -> the
        geometry bucket is all-reduced now, dL/drgb is all-gathered asynchronously and the SH update of ALL views is left
        pending for the side stream (`defer_sh`).  Otherwise: the serial exchange."""
        m = self.model
        if overlap and not surgery:
            for g in m.optimizer.param_groups:         # moments are created (zero-filled) on THIS stream, ahead of
                if g["name"] in ("f_dc", "f_rest"):    # the projection the side stream will wait for
                    m.optimizer._state(g)
            if self.world > 1 or getattr(self, "force_collectives", False):
                self._allreduce_grads(defer_sh=True, rec=rec, sink=sink)
                self.last_exchange = "factorised-deferred"      # bucket all-reduce now, all-view SH update on the side stream
                if self.exchange_algo == "rs_ag" and self.world > 1:
                    self.last_exchange += "+rs_ag"
            else:
                self.last_exchange = "none"
                m.optimizer.grad_scale = 1.0
                self._pending_sh = rec.take_sh_factors() + (int(m.active_sh_degree),)
        else:
            self.join_side()
            self._allreduce_grads(early_feature_step=not surgery, rec=rec, sink=sink)
            collectives = self.world > 1 or getattr(self, "force_collectives", False)
            self.last_exchange = ("factorised" if self.factorised_sh else "dense") if collectives else "none"
            if collectives and self.exchange_algo == "rs_ag" and self.world > 1:
                self.last_exchange += "+rs_ag"

    def _sh_grads_from_rgb(self, drgb_all, campos_all, xyz=None):
        """sum over the step's views of basis_k(dir_view) x dL/drgb_view (HIP kernel vcr_sh_grad_from_rgb); dir_view =
        normalised (xyz - campos_view), `xyz` defaulting to the model's positions."""
        from . import _lib
        m = self.model
        lib = _lib.load()
        N = m._xyz.shape[0]
        d_dc, d_rest = torch.empty_like(m._features_dc), torch.empty_like(m._features_rest)
        _lib.check(lib.vcr_sh_grad_from_rgb(N, int(m.active_sh_degree), int(drgb_all.shape[0]),
                                            (m._xyz.detach() if xyz is None else xyz).data_ptr(),
                                            campos_all.data_ptr(), drgb_all.data_ptr(), d_dc.data_ptr(), d_rest.data_ptr(),
                                            _lib.stream_of(drgb_all)))
        return d_dc, d_rest

    def _densify_stats(self, data):
        """`add_densification_stats` + `max_radii2D` (`scene/gaussian_model.py:669-671`, `trainer.py:345`).  Data parallel: the
        rank accumulates ITS view into rank-local deltas (no collective per step); `sync_densify_stats` folds the deltas of all
        ranks into the replicated accumulators right before anything reads or re-indexes them.  Sums and maxima commute with
        the accumulation over steps, so the values every rank then holds are those of a per-step reduction."""
        m = self.model
        vp = data["viewspace_points_densify"]
        if self.world == 1 and not getattr(self, "force_collectives", False):
            m.add_densification_stats(vp, None, radii=data["radii"])
            return
        N = m._xyz.shape[0]
        if self._stats_delta is None or self._stats_delta[0].shape[0] != N:
            assert self._stats_delta is None or not self._stats_dirty, "densification statistics not synchronised before row surgery"
            self._stats_delta = (torch.zeros(N, 1, device=self.device), torch.zeros(N, 1, device=self.device))
        keep = (m.xyz_gradient_accum, m.denom)
        m.xyz_gradient_accum, m.denom = self._stats_delta
        try:
            m.add_densification_stats(vp, None, radii=data["radii"])     # (max_radii2D: running maximum, local until the sync)
        finally:
            m.xyz_gradient_accum, m.denom = keep
        if "countlist" in data and self.current_iteration > self.cfg.optim.densify_from_iter:        # `trainer.py:350-356`
            cl = data["countlist"]
            self._visi_delta = cl.clone() if self._visi_delta is None else self._visi_delta + cl
        self._stats_dirty = True

    def _stats_target(self, deltas):
        """Context: with `deltas`, the model's statistic accumulators are swapped for this rank's deltas (see `_densify_stats`)
        while a kernel that writes them runs -- the one-kernel tail under data parallelism."""
        import contextlib

        @contextlib.contextmanager
        def swap():
            m = self.model
            N = m._xyz.shape[0]
            if self._stats_delta is None or self._stats_delta[0].shape[0] != N:
                assert self._stats_delta is None or not self._stats_dirty, "densification statistics not synchronised before row surgery"
                self._stats_delta = (torch.zeros(N, 1, device=self.device), torch.zeros(N, 1, device=self.device))
            keep = (m.xyz_gradient_accum, m.denom)
            m.xyz_gradient_accum, m.denom = self._stats_delta
            try:
                yield
            finally:
                m.xyz_gradient_accum, m.denom = keep
            self._stats_dirty = True

        return swap() if deltas else contextlib.nullcontext()

    def sync_densify_stats(self):
        """Data parallel: sum the rank-local statistic deltas (and maximise `max_radii2D`) over the ranks.  `train_step`
        calls it before densify / prune / any row surgery, `Trainer.capture` before a checkpoint is taken (use that, not
        `model.capture()`, while training data-parallel); a no-op on one GPU or when nothing was accumulated."""
        if not self._stats_dirty:
            return
        m = self.model
        pair = torch.cat(self._stats_delta, 1)
        works = [dist.all_reduce(pair, op=dist.ReduceOp.SUM, async_op=True),
                 dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, async_op=True)]
        if self._visi_delta is not None:
            works.append(dist.all_reduce(self._visi_delta, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        m.xyz_gradient_accum += pair[:, :1]
        m.denom += pair[:, 1:]
        if self._visi_delta is not None:
            self.visi_list = self._visi_delta if self.visi_list is None else self.visi_list + self._visi_delta
        self._stats_delta, self._visi_delta, self._stats_dirty = None, None, False

    # ---- checkpoints (`trainer.py:170-172,425-430`): `torch.save((model.capture(), iteration), "chkpntN.pth")` ----------
    def capture(self):
        """-> `(model.capture(), current_iteration)`, the tuple the reference saves.  First applies a still-pending SH
        update (two-stream form) and folds the rank-local densification statistics into the replicated accumulators, so
        that a run resumed from the file densifies exactly like the uninterrupted one.  Data parallel: a COLLECTIVE --
        every rank must call it at the same iteration (all ranks then hold the same tuple; rank 0 writes the file)."""
        self.join_side()
        self.sync_densify_stats()
        return (self.model.capture(), self.current_iteration)

    def save_checkpoint(self, path):
        state = self.capture()
        if self.rank == 0:
            torch.save(state, path)

    def load_checkpoint(self, path, trust_pickle=False):
        """Resume from a `chkpntN.pth` written by `save_checkpoint` or by the reference's trainer.  The captured tuple holds
        tensors, numbers and dicts only, so the file is read with `weights_only=True`; `trust_pickle=True` falls back to the
        reference's unrestricted `torch.load` (`trainer.py:170`), which EXECUTES code from the file -- only for files you wrote."""
        self.join_side()
        try:
            model_params, first_iter = load_capture(path)
        except Exception:
            if not trust_pickle:
                raise
            model_params, first_iter = torch.load(path, map_location="cpu", weights_only=False)
        self.model.restore(model_params, self.cfg.optim, device=self.device)
        self.model.extent = self.extent
        self.current_iteration = int(first_iter)
        self._stats_delta, self._visi_delta, self._stats_dirty = None, None, False
        self._pending_sh, self.visi_list = None, None
        # the camera drawn ahead for the activation prefetch belongs to the run that was interrupted by this load
        self._prefetched = None
        self.model._act_cache = None
        ahead, self._vis_ahead = getattr(self, "_vis_ahead", None), None
        if ahead is not None:
            ahead["thread"].join()       # (its cameras belonged to the interrupted run; the generator is left where that run put it)

    # ---- visibility / importance passes (`tools/prune.py:6-69`, `trainer.py:688-702`), camera-sharded ------------
    @torch.no_grad()
    def visibility_mask(self, cams, batched=True):
        """`Trainer.get_visi_mask_acc` (`trainer.py:688-702`): Gaussians that contribute to a pixel of any of `cams` and lie
        inside the normalised bounding box.  The reference renders the cameras one by one (`get_visi_list`) and tests the
        summed `countlist` for `> 0`; here a rank's share of the cameras goes through ONE batched library call that only
        raises a flag per Gaussian (`batched=False`: the reference's per-camera form, kept for the equality test)."""
        mine = cams[self.rank::self.world]
        if batched and self.model._xyz.is_cuda:
            count = visibility_counts(mine, self.model, self.cfg.pipline, flags_only=True)
        else:
            count = None
            for cam in mine:
                c = visi_acc_render(cam, self.model, self.cfg.pipline, self.background)["countlist"]
                count = c if count is None else count + c
            if count is None:
                count = torch.zeros(self.model._xyz.shape[0], dtype=torch.int32, device=self.device)
        if self.world > 1:
            dist.all_reduce(count, op=dist.ReduceOp.SUM)
        return (count > 0) & self.model.get_inside_gaus_normalized()[0]

    @torch.no_grad()
    def importance_scores(self, cams):
        cnt, imp = None, None
        for cam in cams[self.rank::self.world]:
            pkg = count_render(cam, self.model, self.cfg.pipline, self.background)
            cnt = pkg["gaussians_count"] if cnt is None else cnt + pkg["gaussians_count"]
            imp = pkg["important_score"] if imp is None else imp + pkg["important_score"]
        if self.world > 1:
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            dist.all_reduce(imp, op=dist.ReduceOp.SUM)
        return cnt, imp

    def _visibility_cameras(self, sc):
        """`Trainer.get_visi_mask_acc` (`trainer.py:688-702`): virtual bounding-box cameras when
        `sample_cams.random`, else `num` training cameras drawn with replacement.  Seeded identically on every rank.
        Round 6: the host half of the random cameras (placement, 200 matrix inverses: ~8 ms) is formed AHEAD on a worker thread
        -- right after the previous request, from the same generator in the same order, so the cameras are the ones an
        on-demand call would have produced -- and only the four host-to-device copies remain in the densification step."""
        if sc.random:
            from .camera_utils import sample_cameras_host
            from .cameras import SampleCam
            key = (int(sc.num), bool(getattr(sc, "up", False)), bool(getattr(sc, "around", True)))
            trans, scale = self.model.trans.detach().float().cpu().clone(), self.model.scale.detach().float().cpu().clone()

            def host():
                return sample_cameras_host(key[0], trans, scale, up=key[1], around=key[2], generator=self.gen)

            pending, self._vis_ahead = getattr(self, "_vis_ahead", None), None
            bundle = None
            if pending is not None:
                pending["thread"].join()
                if pending["key"] == key and torch.equal(pending["trans"], trans) and torch.equal(pending["scale"], scale) and \
                        "error" not in pending:
                    bundle = pending["bundle"]
            if bundle is None:
                bundle = host()
            cams = SampleCam.batch_from_host(bundle, self.device)
            if self.prefetch_visibility_cameras:
                import threading
                nxt = {"key": key, "trans": trans, "scale": scale}

                def work():
                    try:
                        nxt["bundle"] = host()
                    except Exception as e:          # (the consumer falls back to the on-demand path)
                        nxt["error"] = e

                nxt["thread"] = threading.Thread(target=work, daemon=True)
                nxt["thread"].start()
                self._vis_ahead = nxt
            return cams
        return [self.cameras[self.rng.randrange(len(self.cameras))] for _ in range(sc.num)]

    def v_imp_score(self, imp, v_pow):
        """`tools/prune.py:6-22` (fixture g12)."""
        from .prune import calculate_v_imp_score
        return calculate_v_imp_score(self.model, imp, v_pow)

    # ---- one iteration ------------------------------------------------------------------------------------------
    def train_step(self):
        cfg, m = self.cfg, self.model
        self.current_iteration += 1
        it = self.current_iteration
        m.update_learning_rate(it)
        if it % 1000 == 0:
            m.oneupSHdegree()
        cam = self.cameras[self._next_cameras()[self.rank]]
        bg = self.bg_table[it % self.bg_table.shape[0]] if cfg.optim.random_background else self.background
        extra = self.active_extra_losses(it)
        fused = getattr(self, "use_fused_losses", True) and not extra
        from .rasterizer import RasterOptions
        overlap = self.overlap_sh and m._xyz.shape[0] >= self.overlap_min_gaussians
        self.factorised_sh = self._factorised_base or overlap
        if not overlap and self._pending_sh is not None:
            self.join_side()
        surgery = (it < cfg.optim.densify_until_iter and it > cfg.optim.densify_from_iter
                   and it % cfg.optim.densification_interval == 0) \
            or it % cfg.optim.opacity_reset_interval == 0 or it in cfg.optim.prune.iterations \
            or (cfg.model.white_background and it == cfg.optim.densify_from_iter)
        from . import fused_losses, gaussian_model
        # the fused static tail needs every gradient path into scaling / rotation / opacity to pass through the fused
        # activation node (the fused loss node guarantees that), un-reduced gradients (one process) and unchanged rows
        dp = self.world > 1 or getattr(self, "force_collectives", False)
        armed = bool(self.fuse_geometry and fused and not surgery and m._xyz.is_cuda and m._xyz.shape[0] > 0
                     and not cfg.pipline.compute_cov3D_python)
        # the iteration's side channel (see GeometrySink): armed = the one-kernel static tail; the l1_scale gradient joins the
        # activation backward's kernel; the loss node's reduction buffer is this trainer's
        sink = gaussian_model.GeometrySink(armed=armed, defer_scale_grad=True, sums=self._loss_sums)
        # Data parallel (round 5): the same one-kernel tail, on the SUM over the ranks -- the rasterizer's backward returns the
        # activated-space gradients (the normal's in the world-space form, `RasterOptions.world_normals`), the exchange sums
        # those 56 B per Gaussian, and `geometry_step` applies adjoint + l1_scale (once) + Adam with grad_scale = 1 / world.
        sink.exchange = armed and dp
        # ... and, on ONE process with the factorised SH gradient, the tail runs INSIDE the rasterizer's backward (projection
        # backward + activation adjoint + statistics + Adam in one kernel): the geometry gradients never reach memory
        raster_tail = armed and not dp and self.fuse_raster_tail and self.factorised_sh
        def next_cam():            # (evaluated when the tail is prepared: the sink then knows this render's `want_normal`)
            nc = self._peek_next_camera() if sink.want_normal is not None else None
            return None if nc is None else nc + (sink.want_normal,)

        if raster_tail:
            stats_on = it < cfg.optim.densify_until_iter
            sink.tail = lambda: m.optimizer.prepare_geometry_step(m, sink, in_registers=True, stats=stats_on, next_cam=next_cam())
        # two-stream form: the deferred SH update is applied on the second stream in ONE pass with the SH -> RGB evaluation
        opts = RasterOptions("rgb" if self.factorised_sh else "full", self.side if overlap else None, None,
                             self._pending_sh_update if overlap else None,
                             self.sort_stream if (overlap and m._xyz.shape[0] >= self.sort_stream_min_gaussians) else None,
                             quad_lists=self._quad_on, tail=sink if raster_tail else None,
                             world_normals=sink if sink.exchange else None)
        m._geom_sink = sink
        ok, left = False, None
        try:
            data = render(cam, m, cfg, bg, dirs=self.dirs, lazy_mask=True, geometry=not fused, raster_options=opts,
                          dist_channels="distortion" in extra or "depth_var" in extra)
            if self._pending_sh is not None:     # the render did not go through the two-stream path (e.g. no Gaussians)
                self.join_side()
            rr = data["raster"]
            if rr.V > 0:
                t = rr.R / rr.V
                self._tiles_per_visible = t if self._tiles_per_visible is None else 0.9 * self._tiles_per_visible + 0.1 * t
                if self._tiles_per_visible < self.quad_lists_below:
                    self._quad_on = True
                elif self._tiles_per_visible > self.quad_lists_below + 0.2:
                    self._quad_on = False
            loss = self._compute_loss(data, cam)
            loss.backward(fused_losses.unit_seed(loss.device))
            ok = True
        finally:
            if not ok and sink.started:
                self.last_tail = "raster-partial"               # geometry stepped inside the failed backward, SH not: do not retry
            m._geom_sink = None
            sink.armed = False
            sink.tail = None                                     # (its closure refers to the sink: no cycle left behind)
            left, sink.scale_grad = sink.scale_grad, None        # (never survives the step, also on errors)
        if not sink.done and sink.grads is None and sink.scale_reg is not None:
            left = fused_losses.scale_grad_from_factors(sink.scale_reg)        # (no activation backward ran: ordinary path)
        if ok and left is not None:                 # (no activation backward consumed it: add it the ordinary way)
            m._scaling.grad = left if m._scaling.grad is None else m._scaling.grad + left
        geom = armed and (sink.done or sink.grads is not None)
        self.last_tail = "raster" if sink.done else ("kernel" if geom else "modular")     # (which form of the tail ran)
        with torch.no_grad():
            self._exchange_grads(overlap, surgery, data["raster"], sink=sink if (geom and sink.exchange) else None)
            if geom:
                # activation adjoint + l1_scale gradient + densification statistics + Adam on the four geometry groups
                stats = it < cfg.optim.densify_until_iter
                if not sink.done:                  # (done: the rasterizer's backward has applied the tail itself)
                    vp = data["viewspace_points_densify"]
                    with self._stats_target(dp and stats):           # (data parallel: statistics into the rank-local deltas)
                        m.optimizer.geometry_step(m, sink, grad2d=vp.grad.contiguous() if (stats and vp.grad is not None) else None,
                                                  radii=data["radii"] if stats else None, next_cam=next_cam(),
                                                  normals_world=sink.exchange)
                # the one-kernel tail has applied Adam to these groups; a gradient that reached them by another path (a loss
                # built on the plain getters) would be applied a SECOND time by optimizer.step() below
                stray = [k for k in ("_scaling", "_rotation", "_opacity") if getattr(m, k).grad is not None]
                if stray:
                    raise RuntimeError(f"fused geometry tail: {stray} received a gradient outside the fused activation node; "
                                       "set Trainer.fuse_geometry = False for losses on the plain getters")
                if stats and it > cfg.optim.densify_from_iter and "countlist" in data:                  # `trainer.py:350-356`
                    cl = data["countlist"]
                    if dp:
                        self._visi_delta = cl.clone() if self._visi_delta is None else self._visi_delta + cl
                    else:
                        self.visi_list = cl if self.visi_list is None else self.visi_list + cl
            if it < cfg.optim.densify_until_iter and not geom:
                self._densify_stats(data)
                if it > cfg.optim.densify_from_iter and "countlist" in data and not self._stats_dirty:        # `trainer.py:350-356`
                    cl = data["countlist"]                # (data parallel: `_densify_stats` accumulated it with the other deltas)
                    self.visi_list = cl if self.visi_list is None else self.visi_list + cl
            if surgery:
                self.sync_densify_stats()                 # (data parallel; before anything reads or re-indexes the statistics)
            if it < cfg.optim.densify_until_iter:
                if it > cfg.optim.densify_from_iter and it % cfg.optim.densification_interval == 0:
                    size_threshold = 20 if it > cfg.optim.opacity_reset_interval else None
                    visi = None
                    dl = cfg.optim.densify_large
                    if dl.percent_dense and dl.sample_cams.num > 0:
                        visi = self.visibility_mask(self._visibility_cameras(dl.sample_cams))
                        if self.visi_list is not None:
                            # `trainer.py:366` writes `visi & self.visi_list > 0`; Python parses that as
                            # `(visi & self.visi_list) > 0` (bool & int tensor promotes to int), i.e. both conditions
                            visi = (visi & self.visi_list) > 0
                    m.densify_and_prune(cfg.optim.densify_grad_threshold, 0.005, self.extent, size_threshold, visi)
                    self.visi_list = None
                if it % cfg.optim.opacity_reset_interval == 0 or (cfg.model.white_background and it == cfg.optim.densify_from_iter):
                    m.reset_opacity()
            if it in cfg.optim.prune.iterations:
                _, imp = self.importance_scores(self.cameras)
                i = cfg.optim.prune.iterations.index(it)
                m.prune_gaussians((cfg.optim.prune.decay ** i) * cfg.optim.prune.percent,
                                  self.v_imp_score(imp, cfg.optim.prune.v_pow))
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
        return data


def make_synthetic_trainer(raw, cams, device, world=1, rank=0, preset="tnt", gt_jitter=0.02, seed=0,
                           force_factorised=False, overlap_sh=None, overlap_min_gaussians=400_000, exchange="allreduce",
                           side_cus=0, **overrides):
    """Model + GT (renders of a perturbed copy of the scene, so every loss is non-trivial) + Trainer."""
    from . import synthetic
    from .config import make_config
    from .graphics_utils import get_all_px_dir
    cfg = make_config(preset, **overrides)
    sem = "obj_dc" in raw
    cfg.model.enable_semantic = sem
    if not sem:
        cfg.optim.loss_weight.semantic = 0.0          # (the tnt preset trains semantics; a scene without object ids cannot)
    if sem:
        cfg.model.ch_sem_feat = raw["obj_dc"].shape[-1]
        cfg.model.num_cls = 2
    model = GaussianModel(cfg.model)
    model.create_from_params(raw, spatial_lr_scale=synthetic.cameras_extent(cams), device=device)
    model.active_sh_degree = cfg.model.sh_degree
    model.training_setup(cfg.optim)
    extent = synthetic.cameras_extent(cams)
    dirs = get_all_px_dir(cams[0].intr, cams[0].image_height, cams[0].image_width) \
        if cfg.model.depth_type == "intersection" else None
    tr = Trainer(cfg, model, cams, extent, device, world=world, rank=rank, dirs=dirs, seed=seed,
                 force_factorised=force_factorised, overlap_sh=overlap_sh, overlap_min_gaussians=overlap_min_gaussians,
                 exchange=exchange, side_cus=side_cus)
    # ground truth from a jittered copy
    g = torch.Generator().manual_seed(seed + 1)
    raw2 = {k: v.clone() for k, v in raw.items()}
    raw2["xyz"] = raw2["xyz"] + gt_jitter * 0.1 * torch.randn(raw["xyz"].shape, generator=g)
    raw2["f_dc"] = raw2["f_dc"] + gt_jitter * 5 * torch.randn(raw["f_dc"].shape, generator=g)
    gt_model = GaussianModel(cfg.model)
    gt_model.create_from_params(raw2, spatial_lr_scale=1.0, device=device)
    gt_model.active_sh_degree = cfg.model.sh_degree
    gt_model.extent = extent
    bg = torch.zeros(3, device=device)
    with torch.no_grad():
        for cam in cams:
            pkg = render(cam, gt_model, cfg, bg, dirs=dirs)
            cam.original_image = pkg["render"].clamp(0, 1).contiguous()
            cam.normal = pkg["est_normal"].contiguous()
            if sem:
                cam.mask = (pkg["depth"][0] > 0).long()
    del gt_model
    return tr


class BenchTrainer:
    """bench.py's step: exactly `Trainer.train_step` on a synthetic workload."""

    def __init__(self, raw, cams, device, world=1, rank=0, preset="tnt", exchange="allreduce", side_cus=0, arena=False):
        self.tr = make_synthetic_trainer(raw, cams, device, world=world, rank=rank, preset=preset, exchange=exchange,
                                         side_cus=side_cus,
                                         optim={"densify_from_iter": 10 ** 9, "prune": {"iterations": []}})
        self.last_R = self.last_V = self.last_E = 0
        self._primed = False
        self.arena_bytes = self.tr.reserve_arena() if arena else 0

    def prime(self, min_seconds=1.0):
        """Untimed set-up.  Runs ordinary training steps -- every camera at least once, so that every instance-count-
        dependent buffer size has been seen by the caching allocator (a first-seen size inside the timed window is a
        hipMalloc stall of several ms), and for at least `min_seconds`, because on a fresh box the first second of a
        process is slowed by code / library page-in on the host, which starves the GPU -- and then puts the model, the
        optimizer and the trainer back into their initial state, so the timed steps run on the workload as specified."""
        if self._primed:
            return
        self._primed = True
        import copy
        import time
        tr, m = self.tr, self.tr.model
        names = ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_objects_dc",
                 "xyz_gradient_accum", "denom", "max_radii2D"]
        saved = {k: getattr(m, k).detach().clone() for k in names if isinstance(getattr(m, k, None), torch.Tensor)}
        host = (tr.current_iteration, tr.rng.getstate(), list(tr.view_order), copy.deepcopy(
            [(g["name"], g["lr"]) for g in m.optimizer.param_groups]), m.active_sh_degree)
        t0, i = time.perf_counter(), 0
        # (multi-GPU: a FIXED count, identical on every rank -- a time-based count would desynchronise the collectives)
        fixed = len(tr.cameras) if tr.world == 1 else max(len(tr.cameras), 64)
        while i < fixed or (tr.world == 1 and time.perf_counter() - t0 < min_seconds):
            self.step(-1 - i)
            i += 1
        tr.join_side()
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in saved.items():
                getattr(m, k).data.copy_(v)
        m.optimizer.state = {}
        m.optimizer.zero_grad(set_to_none=True)
        tr._stats_delta, tr._visi_delta, tr._stats_dirty = None, None, False
        tr.current_iteration, rng_state, tr.view_order, lrs, m.active_sh_degree = host
        tr.rng.setstate(rng_state)
        tr._prefetched, m._act_cache = None, None           # (drawn / evaluated for the priming run's next step)
        for g, (name, lr) in zip(m.optimizer.param_groups, lrs):
            g["lr"] = lr
        torch.cuda.synchronize()

    def step(self, i):
        rec = self.tr.train_step()["raster"]    # (no per-rank fallback: a rank that switched exchange algorithm alone would hang RCCL)
        self.last_R, self.last_V, self.last_E = rec.R, rec.V, rec.emitted

    @torch.no_grad()
    def scene_shape(self):
        """Untimed diagnostics of the workload: longest per-tile list and number of covered pixels (alpha > 0) of the last
        camera, from one extra render with `debug` set (that render synchronises)."""
        tr = self.tr
        tr.join_side()
        cam = tr.cameras[tr._picked[tr.rank]] if tr._picked else tr.cameras[0]
        dbg = tr.cfg.pipline.debug
        tr.cfg.pipline.debug = True
        try:
            pkg = render(cam, tr.model, tr.cfg, tr.background, dirs=tr.dirs)
        finally:
            tr.cfg.pipline.debug = dbg
        rec = pkg["raster"]
        return {"max_tile_len": int(rec.max_tile_len), "covered_pixels": int((pkg["alpha"] > 0).sum()), "emitted": int(rec.emitted),
                "visible": int(rec.V)}

    @torch.no_grad()
    def dense_variant_roofline(self, scale_mult, peak_gbs, sem=0, reps=10):
        """Untimed context for the headline roofline figure: the SAME compositing kernel on the same scene with every scale
        multiplied by `scale_mult` (R/N ~ 10 instead of ~ 3: what trained scenes look like).  -> achieved GB/s on
        algorithmic bytes / fraction, from `reps` forward renders; the model is restored afterwards."""
        import math
        from . import _lib
        tr, m = self.tr, self.tr.model
        tr.join_side()
        cam = tr.cameras[0]
        m._scaling.data += math.log(scale_mult)
        try:
            for i in range(reps + 2):
                if i == 2:
                    torch.cuda.synchronize()
                    _lib.profile_enable(True, stages=["composite_fwd"])
                    _lib.profile_read()
                pkg = render(cam, m, tr.cfg, tr.background, dirs=tr.dirs)
            torch.cuda.synchronize()
            ms, cnt = _lib.profile_read()["composite_fwd"]
        finally:
            _lib.profile_enable(False)
            m._scaling.data -= math.log(scale_mult)
        R, P = pkg["raster"].R, cam.image_height * cam.image_width
        alg = (60 + 4 * sem) * R + (4 * (8 + sem) + 20) * P
        avg = ms / max(cnt, 1)
        ach = alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        return {"scale_mult": scale_mult, "tile_instances_R": R, "avg_ms": avg, "algorithmic_bytes": alg, "achieved": ach,
                "frac": ach / peak_gbs}

    def exchange(self):
        """Which gradient exchange the steps use: none (1 GPU) | factorised (all-gather dL/drgb + bucket all-reduce) | dense."""
        return getattr(self.tr, "last_exchange", "none")

    def describe(self):
        w = self.tr.weights
        return "render fwd (activate, raster, normals) + losses[" + ",".join(sorted(w)) + "] + bwd + " + \
            (("RCCL exchange (all-gather dL/drgb + all-reduce 44 B/Gaussian) + " if self.tr.factorised_sh
              else "RCCL grad all-reduce + ") if self.tr.world > 1 else "") + \
            ("fused Adam, SH coefficients updated on a second stream beside the next step's sort chain"
             if (self.tr.overlap_sh and self.tr.model._xyz.shape[0] >= self.tr.overlap_min_gaussians) else "fused Adam") + " (densify/prune off in the timed window)"
