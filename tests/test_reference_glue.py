"""g13: the operator glue around the rasterizer against the REFERENCE'S OWN code.

`tests/golden/g13_reference_glue.npz` was produced (tests/golden/make_golden_glue.py, build container only) by running the
reference's `render()` (`gaussian_renderer/__init__.py:22-164`) and `Trainer._compute_loss` / `_get_total_loss`
(`trainer.py:233-321`) with the oracle rasterizer standing in for the absent CUDA extension.  It holds, per case: the raw
parameters, camera, ground truth; what the glue handed to the rasterizer; the rasterizer's output; the loss dictionary, the
total, d total / d rendered_out; the gradients the rasterizer returned for its inputs and the raw-parameter gradients.

CPU: `oracle/trainer_torch.py` (the restatement every whole-step test relies on) must reproduce all of it.
GPU: the product's `render()` + `Trainer._compute_loss` (fused loss node and modular path) run with a stand-in rasterizer
that replays the fixture's output and input gradients, so that ONLY the glue -- fused activation / normal kernel, channel
split, masks, normalisation, depth-to-normal, every loss kernel, weighted total, and their adjoints -- is compared with the
reference's own numbers, independent of any rasterizer."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import trainer_torch as OT

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "g13_reference_glue.npz"), allow_pickle=False)
CASES = [str(c) for c in Z["cases"]]
RAW = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "obj_dc"]


class Case:
    def __init__(self, pre):
        self.pre = pre
        self.tag = pre.split("_")[0]
        self.it, self.sh_degree, self.num_dist, self.sem_on, self.has_mask = (int(v) for v in self["meta"])
        self.weights = {k: float(v) for k, v in self["weights"]}
        self.H, self.W = (int(v) for v in self["hw"])

    def __getitem__(self, k):
        return Z[f"{self.pre}__{k}"]

    def __contains__(self, k):
        return f"{self.pre}__{k}" in Z.files

    def t(self, k, device="cpu", dtype=None):
        v = torch.from_numpy(np.asarray(self[k]))
        return v.to(device=device, dtype=dtype) if dtype is not None else v.to(device)

    def losses(self):
        return {k.split("__loss_")[1]: float(Z[k]) for k in Z.files if k.startswith(self.pre + "__loss_")}

    def config(self):
        from vcr_gaus_amd.config import make_config
        cfg = make_config(self.tag)
        lw = cfg.optim.loss_weight
        for k in list(lw.keys()):
            lw[k] = 0.0
        for k, v in self.weights.items():
            lw[k] = v
        cfg.model.enable_semantic = bool(self.sem_on)
        cfg.model.ch_sem_feat, cfg.model.num_cls = 2, 2
        return cfg

    def raw(self, device="cpu"):
        r = {k: self.t("in_" + k, device) for k in RAW}
        if not self.sem_on:
            r.pop("obj_dc")
        return r

    def camera(self, device="cpu"):
        from vcr_gaus_amd.cameras import Camera
        fx, fy = (float(v) for v in self["cam_fov"])
        return Camera(0, self["cam_R"], self["cam_T"], fx, fy, image=self.t("gt_image"), normal=self.t("gt_normal"),
                      mask=self.t("labels") if self.has_mask else None, device=device)


def close(a, b, rel, abs_=0.0):
    return abs(float(a) - float(b)) <= rel * abs(float(b)) + abs_


def field_err(got, ref):
    """(max-norm relative error, 99.9th percentile of |d| / (|ref| + 1e-3 rms))."""
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    d = (got - ref).abs()
    mx = float(d.max() / ref.abs().max().clamp_min(1e-30))
    rms = float(ref.square().mean().sqrt())
    q = d / (ref.abs() + 1e-3 * rms + 1e-30)
    k = max(1, int(math.ceil(0.999 * q.numel())))
    return mx, float(q.kthvalue(k).values)


def test_presets_carry_the_reference_weights():
    """The loss weights the reference's Config resolved for the three datasets = this repo's presets (cases without overrides)."""
    from vcr_gaus_amd.config import make_config
    for pre in CASES:
        c = Case(pre)
        if len(pre.split("_")) > 2:          # a case with overridden weights
            continue
        w = {k: float(v) for k, v in make_config(c.tag).optim.loss_weight.items() if v}
        assert w == c.weights, (pre, w, c.weights)


@pytest.mark.parametrize("pre", CASES)
def test_oracle_iteration_reproduces_the_reference_glue(pre):
    """oracle/trainer_torch.py in fp64 (as the fixture was produced) against the reference's render() + loss assembly:
    rasterizer output, masks, normals, loss dictionary, total, d total / d rendered_out, raw-parameter gradients.  Agreement
    is limited by two places where the reference's code pins float32 whatever the working precision: the pixel grid of
    `depth2point_cam` (`tools/graphics_utils.py:123-124`: est_normal differs by ~1e-6) and the SSIM window
    (`tools/loss_utils.py:49-57`: ssim differs by ~5e-7 relative); the bounds are ~10x what those two produce."""
    c = Case(pre)
    cfg, cam = c.config(), c.camera()
    leaf = {k: v.double().clone().requires_grad_(True) for k, v in c.raw().items()}
    data = OT.render(leaf, cam, cfg, float(c["extent"]), c.t("bg"), c.t("dirs"), c.sh_degree, num_dist=c.num_dist)
    out = data["out"]
    out.retain_grad()
    if c.num_dist == 2:
        data["depth_var"] = out[-1:] / data["alpha"] - (out[-2:-1] / data["alpha"]) ** 2
    if c.num_dist == 1:
        data["distortion"] = out[-1:]
    cls = (c.t("cls_w"), c.t("cls_b")) if c.sem_on else None
    L, total = OT.losses(data, leaf, cam, cfg, c.it, torch.zeros(3), torch.ones(3), classifier=cls)
    total.backward()
    ref_out = c.t("rendered_out")
    assert out.shape == ref_out.shape
    assert torch.equal(data["radii"], c.t("radii"))
    mx, _ = field_err(out, ref_out)
    assert mx < 1e-8, mx                                        # same rasterizer, same activations / normals going in
    assert torch.equal(data["mask"], c.t("mask"))
    assert field_err(data["normal"], c.t("normal"))[0] < 1e-9 and field_err(data["est_normal"], c.t("est_normal"))[0] < 1e-5
    ref_l = c.losses()
    assert set(ref_l) - {"total"} == set(L), (sorted(ref_l), sorted(L))
    for k, v in L.items():
        assert close(v, ref_l[k], 5e-6, 1e-12), (k, float(v), ref_l[k])
    assert close(total, c["total"], 5e-6), (float(total), float(c["total"]))
    mx, p999 = field_err(out.grad, c.t("d_rendered_out"))
    assert mx < 2e-5 and p999 < 1e-4, (mx, p999)
    for k in RAW:
        if f"grad_{k}" in c and k in leaf:
            mx, p999 = field_err(leaf[k].grad, c.t(f"grad_{k}"))
            assert mx < 5e-5 and p999 < 2e-3, (k, mx, p999)


# ---------------------------------------------------------------------------------------------------------------------------
class _Replay(torch.autograd.Function):
    """Stand-in rasterizer for the GPU glue test: returns the fixture's output, records what it was called with and what
    gradient reached it, and hands back the fixture's input gradients."""

    @staticmethod
    def forward(ctx, box, means3D, means2D, means2D_densify, shs, shs_rest, normals, sem, opac, scales, rots):
        ctx.box = box
        box["seen"] = dict(means3D=means3D, shs=shs, shs_rest=shs_rest, normals_precomp=normals, semantics_precomp=sem,
                           opacities=opac, scales=scales, rotations=rots)
        box["seen"] = {k: (None if v is None else v.detach().clone()) for k, v in box["seen"].items()}
        ctx.mark_non_differentiable(box["radii"])
        return box["out"].clone(), box["radii"]

    @staticmethod
    def backward(ctx, g, _r=None):
        b = ctx.box
        b["dout"] = g.detach().clone()
        d = b["dargs"]
        return (None, d["means3D"], d["means2D"], d["means2D_densify"], d["shs"], d["shs_rest"], d["normals_precomp"],
                d.get("semantics_precomp"), d["opacities"], d["scales"], d["rotations"])


def _replay_rasterizer(box):
    from vcr_gaus_amd.rasterizer import RasterRecord

    class Replay(torch.nn.Module):
        def __init__(self, raster_settings, num_dist=None, options=None):
            super().__init__()
            box["settings"], box["num_dist"] = raster_settings, num_dist
            self.record = RasterRecord()

        def forward(self, means3D, means2D, opacities, means2D_densify=None, shs=None, colors_precomp=None, normals_precomp=None,
                    semantics_precomp=None, scales=None, rotations=None, cov3D_precomp=None, dirs=None, inside=None, shs_rest=None):
            assert colors_precomp is None and cov3D_precomp is None and inside is None
            box["dirs"] = dirs
            return _Replay.apply(box, means3D, means2D, means2D_densify, shs, shs_rest, normals_precomp, semantics_precomp, opacities,
                                 scales, rotations)

    return Replay


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("pre", CASES)
def test_product_glue_matches_the_reference(device, monkeypatch, pre, fused):
    from vcr_gaus_amd import gaussian_renderer as GR
    from vcr_gaus_amd.gaussian_model import GaussianModel
    from vcr_gaus_amd.trainer import Trainer
    c = Case(pre)
    cfg, cam = c.config(), c.camera(device)
    model = GaussianModel(cfg.model)
    model.create_from_params(c.raw(), spatial_lr_scale=1.0, device=device)
    model.trans, model.scale = torch.zeros(3, device=device), torch.ones(3, device=device)
    model.active_sh_degree = c.sh_degree
    if c.sem_on:
        with torch.no_grad():
            model.classifier.weight.copy_(c.t("cls_w", device).float())
            model.classifier.bias.copy_(c.t("cls_b", device).float())
    model.training_setup(cfg.optim)
    dirs = c.t("dirs", device).float()
    tr = Trainer(cfg, model, [cam], float(c["extent"]), device, dirs=dirs, overlap_sh=False)
    tr.current_iteration = c.it
    tr.use_fused_losses = fused
    extra = tr.active_extra_losses(c.it)
    takes_fused = fused and not extra
    def g(k):                                      # the fixture is fp64; the product computes in fp32
        v = c.t(k, device)
        return v.float() if v.is_floating_point() else v
    dsh = torch.cat([g("grad_f_dc"), g("grad_f_rest")], 1)            # d/d shs = the raw SH gradients (get_features is a cat)
    box = dict(out=g("rendered_out"), radii=g("radii"),
               dargs=dict(means3D=g("darg_means3D"), means2D=g("darg_means2D"), means2D_densify=g("darg_means2D_densify"),
                          shs=dsh[:, :1].contiguous(), shs_rest=dsh[:, 1:].contiguous(), normals_precomp=g("darg_normals_precomp"),
                          opacities=g("darg_opacities"), scales=g("darg_scales"), rotations=g("darg_rotations")))
    if c.sem_on:
        box["dargs"]["semantics_precomp"] = g("darg_semantics_precomp")
    monkeypatch.setattr(GR, "GaussianRasterizer", _replay_rasterizer(box))
    bg = g("bg")
    # -- the call train_step makes (lazy mask, geometry inside the fused node when it is taken) -----------------------------
    data = GR.render(cam, model, cfg, bg, dirs=dirs, lazy_mask=True, geometry=not takes_fused,
                     dist_channels="distortion" in extra or "depth_var" in extra)
    # what reached the rasterizer = what the reference's glue handed to its extension
    assert bool(c["shs_is_cat"])
    seen = box["seen"]
    for k in ["means3D", "opacities", "scales", "rotations", "normals_precomp"] + (["semantics_precomp"] if c.sem_on else []):
        mx, _ = field_err(seen[k], g("arg_" + k))
        assert mx < 3e-6, (k, mx)
    assert torch.equal(seen["shs"], model._features_dc.detach()) and torch.equal(seen["shs_rest"], model._features_rest.detach())
    rs, ref_rs = box["settings"], c["settings"]
    assert (rs.image_height, rs.image_width, rs.sh_degree, int(rs.f_count)) == (c.H, c.W, int(ref_rs[5]), 0)
    assert close(rs.tanfovx, ref_rs[2], 1e-6) and close(rs.tanfovy, ref_rs[3], 1e-6) and rs.scale_modifier == ref_rs[4]
    assert box["dirs"] is dirs
    want_nd = c.num_dist if ("distortion" in extra or "depth_var" in extra) else 0
    assert (box["num_dist"] or 0) == want_nd or c.num_dist == want_nd
    total = tr._compute_loss(data, cam)
    total.backward()
    torch.cuda.synchronize()
    ref_l = c.losses()
    got = {k: float(v) for k, v in tr.losses.items()}
    assert set(ref_l) == set(got), (sorted(ref_l), sorted(got))
    for k, v in ref_l.items():
        assert close(got[k], v, 1e-4 if k not in ("depth_var", "total") or "depth_var" not in ref_l else 5e-3, 2e-7), (k, got[k], v)
    wide = "depth_var" in ref_l
    mx, p999 = field_err(box["dout"], g("d_rendered_out"))
    assert mx < (2e-2 if wide else 3e-4) and p999 < (5e-1 if wide else 3e-2), (mx, p999)
    attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                rotation="_rotation", obj_dc="_objects_dc")
    for k, a in attr.items():
        if f"grad_{k}" in c and (k != "obj_dc" or c.sem_on):
            gr = getattr(model, a).grad
            assert gr is not None, k
            mx, p999 = field_err(gr, g(f"grad_{k}"))
            assert mx < 2e-5 and p999 < 2e-3, (k, mx, p999)           # (the replayed input gradients: only the activation adjoint differs)
    if c.sem_on:
        assert field_err(model.classifier.weight.grad, g("grad_cls_w"))[0] < 2e-4
        assert field_err(model.classifier.bias.grad, g("grad_cls_b"))[0] < 2e-4
    # -- the reference's own call form: masks, normalised normals and depth-to-normal come back in the dictionary -----------
    with torch.no_grad():
        full = GR.render(cam, model, cfg, bg, dirs=dirs)
    assert torch.equal(full["mask"].cpu(), c.t("mask"))
    assert field_err(full["normal"], c.t("normal"))[0] < 1e-5 and field_err(full["est_normal"], c.t("est_normal"))[0] < 1e-4
    assert torch.equal(full["visibility_filter"].cpu(), c.t("radii") > 0)
