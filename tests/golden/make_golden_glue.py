"""g13: the reference's OWN operator glue -- `render()` (`gaussian_renderer/__init__.py:22-164`) and
`Trainer._compute_loss` / `Trainer._get_total_loss` (`trainer.py:233-321`) -- executed in the build container on the CPU.

The reference binds an un-vendored CUDA extension at `gaussian_renderer/__init__.py:16`; here that one import is served by a
module whose `GaussianRasterizer` calls the torch oracle (`oracle/raster_torch.py`).  Everything AROUND that call is
the reference's code, unmodified: the activations and the shortest-axis normal of its `GaussianModel`, the flip / rotate of
the normals, the keyword call of the rasterizer, the channel split, the masks, `F.normalize`, `compute_normals`, every
branch of the loss dictionary, the weighted total.  The fixture therefore pins the glue that `oracle/trainer_torch.py`
only restates: given the same rasterizer output, loss dictionary, total, d total / d rendered_out and the gradients that
reach the raw parameters must be what the reference's own code produces.

Fixtures hold inputs and the reference's outputs only.  Run:  python tests/golden/make_golden_glue.py
"""
import math
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402  (puts /root/reference on sys.path, stubs torchvision / PIL)
from oracle import raster_torch as OR  # noqa: E402

REF = MG.REF
DT = torch.float64      # see fp64_reference() below
CAPTURE = {}
NUM_DIST = [0]          # the fork's compile-time NUM_DIST (README.md:152-155), set per case


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    f_count: int = 0


class GaussianRasterizer(torch.nn.Module):
    """The stand-in for the absent extension: same constructor and keyword call as `gaussian_renderer/__init__.py:59,107-120`."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, means2D_densify=None, shs=None, colors_precomp=None, normals_precomp=None,
                semantics_precomp=None, scales=None, rotations=None, cov3D_precomp=None, dirs=None, inside=None):
        rs = self.raster_settings
        s = OR.Settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.scale_modifier, rs.viewmatrix,
                        rs.projmatrix, rs.sh_degree, rs.campos)
        out, radii, _ = OR.rasterize(s, means3D, means2D, means2D_densify, shs, colors_precomp, normals_precomp,
                                     semantics_precomp, opacities, scales, rotations, cov3D_precomp, dirs, num_dist=NUM_DIST[0])
        if out.requires_grad:
            out.retain_grad()
        CAPTURE["out"] = out
        # what the reference's glue handed to the extension (and, after backward, what came back for it)
        CAPTURE["args"] = dict(means3D=means3D, means2D=means2D, means2D_densify=means2D_densify, shs=shs, normals_precomp=normals_precomp,
                               semantics_precomp=semantics_precomp, opacities=opacities, scales=scales, rotations=rotations)
        for t in CAPTURE["args"].values():
            if t is not None and t.requires_grad and not t.is_leaf:
                t.retain_grad()
        CAPTURE["settings"] = rs
        return out, radii


def install_stubs():
    mod = types.ModuleType("diff_gaussian_rasterization")
    mod.GaussianRasterizationSettings, mod.GaussianRasterizer = GaussianRasterizationSettings, GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = mod
    RGM = MG._ref_model_module()                       # the reference's scene/gaussian_model.py on the CPU (+ torch.zeros patch)
    sys.modules["scene.gaussian_model"] = RGM
    sys.modules["scene"].GaussianModel = RGM.GaussianModel
    sys.modules["scene"].Scene = None
    for name in ["wandb", "imageio", "torchmetrics", "arguments", "lpips", "cv2", "matplotlib", "matplotlib.pyplot"]:
        sys.modules[name] = types.ModuleType(name)          # logging / plotting imports of trainer.py; never called here
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["torchmetrics"].JaccardIndex = None
    sys.modules["pytorch3d.ops"].knn_points = None
    sys.modules["torchvision"].utils = types.SimpleNamespace()
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules.setdefault("termcolor", tc)
    # every hard-coded "cuda" placement of the glue lands on the CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fname in ["zeros_like", "ones_like", "tensor", "ones", "empty", "full"]:
        real = getattr(torch, fname)

        def routed(*a, __real=real, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return __real(*a, **k)

        setattr(torch, fname, routed)
    return RGM


class fp64_reference:
    """Run the reference's glue in double precision so that the fixture is exact to ~1e-12 rather than to fp32 rounding (the
    depth-to-normal adjoint amplifies fp32 rounding to ~1e-2 on single entries): default dtype float64 for the tensors the glue
    creates (`torch.tensor(0.)`, the SSIM window ...) and its two explicit float32 casts (`gaussian_renderer/__init__.py:100`)
    rounding to float32 as written but continuing in double.  The arithmetic, its order and every branch stay the reference's."""

    def __enter__(self):
        self.real_to = torch.Tensor.to
        real_to = self.real_to

        def to(t, *a, **k):
            if torch.float32 in a or k.get("dtype") is torch.float32:      # honour the rounding, keep the working precision
                return real_to(real_to(t, *a, **k), DT)
            return real_to(t, *a, **k)

        torch.Tensor.to = to
        torch.set_default_dtype(DT)

    def __exit__(self, *exc):
        torch.Tensor.to = self.real_to
        torch.set_default_dtype(torch.float32)


def load_config(tag):
    from configs.config import Config
    path = {"dtu": "configs/dtu/dtu_scan24.yaml", "tnt": "configs/tnt/Barn.yaml", "360": "configs/360_v2/base.yaml"}[tag]
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        return Config(path)
    finally:
        os.chdir(cwd)


def scene_inputs(seed, N, H, W, focal, sem):
    """Seeded scene in the reference's storage layout + one reference `Camera` with ground-truth image / normal / label mask."""
    from vcr_gaus_amd import synthetic
    raw = synthetic.make_gaussians(N, seed=seed, sem_channels=sem)
    raw["scaling"] = raw["scaling"] + 1.8                  # footprints of a few pixels at this resolution
    g = torch.Generator().manual_seed(seed + 100)
    eye = synthetic.orbit_eyes(3, 3.0)[seed % 3]
    R, T = synthetic.look_at_colmap(eye)
    fovx, fovy = 2 * math.atan(W / (2 * focal)), 2 * math.atan(H / (2 * focal))
    image = torch.rand(3, H, W, generator=g, dtype=torch.float32).to(DT)
    normal = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g, dtype=torch.float32).to(DT) + torch.tensor([0.0, 0.0, -1.5]), dim=-1)
    labels = (torch.rand(H, W, generator=g) > 0.55).long()
    return raw, (R, T, fovx, fovy), image, normal, labels


def run_case(RGM, tag, seed, iteration, overrides=(), with_mask=False, N=400, H=48, W=64, focal=105.0, sh_degree=3):
    import gaussian_renderer as RR
    import trainer as RT
    from scene.cameras import Camera as RefCamera
    from tools.graphics_utils import get_all_px_dir  # noqa: F401  (calls .cuda(): patched)
    cfg = load_config(tag)
    lw = cfg.optim.loss_weight
    for k, v in overrides:
        setattr(lw, k, v)
    sem_on = float(getattr(lw, "semantic", 0)) > 0
    NUM_DIST[0] = 2 if float(getattr(lw, "depth_var", 0)) > 0 else (1 if float(getattr(lw, "distortion", 0)) > 0 else 0)
    cfg.model.enable_semantic = sem_on
    cfg.model.use_decoupled_appearance = False           # the appearance network is out of scope
    raw, (R, T, fovx, fovy), image, normal, labels = scene_inputs(seed, N, H, W, focal, 2)
    cam = RefCamera(0, R, T, fovx, fovy, image, None, "g13", 0, normal=normal, mask=(labels[..., None] if (sem_on or with_mask) else None),
                    data_device="cpu")
    cam.idx = 0
    for a in ["world_view_transform", "projection_matrix", "full_proj_transform", "camera_center", "intr"]:
        setattr(cam, a, getattr(cam, a).to(DT))          # (the float32 matrices the reference built, held in double)
    torch.manual_seed(seed)
    m = RGM.GaussianModel(cfg.model)
    attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                rotation="_rotation", obj_dc="_objects_dc")
    for k, a in attr.items():
        setattr(m, a, torch.nn.Parameter(raw[k].to(DT).clone().requires_grad_(True)))
    m.active_sh_degree = sh_degree
    m.trans, m.scale, m.extent = torch.zeros(3), torch.ones(3), 3.6
    if sem_on:
        m.classifier = m.classifier.to(DT)
    dirs = get_all_px_dir(cam.intr.float(), H, W).to(DT)        # built once per scene in fp32 by the reference (`scene/__init__.py:101-102`)
    tr = object.__new__(RT.Trainer)                      # no dataset / logging set-up: only what the two methods read
    tr.cfg, tr.model, tr.losses, tr.sphere = cfg, m, {}, False
    tr.weights = {key: value for key, value in cfg.optim.loss_weight.items() if value}      # trainer.py:142
    tr.current_iteration = iteration
    tr.scene = types.SimpleNamespace(dirs=dirs)
    bg = torch.tensor([0.25, 0.5, 0.75], dtype=DT)           # (fp32-exact)
    # ground truth near the scene's own render (a perturbed copy): with unrelated normals the confidence
    # weight exp((cos - 1) / exp_t) of the D-Normal term is ~0 everywhere and the term would go unchecked
    with torch.no_grad(), fp64_reference():
        pkg = RR.render(cam, m, cfg, bg, dirs=dirs)
        g = torch.Generator().manual_seed(seed + 200)
        seen = (pkg["alpha"][0] > 0.3)[..., None]
        jit = torch.nn.functional.normalize(pkg["normal"] + 0.12 * torch.randn(H, W, 3, generator=g), dim=-1)
        normal = torch.where(seen, jit, normal)
        image = (pkg["render"] + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0.0, 1.0)
        normal, image = normal.float().to(DT), image.float().to(DT)          # fp32-exact values: the product holds them in fp32
        cam.normal, cam.original_image = normal, image
    CAPTURE.clear()
    data = {"viewpoint_cam": cam, "bg": bg}
    with fp64_reference():
        total = tr.model_forward(data, "train")          # render -> _compute_loss -> _get_total_loss (trainer.py:225-231)
        total.backward()
    out = CAPTURE["out"]
    pre = f"{tag}{'_' + '_'.join(k for k, _ in overrides) if overrides else ''}_it{iteration}"
    res = {f"{pre}__in_{k}": raw[k] for k in raw}
    res.update({f"{pre}__cam_R": R, f"{pre}__cam_T": T, f"{pre}__cam_fov": np.array([fovx, fovy]), f"{pre}__gt_image": image.float(), f"{pre}__gt_normal": normal.float(), f"{pre}__labels": labels, f"{pre}__bg": bg, f"{pre}__hw": np.array([H, W]),
                f"{pre}__meta": np.array([iteration, sh_degree, NUM_DIST[0], int(sem_on), int(sem_on or with_mask)]),
                f"{pre}__extent": np.array(m.extent), f"{pre}__rendered_out": out.detach(), f"{pre}__d_rendered_out": out.grad,
                f"{pre}__total": total.detach(), f"{pre}__mask": data["mask"], f"{pre}__normal": data["normal"].detach(),
                f"{pre}__est_normal": data["est_normal"].detach(), f"{pre}__radii": data["radii"]})
    for k, v in tr.losses.items():
        res[f"{pre}__loss_{k}"] = torch.as_tensor(v).detach()
    for k, t in CAPTURE["args"].items():
        if k == "shs":        # [N,16,3] = cat(_features_dc, _features_rest) (`scene/gaussian_model.py:139-142`): stored as that fact
            res[f"{pre}__shs_is_cat"] = np.array(torch.equal(t.detach(), torch.cat([m._features_dc, m._features_rest], 1).detach()))
            continue
        if t is not None:
            if not k.startswith("means2D"):          # (the two gradient holders are zeros)
                res[f"{pre}__arg_{k}"] = t.detach()
            if t.grad is not None:
                res[f"{pre}__darg_{k}"] = t.grad
    rs = CAPTURE["settings"]
    res[f"{pre}__settings"] = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree,
                                        int(rs.prefiltered), int(rs.debug), rs.f_count], dtype=np.float64)
    res[f"{pre}__dirs"] = dirs.float()
    for k, a in attr.items():
        gr = getattr(m, a).grad
        if gr is not None:
            res[f"{pre}__grad_{k}"] = gr
    if sem_on:
        res[f"{pre}__cls_w"], res[f"{pre}__cls_b"] = m.classifier.weight.detach(), m.classifier.bias.detach()
        res[f"{pre}__grad_cls_w"], res[f"{pre}__grad_cls_b"] = m.classifier.weight.grad, m.classifier.bias.grad
    res[f"{pre}__weights"] = np.array(sorted(tr.weights.items()), dtype=object).astype(str)
    print(pre, "total", float(total), {k: round(float(v), 6) for k, v in tr.losses.items()})
    return pre, res


def main():
    RGM = install_stubs()
    allres, tags = {}, []
    cases = [("dtu", 1, 1, (), False),                                    # before any *_from_iter: l1, ssim, l1_scale, mono_normal
             ("dtu", 2, 15001, (), False),                                # consistent_normal + distortion (NUM_DIST = 1) active
             ("tnt", 3, 1, (), True),                                     # depth_normal with the cos weight + semantic loss + label mask
             ("360", 4, 1, (), False),                                    # depth_normal, mask from the depth threshold only
             ("360", 5, 7001, (("depth_var", 0.5), ("curv", 0.05), ("entropy", 0.01)), True)]   # the optional regularisers
    for tag, seed, it, ov, wm in cases:
        pre, res = run_case(RGM, tag, seed, it, ov, wm)
        tags.append(pre)
        allres.update(res)
    allres["cases"] = np.array(tags)
    MG.save("g13_reference_glue.npz", **allres)


if __name__ == "__main__":
    main()
