"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own Python
functions from /root/reference (possible only in the build container; the reference cannot travel).
Fixtures hold inputs and the reference's outputs only.  Run:  python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

# tools.general_utils imports torchvision / PIL at module level; they are not needed by the functions used
for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "PIL"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
sys.modules["PIL"].ImageFile = types.SimpleNamespace(LOAD_TRUNCATED_IMAGES=False)

from tools import graphics_utils as RG  # noqa: E402
from tools import loss_utils as RL  # noqa: E402
from tools import normal_utils as RN  # noqa: E402
from tools import sh_utils as RS  # noqa: E402
from tools import math_utils as RM  # noqa: E402
from tools import image_utils as RI  # noqa: E402
from tools import general_utils as RGU  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name)


def depth_cases():
    g = torch.Generator().manual_seed(0)
    out = {}
    for tag, (H, W) in {"a": (48, 64), "b": (37, 53), "c": (256, 256)}.items():     # c: SURVEY 8(c)'s 256 x 256 case
        K = RG.getIntrinsic(1.1, 0.9, H, W)
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        plane = 2.0 + 0.01 * xx + 0.02 * yy
        sphere = 3.0 - torch.sqrt(torch.clamp(1.0 - ((xx - W / 2) / W) ** 2 - ((yy - H / 2) / H) ** 2, min=0.05))
        rnd = 1.5 + torch.rand(H, W, generator=g)
        kinds = {"plane": plane, "sphere": sphere, "rand": rnd} if tag != "c" else {"sphere": sphere + 0.02 * (rnd - 2.0)}
        for dn, d in kinds.items():
            d = d[None].clone().requires_grad_(True)
            n = RN.compute_normals(d, K)
            gt = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
            rn = torch.nn.functional.normalize(gt + 0.3 * torch.randn(H, W, 3, generator=g), dim=-1)
            mask = torch.rand(H, W, generator=g) > 0.3
            w = RL.cos_weight(rn, gt, 0.01)
            loss = RL.monosdf_normal_loss(n[mask], gt[mask], w[mask])
            loss.backward()
            plain = RL.monosdf_normal_loss(n.detach(), gt)
            out.update({f"{tag}_{dn}_depth": d.detach(), f"{tag}_{dn}_K": K, f"{tag}_{dn}_normal": n.detach(),
                        f"{tag}_{dn}_gt": gt, f"{tag}_{dn}_rn": rn, f"{tag}_{dn}_mask": mask, f"{tag}_{dn}_w": w,
                        f"{tag}_{dn}_loss": loss.detach(), f"{tag}_{dn}_plain": plain,
                        f"{tag}_{dn}_ddepth": d.grad})
    save("g1_depth_normal.npz", **out)


def image_cases():
    g = torch.Generator().manual_seed(1)
    out = {}
    for tag, (H, W) in {"a": (40, 56), "b": (33, 70)}.items():
        a = torch.rand(3, H, W, generator=g).requires_grad_(True)
        b = (a.detach() + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
        l1 = RL.l1_loss(a, b)
        s = RL.ssim(a, b)
        (0.8 * l1 + 0.2 * (1 - s)).backward()
        out.update({f"{tag}_a": a.detach(), f"{tag}_b": b, f"{tag}_l1": l1.detach(), f"{tag}_ssim": s.detach(),
                    f"{tag}_grad": a.grad, f"{tag}_psnr": RI.psnr(a.detach(), b)})
    save("g2_l1_ssim.npz", **out)


def sh_cases():
    g = torch.Generator().manual_seed(2)
    N = 257
    sh = torch.randn(N, 3, 16, generator=g)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    out = {"sh": sh, "dirs": d}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = torch.clamp_min(RS.eval_sh(deg, sh, d) + 0.5, 0.0)   # gaussian_renderer/__init__.py:86-87
    out["rgb2sh"] = RS.RGB2SH(torch.linspace(0, 1, 11))
    save("g3_sh.npz", **out)


def camera_cases():
    out = {}
    g = np.random.RandomState(3)
    for i in range(3):
        A = g.randn(3, 3)
        R, _ = np.linalg.qr(A)
        if np.linalg.det(R) < 0:
            R[:, 0] *= -1
        T = g.randn(3)
        fovx, fovy = 0.6 + 0.3 * i, 0.5 + 0.2 * i
        H, W = 60 + 7 * i, 80 + 5 * i
        w2c = RG.getWorld2View2(R, T, np.array([0.1 * i, 0.0, -0.05]), 1.0 + 0.5 * i)
        view = torch.tensor(w2c).transpose(0, 1)
        proj = RG.getProjectionMatrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
        full = view.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        K = RG.getIntrinsic(fovx, fovy, H, W)
        # get_all_px_dir hard-codes .cuda(); same arithmetic on the CPU (tools/graphics_utils.py:143-155)
        _, ray = RG.depth2point_cam(torch.ones(1, 1, 1, H, W), K[None])
        dirs = torch.nn.functional.normalize(ray.squeeze(), dim=-1).permute(2, 0, 1)
        out.update({f"c{i}_R": R, f"c{i}_T": T, f"c{i}_trans": np.array([0.1 * i, 0.0, -0.05]), f"c{i}_scale": 1.0 + 0.5 * i,
                    f"c{i}_fov": np.array([fovx, fovy]), f"c{i}_hw": np.array([H, W]), f"c{i}_view": view, f"c{i}_proj": proj,
                    f"c{i}_full": full, f"c{i}_center": view.inverse()[3, :3], f"c{i}_K": K, f"c{i}_dirs": dirs,
                    f"c{i}_focal": np.array([RG.fov2focal(fovx, W), RG.focal2fov(RG.fov2focal(fovx, W), W)])})
    save("g4_cameras.npz", **out)


def misc_cases():
    g = torch.Generator().manual_seed(4)
    img = torch.rand(3, 30, 41, generator=g)
    dist = torch.rand(1, 30, 41, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(30, 41, 3, generator=g), dim=-1)
    mask = (torch.rand(30, 41, 1, generator=g) > 0.2).float()
    op = torch.rand(100, 1, generator=g)
    pts = torch.randn(200, 3, generator=g)
    trans, scale = torch.tensor([0.1, -0.2, 0.3]), torch.tensor([1.5, 1.0, 0.8])
    inside, npts = RM.get_inside_normalized(pts, trans, scale)
    lr = RGU.get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    save("g5_misc.npz", img=img, dist=dist, edge=RN.get_edge_aware_distortion_map(img, dist), nrm=nrm, mask=mask,
         curv=RL.normal2curv(nrm, mask), op=op, entropy=RL.entropy_loss(op), pts=pts, trans=trans, scale=scale,
         inside=inside, npts=npts, lr_steps=np.array([0, 1, 100, 15000, 30000]),
         lr_vals=np.array([lr(s) for s in [0, 1, 100, 15000, 30000]]))


def misc_grad_cases():
    """Gradients of the small image / per-Gaussian losses of `trainer.py:243-300` through the reference's functions."""
    g = torch.Generator().manual_seed(6)
    out = {}
    H, W = 37, 53
    img = torch.rand(3, H, W, generator=g)
    dist = torch.rand(1, H, W, generator=g).requires_grad_(True)
    ed = RN.get_edge_aware_distortion_map(img, dist).mean()            # trainer.py:295-298
    ed.backward()
    nrm = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).requires_grad_(True)
    mask = (torch.rand(H, W, 1, generator=g) > 0.25).float()
    cv = RL.l1_loss(RL.normal2curv(nrm, mask), 0)                       # trainer.py:282-287
    cv.backward()
    op = torch.rand(300, 1, generator=g).requires_grad_(True)
    en = RL.entropy_loss(op)                                             # trainer.py:247-249
    en.backward()
    save("g7_misc_grads.npz", img=img, dist=dist.detach(), edge_loss=ed.detach(), edge_grad=dist.grad, nrm=nrm.detach(),
         mask=mask, curv_loss=cv.detach(), curv_grad=nrm.grad, op=op.detach(), entropy_loss=en.detach(), entropy_grad=op.grad)


def tsdf_input_cases():
    """`tools/graphics_utils.py:134-141` (depth2point) and the depth masking of `tools/depth2mesh.py:37-52` (alpha threshold,
    gt alpha mask, bounding-box test on the back-projected world points) run through the reference's functions."""
    g = torch.Generator().manual_seed(8)
    out = {}
    for tag, (H, W) in {"a": (40, 56), "b": (33, 47)}.items():
        K = RG.getIntrinsic(0.9, 0.7, H, W)
        A = torch.randn(3, 3, generator=g)
        R, _ = torch.linalg.qr(A)
        w2c = torch.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R, torch.tensor([0.1, -0.2, 2.5])
        depth = (1.0 + 3.0 * torch.rand(1, H, W, generator=g))
        alpha = torch.rand(1, H, W, generator=g)
        gt_alpha = torch.rand(1, H, W, generator=g)
        trans, scale = torch.tensor([0.1, 0.0, -0.1]), torch.tensor([1.2, 1.0, 0.9])
        xyz_cam, xyz_world = RG.depth2point(depth[0], K, w2c)
        d = depth.clone()
        d[(gt_alpha < 0.5)] = 0                                   # depth2mesh.py:45-46
        d[(alpha < 0.5)] = 0                                      # :48 (alpha_thres 0.5)
        world = RG.depth2point(d[0], K, w2c)[1]                   # :50
        inside = RM.get_inside_normalized(world.view(-1, 3), trans, scale)[0]
        d.view(-1)[~inside] = 0                                   # :51-52
        out.update({f"{tag}_K": K, f"{tag}_w2c": w2c, f"{tag}_depth": depth, f"{tag}_alpha": alpha, f"{tag}_gt_alpha": gt_alpha,
                    f"{tag}_trans": trans, f"{tag}_scale": scale, f"{tag}_xyz_cam": xyz_cam, f"{tag}_xyz_world": xyz_world,
                    f"{tag}_masked": d})
    save("g8_tsdf_input.npz", **out)


def config_cases():
    """The reference's EFFECTIVE configurations (`configs/config.py` resolving `_parent_` chains of `configs/*.yaml`) for the
    three dataset presets, reduced to the keys of the hot path."""
    import json
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules.setdefault("termcolor", tc)
    from configs.config import Config
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        out = {}
        for tag, path in {"dtu": "configs/dtu/dtu_scan24.yaml", "tnt": "configs/tnt/Barn.yaml", "360": "configs/360_v2/base.yaml"}.items():
            c = Config(path)
            o, m = c.optim, c.model
            out[tag] = dict(
                loss_weight={k: float(v) for k, v in dict(o.loss_weight).items()},
                exp_t=float(o.exp_t), mask_depth_thr=float(o.mask_depth_thr), random_background=bool(o.random_background),
                normal_from_iter=int(o.normal_from_iter), dnormal_from_iter=int(o.dnormal_from_iter),
                consistent_normal_from_iter=int(o.consistent_normal_from_iter), close_depth_from_iter=int(o.close_depth_from_iter),
                densify_large=dict(percent_dense=float(o.densify_large.percent_dense),
                                   sample_cams=dict(random=bool(o.densify_large.sample_cams.random), num=int(o.densify_large.sample_cams.num))),
                prune=dict(iterations=[int(i) for i in o.prune.iterations], percent=float(o.prune.percent), decay=float(o.prune.decay),
                           v_pow=float(o.prune.v_pow)),
                iterations=int(o.iterations), position_lr_init=float(o.position_lr_init), position_lr_final=float(o.position_lr_final),
                position_lr_delay_mult=float(o.position_lr_delay_mult), position_lr_max_steps=int(o.position_lr_max_steps),
                feature_lr=float(o.feature_lr), opacity_lr=float(o.opacity_lr), scaling_lr=float(o.scaling_lr),
                rotation_lr=float(o.rotation_lr), percent_dense=float(o.percent_dense),
                densification_interval=int(o.densification_interval), opacity_reset_interval=int(o.opacity_reset_interval),
                densify_from_iter=int(o.densify_from_iter), densify_until_iter=int(o.densify_until_iter),
                densify_grad_threshold=float(o.densify_grad_threshold),
                sh_degree=int(m.sh_degree), white_background=bool(m.white_background), depth_type=str(m.depth_type),
                cls_lr=float(getattr(o, "cls_lr", 5e-4)))
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "g9_effective_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote g9_effective_configs.json")


def _scene_stub():
    """`scene` as a bare namespace package, so that `scene.cameras` imports without running `scene/__init__` (dataset readers)."""
    if "scene" not in sys.modules:
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        sys.modules["scene"] = pkg


def bb_camera_cases():
    """`tools/camera_utils.py:315-401` (`bb_camera`) in every placement mode, on a vector `trans` and on a 4x4 world -> box
    transform; the 'random' placements are seeded through the global RNG the reference draws from."""
    _scene_stub()
    from tools import camera_utils as RC
    out = {}
    A = torch.tensor([[0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]])       # world -> box: axes permuted, up flipped
    T4 = torch.eye(4)
    T4[:3, :3], T4[:3, 3] = A, torch.tensor([0.2, -0.1, 0.4])
    boxes = {"vec": (torch.tensor([0.3, -0.2, 0.5]), torch.tensor([1.5, 1.0, 0.8])), "mat": (T4, torch.tensor([1.2, 0.7, 2.0]))}
    cases = {
        "random_around": dict(n=17, up=False, around=True, sample_mode="random"),
        "random_up": dict(n=9, up=True, around=False, sample_mode="random"),
        "random_both": dict(n=40, up=True, around=True, sample_mode="random", bidirect=True),
        "grid_both": dict(n=60, up=True, around=True, sample_mode="grid"),
        "grid_around": dict(n=50, up=False, around=True, sample_mode="grid"),
        "grid_up": dict(n=30, up=True, around=False, sample_mode="grid"),
        "grid_bidirect": dict(n=80, up=False, around=True, sample_mode="grid", bidirect=True),
        "grid_bidirect_up": dict(n=90, up=True, around=True, sample_mode="grid", bidirect=True),
        "grid_direction": dict(n=70, up=True, around=True, sample_mode="grid", look_mode="direction"),
        "random_direction": dict(n=21, up=True, around=True, sample_mode="random", look_mode="direction"),
        "grid_opengl_target": dict(n=45, up=False, around=True, sample_mode="grid", opengl=True, target=np.array([[0.1, 0.2, -0.3]], dtype=np.float32)),
    }
    import json
    meta = {}
    for bname, (trans, scale) in boxes.items():
        out[f"{bname}_trans"], out[f"{bname}_scale"] = trans, scale
        for cname, kw in cases.items():
            kw = dict(kw)
            n = kw.pop("n")
            seed = 100 + len(meta)
            torch.manual_seed(seed)
            out[f"{bname}_{cname}"] = RC.bb_camera(n, trans, scale, None, **kw)
            meta[f"{bname}_{cname}"] = dict(n=n, seed=seed, **{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    out["meta"] = np.array(json.dumps(meta))
    # SampleCam matrices of `Trainer.sample_cameras` (`trainer.py:621-634`: 1500 x 1500, FoV 2.5 rad) for two of them
    from scene.cameras import SampleCam
    for k in ("vec_random_around", "mat_grid_both"):
        cam = SampleCam(out[k][1], 1500, 1500, 2.5, 2.5, device="cpu")
        out[f"{k}_cam_view"], out[f"{k}_cam_full"], out[f"{k}_cam_center"] = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
    save("g11_bb_camera.npz", **out)


def prune_score_cases():
    """`tools/prune.py:6-22` (`calculate_v_imp_score`) and the percentile threshold of `GaussianModel.prune_gaussians` inputs."""
    gr = types.ModuleType("gaussian_renderer")
    gr.count_render = gr.visi_acc_render = None
    sys.modules.setdefault("gaussian_renderer", gr)
    from tools import prune as RP
    g = torch.Generator().manual_seed(12)
    out = {}
    for tag, n in {"a": 1000, "b": 37}.items():
        scaling = torch.exp(torch.randn(n, 3, generator=g) - 3.0)
        imp = torch.rand(n, generator=g) * (torch.rand(n, generator=g) > 0.2)
        for v_pow in (0.1, 0.5):
            out[f"{tag}_v{int(v_pow * 10)}"] = RP.calculate_v_imp_score(types.SimpleNamespace(get_scaling=scaling), imp, v_pow)
        out[f"{tag}_scaling"], out[f"{tag}_imp"] = scaling, imp
    save("g12_v_imp_score.npz", **out)


def _load_reference_gaussian_model():
    """Import /root/reference/scene/gaussian_model.py on the CPU: stub the absent third-party modules, keep `scene/__init__`
    (dataset readers, PIL, cv2 ...) from running, and route the hard-coded device="cuda" allocations to the CPU."""
    import importlib.util
    for name in ["plyfile", "simple_knn", "simple_knn._C", "pytorch3d", "pytorch3d.ops", "open3d"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    captured = {}

    class _PlyElement:                              # stand-ins that CAPTURE what the reference hands to plyfile
        @staticmethod
        def describe(elements, name):
            captured["names"], captured["elements"], captured["element"] = list(elements.dtype.names), elements, name
            return elements

    class _PlyData:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path

    sys.modules["plyfile"].PlyData, sys.modules["plyfile"].PlyElement = _PlyData, _PlyElement
    sys.modules["plyfile"].captured = captured
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["pytorch3d.ops"].ball_query = sys.modules["pytorch3d.ops"].knn_points = None
    pkg = types.ModuleType("scene")
    pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = pkg
    real_zeros = torch.zeros

    def zeros(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return real_zeros(*a, **k)

    torch.zeros = zeros
    torch.cuda.memory_allocated = lambda *a, **k: 0
    torch.cuda.empty_cache = lambda: None
    torch.nn.Module.cuda = lambda self, *a, **k: self
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene", "gaussian_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_RGM = None


def _ref_model_module():
    global _RGM
    if _RGM is None:
        _RGM = _load_reference_gaussian_model()
    return _RGM


def checkpoint_cases():
    """The reference's checkpoint wire format -- `torch.save((model.capture(), iteration), "chkpntN.pth")`
    (`trainer.py:425-430`, `scene/gaussian_model.py:88-123`) -- written by the reference's own GaussianModel and
    torch.optim.Adam after three real optimizer steps, plus what ONE MORE step from that state gives (gradients in,
    parameters out), so that a restore can be checked by continuing the run."""
    RGM = _ref_model_module()
    N = 48
    cfgm = types.SimpleNamespace(sh_degree=3, max_mem=22, use_decoupled_appearance=False, enable_semantic=True, ch_sem_feat=2,
                                 num_cls=2)
    targs = types.SimpleNamespace(percent_dense=0.01, densify_large=types.SimpleNamespace(percent_dense=2e-3),
                                  position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                  position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3,
                                  rotation_lr=1e-3, cls_lr=5e-4)
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(21)                                  # (the classifier's Conv2d initialisation)
    m = RGM.GaussianModel(cfgm)
    shapes = dict(_xyz=(N, 3), _features_dc=(N, 1, 3), _features_rest=(N, 15, 3), _opacity=(N, 1), _scaling=(N, 3),
                  _rotation=(N, 4), _objects_dc=(N, 1, 2))
    for a, shp in shapes.items():
        setattr(m, a, torch.nn.Parameter(torch.randn(shp, generator=g).requires_grad_(True)))
    m.max_radii2D = torch.zeros(N)
    m.spatial_lr_scale = 2.5
    m.training_setup(targs)

    def one_step(it):
        m.update_learning_rate(it)
        grads = {}
        for grp in m.optimizer.param_groups:
            for k, p in enumerate(grp["params"]):
                p.grad = 1e-2 * torch.randn(p.shape, generator=g) * (torch.rand(p.shape, generator=g) > 0.1)
                grads[grp["name"] if len(grp["params"]) == 1 else f"{grp['name']}.{k}"] = p.grad.clone()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        return grads

    for it in (1, 2, 3):
        one_step(it)
    m.active_sh_degree = 2
    m.xyz_gradient_accum = 1e-3 * torch.rand(N, 1, generator=g)
    m.denom = torch.randint(0, 3, (N, 1), generator=g).float()
    m.max_radii2D = 30.0 * torch.rand(N, generator=g)
    cls0 = [p.detach().clone() for p in m.classifier.parameters()]
    torch.save((m.capture(), 3), os.path.join(HERE, "g10_chkpnt3.pth"))
    print("wrote g10_chkpnt3.pth")
    grads = one_step(4)
    out = {f"grad_{k}": v for k, v in grads.items()}
    for grp in m.optimizer.param_groups:
        for k, p in enumerate(grp["params"]):
            out["after_" + (grp["name"] if len(grp["params"]) == 1 else f"{grp['name']}.{k}")] = p.detach().clone()
        out[f"lr_{grp['name']}"] = np.array(grp["lr"])
    out["classifier_weight"], out["classifier_bias"] = cls0
    save("g10_chkpnt3_next_step.npz", **out)


def densify_cases():
    """`scene/gaussian_model.py:361-364,425-671` executed by the reference's own GaussianModel on seeded inputs: the
    state (parameters, Adam moments, densification statistics) before and after each operation."""
    RGM = _ref_model_module()
    out = {}
    N, extent = 320, 3.3
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "obj_dc"]
    attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                rotation="_rotation", obj_dc="_objects_dc")
    cfgm = types.SimpleNamespace(sh_degree=3, max_mem=22, use_decoupled_appearance=False, enable_semantic=True, ch_sem_feat=2,
                                 num_cls=2)
    targs = types.SimpleNamespace(percent_dense=0.01, densify_large=types.SimpleNamespace(percent_dense=2e-3),
                                  position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                  position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3,
                                  rotation_lr=1e-3, cls_lr=5e-4)

    def fresh(seed):
        g = torch.Generator().manual_seed(seed)
        raw = dict(xyz=1.3 * (2 * torch.rand(N, 3, generator=g) - 1),                   # some outside the unit box
                   f_dc=torch.randn(N, 1, 3, generator=g), f_rest=0.05 * torch.randn(N, 15, 3, generator=g),
                   # scales straddle percent_dense * extent (0.033), large_percent_dense * extent (0.0066) and 0.1 * extent
                   scaling=math.log(0.03) + 1.2 * torch.randn(N, 1, generator=g) + 0.5 * torch.randn(N, 3, generator=g),
                   rotation=torch.randn(N, 4, generator=g),                                # un-normalised, as trained
                   opacity=1.5 * torch.randn(N, 1, generator=g) - 4.0 * (torch.rand(N, 1, generator=g) > 0.8),
                   obj_dc=torch.randn(N, 1, 2, generator=g))
        m = RGM.GaussianModel(cfgm)
        for k in names:
            setattr(m, attr[k], torch.nn.Parameter(raw[k].clone().requires_grad_(True)))
        m.max_radii2D = torch.zeros(N)
        m.trans, m.scale, m.extent = torch.zeros(3), torch.ones(3), extent
        m.spatial_lr_scale = 3.0
        m.training_setup(targs)
        for grp in m.optimizer.param_groups:        # populate Adam state with non-trivial moments
            if grp["name"] == "classifier":
                continue
            p = grp["params"][0]
            m.optimizer.state[p] = dict(step=torch.tensor(7.0), exp_avg=0.01 * torch.randn(p.shape, generator=g),
                                        exp_avg_sq=1e-4 * torch.rand(p.shape, generator=g))
        m.xyz_gradient_accum = 2e-3 * torch.rand(N, 1, generator=g) * (torch.rand(N, 1, generator=g) > 0.3)
        m.denom = torch.randint(0, 4, (N, 1), generator=g).float()          # zeros -> NaN grads -> 0 (`:644-645`)
        m.max_radii2D = 40.0 * torch.rand(N, generator=g)
        return m, g

    def dump(tag, m):
        for grp in m.optimizer.param_groups:
            if grp["name"] == "classifier":
                continue
            p = grp["params"][0]
            out[f"{tag}_{grp['name']}"] = p.detach().clone()
            st = m.optimizer.state.get(p)
            if st is not None:
                out[f"{tag}_{grp['name']}_m"], out[f"{tag}_{grp['name']}_v"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
        out[f"{tag}_accum"], out[f"{tag}_denom"], out[f"{tag}_radii"] = m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()

    # (a) add_densification_stats (+ the max_radii2D update of trainer.py:345)
    m, g = fresh(11)
    dump("stats_in", m)
    vp = torch.zeros(N, 3)
    vp.grad = 1e-3 * torch.randn(N, 3, generator=g)
    radii = ((torch.rand(N, generator=g) > 0.35) * torch.randint(1, 60, (N,), generator=g)).int()
    vis = radii > 0
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
    m.add_densification_stats(vp, vis)
    out["stats_vpgrad"], out["stats_radii"] = vp.grad, radii
    dump("stats_out", m)
    # (b) clone, (c) split, (d) prune_points, (e) reset_opacity, (g) prune_gaussians: one operation each from the same start
    for op in ["clone", "split", "split_visi", "prune", "reset", "prune_gaussians", "densify_and_prune", "densify_and_prune_sized"]:
        m, g = fresh(12)
        if op == "clone":
            dump("start", m)
        grads = m.xyz_gradient_accum / m.denom
        grads[grads.isnan()] = 0.0
        visi = torch.rand(N, generator=g) > 0.4          # [N] as `trainer.py:360-366` passes it (pre-clone length)
        if op == "clone":
            m.densify_and_clone(grads, 5e-4, extent)
        elif op == "split":
            m.densify_and_split_along_maxscaling(grads, 5e-4, extent)
        elif op == "split_visi":
            out["split_visi_mask"] = visi
            m.densify_and_split_along_maxscaling(grads, 5e-4, extent, visi=visi)
        elif op == "prune":
            mask = torch.rand(N, generator=g) > 0.6
            out["prune_mask"] = mask
            m.prune_points(mask)
        elif op == "reset":
            m.reset_opacity()
        elif op == "prune_gaussians":
            score = torch.rand(N, generator=g)
            out["prune_gaussians_score"] = score
            m.prune_gaussians(0.3, score)
        elif op == "densify_and_prune":
            out["dap_visi"] = visi
            m.densify_and_prune(5e-4, 0.005, extent, None, visi)
        else:
            out["daps_visi"] = visi
            m.densify_and_prune(5e-4, 0.005, extent, 20, visi)
        dump(op, m)
    out["extent"] = np.array(extent)
    # PLY wire format (`scene/gaussian_model.py:272-311`): field names + the vertex table the reference passes to plyfile
    m, g = fresh(13)
    import tempfile
    real_save = torch.save
    torch.save = lambda *a, **k: None               # (model.pth with the classifier weights is not part of the fixture)
    try:
        m.save_ply(os.path.join(tempfile.mkdtemp(), "point_cloud", "point_cloud.ply"))
    finally:
        torch.save = real_save
    cap = sys.modules["plyfile"].captured
    out["ply_names"] = np.array(cap["names"])
    out["ply_table"] = np.stack([cap["elements"][nm] for nm in cap["names"]], 1).astype("<f4")
    dump("ply", m)
    save("g6_densify.npz", **out)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = dict(g1=depth_cases, g2=image_cases, g3=sh_cases, g4=camera_cases, g5=misc_cases, g7=misc_grad_cases, g8=tsdf_input_cases, g9=config_cases, g11=bb_camera_cases, g12=prune_score_cases,
                g6=densify_cases, g10=checkpoint_cases)
    for k, fn in todo.items():          # g6 / g10 last: it monkey-patches torch.zeros / torch.cuda for the reference model
        if not a.only or k in a.only.split(","):
            fn()
