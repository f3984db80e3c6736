"""Round-3 reference fixtures: virtual visibility cameras (g11, `tools/camera_utils.py:315-481`), the volume-weighted pruning
score (g12, `tools/prune.py:6-22`) and the checkpoint wire format (g10: a `chkpnt3.pth` written by the reference's own
GaussianModel + torch.optim.Adam, and the parameters ONE MORE optimizer step gives) in both directions:
reference file -> this repo's `restore` -> continue training; this repo's `capture` -> `torch.optim.Adam.load_state_dict`."""
import json
import os
import types

import numpy as np
import pytest
import torch

from vcr_gaus_amd.config import make_config
from vcr_gaus_amd.gaussian_model import GaussianModel

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
          "rotation": "_rotation", "obj_dc": "_objects_dc"}


def load(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind in "fiub" else v) for k, v in np.load(os.path.join(G, name)).items()}


# ---- g11: bb_camera ------------------------------------------------------------------------------------------------
def _bb_cases():
    meta = json.loads(str(np.load(os.path.join(G, "g11_bb_camera.npz"))["meta"]))
    return sorted(meta)


@pytest.mark.parametrize("case", _bb_cases())
def test_bb_camera_matches_reference(case):
    from vcr_gaus_amd.camera_utils import bb_camera
    g = load("g11_bb_camera.npz")
    kw = json.loads(str(g["meta"]))[case]
    box = case.split("_", 1)[0]
    n, seed = kw.pop("n"), kw.pop("seed")
    if "target" in kw:
        kw["target"] = torch.tensor(kw["target"])
    torch.manual_seed(seed)                         # the reference draws the 'random' centres from the global RNG
    T = bb_camera(n, g[f"{box}_trans"], g[f"{box}_scale"], None, **kw)
    ref = g[case]
    assert T.shape == ref.shape, (T.shape, ref.shape)
    assert torch.allclose(T, ref, atol=1e-6), float((T - ref).abs().max())


def test_bb_camera_generator_equals_global_rng_and_sample_cam_matrices():
    from vcr_gaus_amd.camera_utils import bb_camera, sample_cameras
    g = load("g11_bb_camera.npz")
    meta = json.loads(str(g["meta"]))
    kw = meta["vec_random_around"]
    T = bb_camera(kw["n"], g["vec_trans"], g["vec_scale"], up=kw["up"], around=kw["around"], sample_mode="random",
                  generator=torch.Generator().manual_seed(kw["seed"]))
    assert torch.allclose(T, g["vec_random_around"], atol=1e-6)
    cams = sample_cameras(kw["n"], g["vec_trans"], g["vec_scale"], device="cpu", generator=torch.Generator().manual_seed(kw["seed"]))
    c = cams[1]                                      # `Trainer.sample_cameras` (`trainer.py:621-634`): 1500 x 1500, FoV 2.5
    assert (c.image_width, c.image_height, c.FoVx, c.FoVy) == (1500, 1500, 2.5, 2.5)
    assert torch.allclose(c.world_view_transform, g["vec_random_around_cam_view"], atol=1e-6)
    assert torch.allclose(c.full_proj_transform, g["vec_random_around_cam_full"], atol=1e-5)
    assert torch.allclose(c.camera_center, g["vec_random_around_cam_center"], atol=1e-5)


# ---- g12: calculate_v_imp_score --------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b"])
def test_v_imp_score_matches_reference(tag):
    from vcr_gaus_amd.prune import calculate_v_imp_score
    g = load("g12_v_imp_score.npz")
    for v_pow in (0.1, 0.5):
        got = calculate_v_imp_score(types.SimpleNamespace(get_scaling=g[f"{tag}_scaling"]), g[f"{tag}_imp"], v_pow)
        assert torch.allclose(got, g[f"{tag}_v{int(v_pow * 10)}"], rtol=1e-6, atol=0)


# ---- g10: checkpoint wire format ---------------------------------------------------------------------------------------
def _restored(device):
    cfg = make_config("tnt")
    cfg.model.enable_semantic, cfg.model.ch_sem_feat, cfg.model.num_cls = True, 2, 2
    from vcr_gaus_amd.trainer import load_capture
    ckpt, it = load_capture(os.path.join(G, "g10_chkpnt3.pth"))          # weights_only: no code from the file is executed
    m = GaussianModel(cfg.model)
    nxt = load("g10_chkpnt3_next_step.npz")
    m.classifier = torch.nn.Conv2d(2, 2, kernel_size=1)   # (the classifier's weights travel in model.pth, not in the checkpoint:
    with torch.no_grad():                               #  loaded BEFORE restore(), as the reference's load_ply path does)
        m.classifier.weight.copy_(nxt["classifier_weight"])
        m.classifier.bias.copy_(nxt["classifier_bias"])
    m.restore(ckpt, cfg.optim, device=device)
    return m, ckpt, it, nxt, cfg


def _wire_key(group_name):
    return {"classifier.weight": "classifier.0", "classifier.bias": "classifier.1"}.get(group_name, group_name)


def test_checkpoint_written_under_numpy_1_loads(tmp_path):
    """The reference's environment (pytorch 2.0.1) implies numpy 1.x, whose scalars pickle as `numpy.core.multiarray.scalar`;
    g10 was written under numpy 2 (`numpy._core...`).  The same file with the 1.x global name must load too."""
    import zipfile
    from vcr_gaus_amd.trainer import load_capture
    src = os.path.join(G, "g10_chkpnt3.pth")
    dst = str(tmp_path / "chkpnt_numpy1.pth")
    n = 0
    with zipfile.ZipFile(src) as zi, zipfile.ZipFile(dst, "w", zipfile.ZIP_STORED) as zo:
        for item in zi.infolist():
            data = zi.read(item.filename)
            if item.filename.endswith("data.pkl"):
                n = data.count(b"cnumpy._core.multiarray\nscalar\n")
                data = data.replace(b"cnumpy._core.multiarray\nscalar\n", b"cnumpy.core.multiarray\nscalar\n")   # (GLOBAL opcode: newline-terminated)
            zo.writestr(item, data)
    assert n > 0, "fixture holds no numpy scalar any more"
    a, it_a = load_capture(src)
    b, it_b = load_capture(dst)
    assert it_a == it_b and len(a) == len(b)
    for x, y in zip(a, b):
        if torch.is_tensor(x):
            assert torch.equal(x, y)
        elif not isinstance(x, dict):
            assert x == y or (x != x and y != y)


def test_reference_checkpoint_restores_into_this_model():
    m, ckpt, it, nxt, cfg = _restored("cpu")
    assert it == 3 and m.active_sh_degree == 2 and abs(m.spatial_lr_scale - 2.5) < 1e-12
    ref_opt = ckpt[11]
    names = [g["name"] for g in ref_opt["param_groups"]]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "obj_dc", "classifier"]
    idx = 0
    for wg in ref_opt["param_groups"]:
        mine = [g for g in m.optimizer.param_groups if g["name"].split(".")[0] == wg["name"]]
        assert len(mine) == len(wg["params"])
        for g, i in zip(mine, wg["params"]):
            assert i == idx
            idx += 1
            st, rs = m.optimizer.state[g["name"]], ref_opt["state"][i]
            assert st["step"] == 3 and isinstance(st["step"], int)
            assert torch.equal(st["exp_avg"], rs["exp_avg"]) and torch.equal(st["exp_avg_sq"], rs["exp_avg_sq"])
            assert g["lr"] == wg["lr"]
    for k, a in PARAMS.items():
        assert torch.equal(getattr(m, a).detach(), ckpt[{"xyz": 1, "f_dc": 2, "f_rest": 3, "scaling": 4, "rotation": 5,
                                                         "opacity": 6, "obj_dc": 7}[k]].detach())
    assert torch.equal(m.max_radii2D, ckpt[8]) and torch.equal(m.xyz_gradient_accum, ckpt[9]) and torch.equal(m.denom, ckpt[10])


def test_capture_loads_into_torch_adam_and_continues_like_the_reference():
    """this repo -> reference direction: `capture()`'s optimizer dictionary goes through `torch.optim.Adam.load_state_dict`
    (what the reference's `restore` does) and one more torch Adam step lands on the reference's parameters, bit for bit."""
    m, ckpt, it, nxt, cfg = _restored("cpu")
    cap = m.capture()
    assert len(cap) == 13 and cap[0] == 2
    sd = cap[11]
    assert all(isinstance(k, int) for k in sd["state"]) and all(torch.is_tensor(v["step"]) and v["step"].dtype == torch.float32
                                                                for v in sd["state"].values())
    # a round trip through this repo's own loader keeps everything (done first: torch's optimizer aliases the tensors it loads)
    m2 = GaussianModel(cfg.model)
    m2.classifier = torch.nn.Conv2d(2, 2, kernel_size=1)
    m2.restore(cap, cfg.optim, device="cpu")
    for name, st in m.optimizer.state.items():
        s2 = m2.optimizer.state[name]
        assert s2["step"] == st["step"] and torch.equal(s2["exp_avg"], st["exp_avg"]) and torch.equal(s2["exp_avg_sq"], st["exp_avg_sq"])
    # the reference's own group table (`scene/gaussian_model.py:241-256`), parameters from the captured tuple
    tensors = dict(xyz=cap[1], f_dc=cap[2], f_rest=cap[3], scaling=cap[4], rotation=cap[5], opacity=cap[6], obj_dc=cap[7])
    order = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "obj_dc"]
    groups = [{"params": [torch.nn.Parameter(tensors[k].detach().clone())], "lr": 0.123, "name": k} for k in order]
    groups.append({"params": [torch.nn.Parameter(p.detach().clone()) for p in m.classifier.parameters()], "lr": 0.5,
                   "name": "classifier"})
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    opt.load_state_dict(sd)
    for g in opt.param_groups:
        if g["name"] == "xyz":                          # `update_learning_rate(4)` of the next iteration
            g["lr"] = float(nxt["lr_xyz"])
        else:
            assert g["lr"] == pytest.approx(float(nxt[f"lr_{g['name']}"]), rel=0, abs=0)
        for k, p in enumerate(g["params"]):
            p.grad = nxt["grad_" + (g["name"] if len(g["params"]) == 1 else f"{g['name']}.{k}")].clone()
    opt.step()
    for g in opt.param_groups:
        for k, p in enumerate(g["params"]):
            want = nxt["after_" + (g["name"] if len(g["params"]) == 1 else f"{g['name']}.{k}")]
            assert torch.equal(p.detach(), want), g["name"]


def test_legacy_name_keyed_optimizer_state_still_loads():
    m, *_ = _restored("cpu")
    legacy = dict(state={k: dict(v) for k, v in m.optimizer.state.items()},
                  param_groups=[dict(name=g["name"], lr=g["lr"]) for g in m.optimizer.param_groups])
    before = {k: v["exp_avg"].clone() for k, v in m.optimizer.state.items()}
    m.optimizer.load_state_dict(legacy)
    assert all(torch.equal(m.optimizer.state[k]["exp_avg"], v) for k, v in before.items())
    # a legacy file saved before the first step has an EMPTY state: it is recognised by its groups (no `params` lists) and
    # its learning rates -- including the classifier's two single-tensor groups -- are taken over
    empty = dict(state={}, param_groups=[dict(name=g["name"], lr=0.25 + i) for i, g in enumerate(m.optimizer.param_groups)])
    m.optimizer.load_state_dict(empty)
    assert m.optimizer.state == {} and [g["lr"] for g in m.optimizer.param_groups] == [0.25 + i for i in range(len(m.optimizer.param_groups))]
    # moments of a legacy file land on the parameter's device in float32, and a wrong shape is refused
    half = {k: dict(step=3, exp_avg=v.double(), exp_avg_sq=v.double()) for k, v in before.items()}
    m.optimizer.load_state_dict(dict(state=half, param_groups=legacy["param_groups"]))
    assert all(v["exp_avg"].dtype == torch.float32 for v in m.optimizer.state.values())
    bad = dict(half, xyz=dict(step=3, exp_avg=torch.zeros(5, 3), exp_avg_sq=torch.zeros(5, 3)))
    with pytest.raises(ValueError, match="shape"):
        m.optimizer.load_state_dict(dict(state=bad, param_groups=legacy["param_groups"]))


def test_restore_without_a_classifier_skips_its_moments_and_says_so():
    """The checkpoint holds the classifier's Adam moments but not its weights (they live in the reference's model.pth): a
    classifier created by `restore()` is random, so its saved moments are dropped and a warning is raised; without `num_cls`
    the call refuses to guess."""
    from vcr_gaus_amd.trainer import load_capture
    cfg = make_config("tnt")
    ckpt, _ = load_capture(os.path.join(G, "g10_chkpnt3.pth"))
    cfg.model.enable_semantic, cfg.model.ch_sem_feat, cfg.model.num_cls = True, 2, 2
    m = GaussianModel(cfg.model)
    with pytest.warns(UserWarning, match="randomly initialised"):
        m.restore(ckpt, cfg.optim, device="cpu")
    assert "classifier.weight" not in m.optimizer.state and "xyz" in m.optimizer.state
    cfg.model.num_cls = 0
    with pytest.raises(ValueError, match="num_cls"):
        GaussianModel(cfg.model).restore(ckpt, cfg.optim, device="cpu")
    # with the classifier in place before restore() its moments are kept
    cfg.model.num_cls = 2
    m2 = GaussianModel(cfg.model)
    m2.classifier = torch.nn.Conv2d(2, 2, kernel_size=1)
    m2.restore(ckpt, cfg.optim, device="cpu")
    assert m2.optimizer.state["classifier.weight"]["step"] == 3


@pytest.mark.gpu
def test_restored_reference_checkpoint_continues_on_the_hip_adam(device):
    """reference -> this repo direction, continued on the device: restore the reference's chkpnt3.pth, feed the reference's
    next gradients to the fused HIP Adam and land on the reference's next parameters."""
    m, ckpt, it, nxt, cfg = _restored(device)
    m.update_learning_rate(it + 1)
    for g in m.optimizer.param_groups:
        wk = _wire_key(g["name"])
        assert g["lr"] == pytest.approx(float(nxt["lr_" + wk.split(".")[0]]), rel=1e-12)
        g["params"][0].grad = nxt["grad_" + wk].to(device)
    m.optimizer.step()
    torch.cuda.synchronize()
    for g in m.optimizer.param_groups:
        want = nxt["after_" + _wire_key(g["name"])]
        got = g["params"][0].detach().cpu()
        assert float((got - want).abs().max()) <= 2e-7 * max(1.0, float(want.abs().max())), g["name"]
        assert m.optimizer.state[g["name"]]["step"] == 4
