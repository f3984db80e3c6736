"""GPU parity of the image-space / per-Gaussian HIP kernels against the reference's golden vectors
(tests/golden/*.npz, captured from the reference's own functions) and the torch oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_torch as OL
from oracle import model_torch as OM

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name)).items()}


@pytest.mark.parametrize("tag", ["a_plane", "a_sphere", "a_rand", "b_plane", "b_sphere", "b_rand", "c_sphere"])
def test_depth_normal_and_dnormal_loss_vs_reference_vectors(device, tag):
    from vcr_gaus_amd.loss_utils import normal_loss
    from vcr_gaus_amd.normal_utils import compute_normals
    g = load("g1_depth_normal.npz")
    d = g[f"{tag}_depth"].to(device).requires_grad_(True)
    n = compute_normals(d, g[f"{tag}_K"])
    assert torch.allclose(n.cpu(), g[f"{tag}_normal"], atol=5e-5 if tag[0] != "c" else 1.5e-4)      # fp32 tolerance (c: see test_oracle_cpu)
    loss = normal_loss(n, g[f"{tag}_gt"].to(device), weight_src=g[f"{tag}_rn"].to(device), exp_t=0.01,
                       mask=g[f"{tag}_mask"].to(device))
    ref = float(g[f"{tag}_loss"])
    assert abs(float(loss) - ref) < 1e-4 * max(1.0, abs(ref))
    loss.backward()
    rd = g[f"{tag}_ddepth"]
    assert float((d.grad.cpu() - rd).abs().max()) <= 1e-3 * float(rd.abs().max()) + 1e-7
    plain = normal_loss(n.detach(), g[f"{tag}_gt"].to(device))
    assert abs(float(plain) - float(g[f"{tag}_plain"])) < 1e-4 * max(1.0, abs(float(g[f"{tag}_plain"])))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_l1_ssim_vs_reference_vectors(device, tag):
    from vcr_gaus_amd.loss_utils import l1_ssim, psnr
    g = load("g2_l1_ssim.npz")
    a = g[f"{tag}_a"].to(device).requires_grad_(True)
    b = g[f"{tag}_b"].to(device)
    l1, s = l1_ssim(a, b)
    assert abs(float(l1) - float(g[f"{tag}_l1"])) < 1e-6
    assert abs(float(s) - float(g[f"{tag}_ssim"])) < 2e-5
    (0.8 * l1 + 0.2 * (1 - s)).backward()
    rg = g[f"{tag}_grad"]
    assert float((a.grad.cpu() - rg).abs().max()) <= 1e-3 * float(rg.abs().max())
    assert torch.allclose(psnr(a.detach(), b).cpu(), g[f"{tag}_psnr"], atol=1e-3)


def test_normal_consistency_loss_grads_both_sides(device):
    from vcr_gaus_amd.loss_utils import normal_loss
    g = torch.Generator().manual_seed(0)
    a = torch.nn.functional.normalize(torch.randn(40, 50, 3, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(40, 50, 3, generator=g), dim=-1)
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    OL.monosdf_normal_loss(ad, bd).backward()
    ah, bh = a.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    l = normal_loss(ah, bh)
    l.backward()
    assert torch.allclose(ah.grad.cpu().double(), ad.grad, atol=1e-7)
    assert torch.allclose(bh.grad.cpu().double(), bd.grad, atol=1e-7)


def test_normalize_rendered_normal(device):
    from vcr_gaus_amd.normal_utils import normalize_rendered_normal
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 33, 47, generator=g)
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xd.permute(1, 2, 0), dim=-1)
    w = torch.randn(33, 47, 3, generator=g)
    (ref * w.double()).sum().backward()
    xh = x.to(device).requires_grad_(True)
    out = normalize_rendered_normal(xh)
    (out * w.to(device)).sum().backward()
    assert torch.allclose(out.cpu().double(), ref.detach(), atol=1e-6)
    assert torch.allclose(xh.grad.cpu().double(), xd.grad, atol=1e-5)


def test_fused_activation_and_camera_normals(device):
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.gaussian_model import _FusedActivate
    raw = synthetic.make_gaussians(5000, seed=5)
    cam = synthetic.make_cameras(3, 64, 48, 60.0)[1]
    rd = {k: v.double().requires_grad_(True) for k, v in raw.items()}
    act = OM.activations(rd)
    nw = OM.get_normal(act["rotation"], act["scaling"])
    nc = OM.camera_normals(nw, act["xyz"], cam.camera_center.double(), cam.R_w2c.double())
    g = torch.Generator().manual_seed(2)
    ws = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in (act["scaling"], act["rotation"], act["opacity"], nc)]
    sum((t * w).sum() for t, w in zip((act["scaling"], act["rotation"], act["opacity"], nc), ws)).backward()
    rh = {k: v.to(device).requires_grad_(True) for k, v in raw.items()}
    s, r, o, n = _FusedActivate.apply(rh["scaling"], rh["rotation"], rh["opacity"], rh["xyz"], cam.camera_center.to(device),
                                      cam.R_w2c.to(device), True)
    for got, ref in zip((s, r, o, n), (act["scaling"], act["rotation"], act["opacity"], nc)):
        assert torch.allclose(got.cpu().double(), ref.detach(), atol=2e-6, rtol=1e-5)
    sum((t * w.float().to(device)).sum() for t, w in zip((s, r, o, n), ws)).backward()
    for k in ["scaling", "rotation", "opacity"]:
        e = float((rh[k].grad.cpu().double() - rd[k].grad).abs().max() / rd[k].grad.abs().max())
        assert e < 1e-5, (k, e)


def test_fused_adam_matches_torch(device):
    from vcr_gaus_amd.gaussian_model import FusedAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 4)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 1e-3]
    ps = [torch.randn(s, generator=g) for s in shapes]  # 1001: exercises the scalar tails
    tp = [torch.nn.Parameter(p.clone().to(device)) for p in ps]
    hp = [torch.nn.Parameter(p.clone().to(device)) for p in ps]
    topt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(tp, lrs)], lr=0.0, eps=1e-15)
    hopt = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(hp, lrs))], eps=1e-15)
    for it in range(5):
        for a, b in zip(tp, hp):
            gr = (torch.randn(a.shape, generator=g) * 10.0 ** (it - 3)).to(device)
            a.grad, b.grad = gr.clone(), gr.clone()
        topt.step(); hopt.step()
    for a, b in zip(tp, hp):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_render_keys_and_training_reduces_loss(device):
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(20000, seed=0)
    raw["scaling"] = raw["scaling"] + 1.0
    cams = synthetic.make_cameras(4, 160, 120, 140.0, device=device)
    tr = make_synthetic_trainer(raw, cams, device, preset="dtu_c3", gt_jitter=0.05,
                                optim={"densify_from_iter": 40, "densification_interval": 10, "densify_until_iter": 1000})
    n0 = tr.model._xyz.shape[0]
    hist = []
    for i in range(60):
        data = tr.train_step()
        hist.append(float(tr.losses["total"]))
        assert np.isfinite(hist[-1])
    first, tot = sum(hist[:4]) / 4, sum(hist[36:40]) / 4      # before the first densification (iteration 50)
    from vcr_gaus_amd.gaussian_renderer import render
    data = render(cams[0], tr.model, tr.cfg, tr.background, dirs=tr.dirs)        # the reference's operator surface
    for k in ["render", "depth", "normal", "est_normal", "alpha", "viewspace_points", "viewspace_points_densify",
              "visibility_filter", "mask", "radii"]:
        assert k in data and data[k] is not None
    assert data["render"].shape == (3, 120, 160) and data["normal"].shape == (120, 160, 3)
    assert data["est_normal"].shape == (120, 160, 3) and data["depth"].shape == (1, 120, 160)
    assert tot < first, (first, tot)
    assert tr.model._xyz.shape[0] != n0          # densify / prune ran
    for k in ["l1", "ssim", "l1_scale", "mono_normal", "depth_normal", "consistent_normal"]:
        assert k in tr.losses


def test_visibility_and_importance_passes(device):
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(5000, seed=1)
    raw["scaling"] = raw["scaling"] + 1.0
    cams = synthetic.make_cameras(3, 96, 64, 80.0, device=device)
    tr = make_synthetic_trainer(raw, cams, device)
    vis = tr.visibility_mask(cams)
    cnt, imp = tr.importance_scores(cams)
    assert vis.dtype == torch.bool and vis.shape[0] == 5000 and int(vis.sum()) > 0
    assert bool(((cnt > 0) >= vis).all())      # visible & inside implies counted
    assert float(imp.min()) >= 0.0 and float(imp.max()) > 0.0
    assert tr.v_imp_score(imp, 0.1).shape[0] == 5000
    # the remaining wrappers of the reference's renderer module and its tools/prune.py surface
    from vcr_gaus_amd.gaussian_renderer import count_render, render, render_fast, visi_render
    from vcr_gaus_amd.prune import get_visi_list, prune_list
    full = render(cams[0], tr.model, tr.cfg, tr.background, dirs=None)
    fast = render_fast(cams[0], tr.model, tr.cfg, tr.background)
    assert set(fast) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert torch.equal(fast["render"], full["render"]) and torch.equal(fast["radii"], full["radii"])
    fast["render"].sum().backward()
    assert tr.model._features_dc.grad is not None and fast["viewspace_points"].grad is not None
    tr.model.optimizer.zero_grad()
    vr, cr = visi_render(cams[1], tr.model, tr.cfg.pipline, tr.background), count_render(cams[1], tr.model, tr.cfg.pipline, tr.background)
    assert torch.equal(vr["countlist"], cr["gaussians_count"])
    assert torch.allclose(vr["important_score"], cr["important_score"], rtol=1e-5, atol=1e-7)     # (fp32 atomics: run-to-run order)
    assert not vr["important_score"].requires_grad and not cr["gaussians_count"].requires_grad
    c2, i2 = prune_list(tr.model, list(cams), tr.cfg.pipline, tr.background)
    assert torch.equal(c2, cnt) and torch.allclose(i2, imp, rtol=1e-5, atol=1e-6)
    assert torch.equal(get_visi_list(tr.model, list(cams), tr.cfg.pipline, tr.background)["visi"] & tr.model.get_inside_gaus_normalized()[0], vis)
    # GaussianModel.get_normal (`scene/gaussian_model.py:168-192`): zero rows outside the bounding box unless is_all
    m = tr.model
    m.scale = torch.full((3,), 0.5, device=device)
    inside = m.get_inside_gaus_normalized()[0]
    act = OM.activations({k: getattr(m, a).detach().cpu() for k, a in dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling", rotation="_rotation").items()})
    want = OM.get_normal(act["rotation"], act["scaling"])
    n_all, n_in = m.get_normal(is_all=True).cpu(), m.get_normal().cpu()
    assert 0 < int(inside.sum()) < inside.numel()
    assert torch.allclose(n_all, want.float(), atol=1e-5)
    assert torch.equal(n_in[~inside.cpu()], torch.zeros_like(n_in[~inside.cpu()])) and torch.equal(n_in[inside.cpu()], n_all[inside.cpu()])


def test_scale_regulariser_and_fused_depth_mask(device):
    from vcr_gaus_amd.loss_utils import normal_loss, scale_regulariser
    g = torch.Generator().manual_seed(7)
    N = 4097
    sc = torch.randn(N, 3, generator=g) - 3
    xyz = torch.randn(N, 3, generator=g)
    trans, scale = torch.tensor([0.1, -0.2, 0.05]), torch.tensor([1.2, 0.9, 1.1])
    sd = sc.double().requires_grad_(True)
    inside = torch.all(torch.abs((xyz.double() - trans.double()) / scale.double()) < 1, dim=-1)      # tools/math_utils.py:70-74
    ref = torch.exp(sd)[inside].min(-1)[0].abs().mean()                                               # trainer.py:243-245
    ref.backward()
    sh = sc.to(device).requires_grad_(True)
    l = scale_regulariser(sh, xyz.to(device), trans.to(device), scale.to(device))
    l.backward()
    assert abs(float(l) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    assert torch.allclose(sh.grad.cpu().double(), sd.grad, atol=1e-9, rtol=1e-4)
    # depth threshold fused into the loss kernel == explicit boolean mask
    H, W = 37, 53
    p = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).to(device)
    q = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).to(device)
    depth = (torch.rand(1, H, W, generator=g) * 4).to(device)
    base = torch.rand(H, W, generator=g).to(device) > 0.3
    a = normal_loss(p, q, weight_src=p, exp_t=0.01, mask=base & (depth[0] < 2.5))
    b = normal_loss(p, q, weight_src=p, exp_t=0.01, mask=base, depth=depth, depth_max=2.5)
    assert float(a) == float(b)


def test_knn3_init_matches_bruteforce(device):
    """vcr_knn3_mean_dist2 (replaces simple_knn.distCUDA2) and create_from_pcd (`scene/gaussian_model.py:199-229`)."""
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(11)
    pts = torch.rand(3000, 3, generator=g)
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    ref = d.topk(3, largest=False).values.mean(1)
    m = GaussianModel(make_config("tnt").model)
    m.create_from_pcd(pts.numpy(), torch.rand(3000, 3, generator=g).numpy(), 1.0, device=device)
    got = torch.exp(m._scaling.detach()[:, 0]).cpu().double() ** 2
    assert torch.allclose(got, ref.clamp_min(1e-7), rtol=1e-4)
    assert m._rotation.shape == (3000, 4) and float(m._rotation[:, 0].min()) == 1.0
    assert abs(float(torch.sigmoid(m._opacity).mean()) - 0.1) < 1e-6 and m._features_rest.abs().max() == 0


def _knn3_reference(pts):
    """mean of the 3 smallest squared distances to OTHER points, fp64, chunked"""
    p = pts.double()
    out = torch.empty(len(p), dtype=torch.float64)
    for a in range(0, len(p), 2048):
        d = torch.cdist(p[a:a + 2048], p) ** 2
        d[torch.arange(d.shape[0]), torch.arange(a, a + d.shape[0])] = float("inf")
        out[a:a + 2048] = d.topk(3, largest=False).values.mean(1)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("cloud", ["uniform", "clustered", "planar", "line_with_duplicates", "offset_far", "n2049"])
def test_knn3_grid_search_is_exact(device, cloud):
    """The grid search of vcr_knn3_mean_dist2 (N > 2048) returns what the brute force returns on clouds that stress it:
    empty cells between clusters, a degenerate axis, coincident points, coordinates far from the origin."""
    from vcr_gaus_amd import _lib
    g = torch.Generator().manual_seed(3)
    n = 20000
    if cloud == "uniform":
        pts = torch.rand(n, 3, generator=g) * torch.tensor([4.0, 1.0, 0.3])
    elif cloud == "clustered":
        centres = torch.randn(12, 3, generator=g) * 5
        pts = centres[torch.randint(0, 12, (n,), generator=g)] + 0.02 * torch.randn(n, 3, generator=g)
        pts[:40] = torch.randn(40, 3, generator=g) * 30                       # far outliers: many empty shells to cross
    elif cloud == "planar":
        pts = torch.rand(n, 3, generator=g)
        pts[:, 1] = 0.25
    elif cloud == "line_with_duplicates":
        pts = torch.zeros(n, 3)
        pts[:, 0] = torch.randint(0, 500, (n,), generator=g).float() * 0.01   # ~40 coincident points per site
    elif cloud == "offset_far":
        pts = torch.rand(n, 3, generator=g) * 0.5 + torch.tensor([1000.0, -2000.0, 500.0])
    else:
        pts = torch.randn(2049, 3, generator=g)
    ref = _knn3_reference(pts)
    lib = _lib.load()
    d_pts = pts.to(device).contiguous()
    out = torch.empty(len(pts), device=device)
    _lib.check(lib.vcr_knn3_mean_dist2(len(pts), d_pts.data_ptr(), out.data_ptr(), _lib.stream_of(d_pts)))
    got = out.cpu().double()
    # fp32 differences of fp32 coordinates: relative to the squared coordinate spread, not to the (possibly zero) distance
    tol = 1e-5 * ref + 1e-6 * float((pts - pts.mean(0)).abs().max()) ** 2 * (1e-2 if cloud != "offset_far" else 1.0)
    assert bool(((got - ref).abs() <= tol + 1e-12).all()), float(((got - ref).abs() - tol).max())


def test_factorised_sh_path_trains_like_the_dense_path(device):
    """The DP exchange path (backward leaves dL/drgb, SH gradients rebuilt by vcr_sh_grad_from_rgb) run at world size 1
    must follow the same trajectory as the ordinary path."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(8000, seed=2)
    raw["scaling"] = raw["scaling"] + 1.0
    finals = []
    for force in (False, True):
        cams = synthetic.make_cameras(3, 128, 96, 110.0, device=device)
        tr = make_synthetic_trainer(raw, cams, device, preset="tnt", force_factorised=force, overlap_sh=False,
                                    optim={"densify_from_iter": 10 ** 9})
        for _ in range(12):
            tr.train_step()
        finals.append({k: getattr(tr.model, k).detach().clone() for k in ["_features_dc", "_features_rest", "_xyz", "_opacity"]})
    for k in finals[0]:
        # a dozen steps amplify fp32 atomic-order noise, and right after the opacity reset (moments zeroed) Adam moves an entry
        # by +-lr whatever the size of its gradient, so an entry whose gradient is ~0 can land 2 lr per step apart in the two
        # runs: the bound holds for all but a few per mille of the entries
        d = (finals[0][k] - finals[1][k]).abs()
        tol = 5e-3 * max(1.0, float(finals[0][k].abs().max()))
        assert float((d > tol).double().mean()) < 5e-3, (k, float(d.max()), float((d > tol).double().mean()))
        assert float(d.median()) < 0.1 * tol, (k, float(d.median()))


def test_two_stream_sh_path_trains_like_the_serial_loop(device):
    """Single-GPU default: SH Adam (gradient formed on the fly from dL/drgb x basis, vcr_sh_adam_from_rgb) and the next
    iteration's SH -> RGB evaluation run on a second stream.  Same trajectory as the serial loop with the dense SH
    gradient + vcr_adam_step, including across a densification (surgery) step and an SH-degree bump."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.gaussian_renderer import render
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(8000, seed=5)
    raw["scaling"] = raw["scaling"] + 1.0
    finals = []
    for overlap in (False, True):
        cams = synthetic.make_cameras(3, 128, 96, 110.0, device=device)
        tr = make_synthetic_trainer(raw, cams, device, preset="tnt", overlap_sh=overlap, overlap_min_gaussians=0,
                                    optim={"densify_from_iter": 10 ** 9, "opacity_reset_interval": 9})
        assert tr.overlap_sh == overlap
        tr.model.active_sh_degree = 2
        real_exchange = tr._exchange_grads

        def exchange_after_an_eval_render(ov, surgery, rec, tr=tr, cams=cams, **kw):
            # an evaluation render (with its own backward, in "rgb" mode) between this step's backward and its gradient
            # exchange: dL/drgb travels on the record of the render that produced it, so the step is unaffected
            from vcr_gaus_amd.rasterizer import RasterOptions
            if tr.current_iteration in (3, 4):
                keep = {g["name"]: g["params"][0].grad for g in tr.model.optimizer.param_groups}
                for g in tr.model.optimizer.param_groups:
                    g["params"][0].grad = None           # (autograd accumulates IN PLACE into a gradient that is already there)
                tr.join_side()
                with torch.enable_grad():
                    pkg = render(cams[2], tr.model, tr.cfg, tr.background, dirs=tr.dirs, raster_options=RasterOptions("rgb"))
                    (7.0 * pkg["render"]).sum().backward()
                assert pkg["raster"].drgb is not None and pkg["raster"] is not rec
                for g in tr.model.optimizer.param_groups:
                    g["params"][0].grad = keep[g["name"]]
            return real_exchange(ov, surgery, rec, **kw)

        tr._exchange_grads = exchange_after_an_eval_render
        for it in range(14):
            if it == 6:
                tr.model.active_sh_degree = 3
            if it == 11:
                tr.overlap_min_gaussians = 10 ** 9       # model "shrank" below the threshold: back to one stream
            tr.train_step()                              # iteration 9 resets the opacities (a surgery step)
        tr.join_side()
        torch.cuda.synchronize()
        finals.append({k: getattr(tr.model, k).detach().clone()
                       for k in ["_features_dc", "_features_rest", "_xyz", "_opacity", "_scaling", "_rotation"]})
        st = tr.model.optimizer.state
        finals[-1]["m_rest"] = st["f_rest"]["exp_avg"].clone()
        finals[-1]["v_dc"] = st["f_dc"]["exp_avg_sq"].clone()
        assert st["f_rest"]["step"] == 14 and st["xyz"]["step"] == 14
    for k in finals[0]:
        # a dozen steps amplify fp32 atomic-order noise, and right after the opacity reset (moments zeroed) Adam moves an entry
        # by +-lr whatever the size of its gradient, so an entry whose gradient is ~0 can land 2 lr per step apart in the two
        # runs: the bound holds for all but a few per mille of the entries
        d = (finals[0][k] - finals[1][k]).abs()
        tol = 5e-3 * max(1.0, float(finals[0][k].abs().max()))
        assert float((d > tol).double().mean()) < 5e-3, (k, float(d.max()), float((d > tol).double().mean()))
        assert float(d.median()) < 0.1 * tol, (k, float(d.median()))


def test_rccl_exchange_path_single_rank_group(device):
    """Runs the REAL collective branch of Trainer._allreduce_grads (all_gather_into_tensor of dL/drgb + all_reduce of the
    remaining gradients, backend 'nccl' = RCCL) inside a one-rank process group: same trajectory as without it."""
    import os
    import socket
    import torch.distributed as dist
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        raw = synthetic.make_gaussians(6000, seed=4)
        raw["scaling"] = raw["scaling"] + 1.0
        finals = []
        for coll, overlap in ((False, False), (True, False), (True, True)):
            # (True, True): the data-parallel two-stream form -- the all-gather is awaited by the second stream only and the
            # SH update (vcr_sh_adam_from_rgb_views) is issued from the next forward's colour-stream hook
            cams = synthetic.make_cameras(3, 128, 96, 110.0, device=device)
            tr = make_synthetic_trainer(raw, cams, device, preset="tnt", force_factorised=not overlap, overlap_sh=overlap,
                                        overlap_min_gaussians=0,
                                        optim={"densify_from_iter": 10 ** 9, "densify_until_iter": 100})
            tr.force_collectives = coll
            for _ in range(8):
                tr.train_step()
            assert tr.last_exchange == {(False, False): "none", (True, False): "factorised",
                                        (True, True): "factorised-deferred"}[(coll, overlap)]
            tr.join_side()
            tr.sync_densify_stats()            # (collective runs: the statistics are rank-local deltas until they are read)
            torch.cuda.synchronize()
            finals.append({k: getattr(tr.model, k).detach().clone() for k in ["_features_rest", "_xyz", "_scaling"]})
            stats = (tr.model.xyz_gradient_accum.clone(), tr.model.denom.clone(), tr.model.max_radii2D.clone())
            finals[-1]["stats"] = torch.cat([stats[0].flatten(), stats[1].flatten(), stats[2].flatten()])
        for k in finals[0]:
            assert torch.allclose(finals[0][k], finals[1][k], rtol=5e-3, atol=1e-5), k      # fp32 atomics: run-to-run noise
            assert torch.allclose(finals[0][k], finals[2][k], rtol=5e-3, atol=1e-5), k
    finally:
        dist.destroy_process_group()


def test_fused_loss_node_matches_modular_losses(device):
    """fused_losses (one autograd node) == the per-loss operators: same loss values, same training trajectory."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(6000, seed=6)
    raw["scaling"] = raw["scaling"] + 1.0
    finals, losses = [], []
    for fused in (False, True):
        cams = synthetic.make_cameras(3, 130, 94, 110.0, device=device)        # ragged size
        tr = make_synthetic_trainer(raw, cams, device, preset="dtu_c3", optim={"densify_from_iter": 10 ** 9})
        tr.use_fused_losses = fused
        for _ in range(10):
            tr.train_step()
        losses.append({k: float(v) for k, v in tr.losses.items()})
        finals.append({k: getattr(tr.model, k).detach().clone() for k in ["_features_dc", "_xyz", "_scaling", "_rotation", "_opacity"]})
    assert set(losses[0]) == set(losses[1])
    for k in losses[0]:
        assert abs(losses[0][k] - losses[1][k]) < 2e-4 * max(1.0, abs(losses[0][k])), (k, losses[0][k], losses[1][k])
    for k in finals[0]:
        # 10 steps amplify fp32 atomic-order noise, and Adam moves an entry whose gradient is ~0 by +-lr per step whatever its
        # size (0.05 for an opacity logit): the bound holds for all but a few per mille of the entries, as in the trajectory
        # tests above
        d = (finals[0][k] - finals[1][k]).abs()
        tol = 5e-3 * max(1.0, float(finals[0][k].abs().max()))
        assert float((d > tol).double().mean()) < 5e-3, (k, float(d.max()), float((d > tol).double().mean()))
        assert float(d.median()) < 0.1 * tol, (k, float(d.median()))


@pytest.mark.parametrize("n,bits,iota", [(1, 32, True), (777, 32, True), (8193, 9, False), (300001, 32, True),
                                         (1000000, 13, False), (70000, 17, False), (1000000, 27, True), (300001, 18, False),
                                         (5000000, 27, True)])
def test_radix_sort_pairs_is_stable_and_matches_torch(device, n, bits, iota):
    """vcr_sort_pairs_u32 (the rasterizer's depth / tile sort): stable, exact, ragged sizes, partial last pass."""
    import ctypes as C
    from vcr_gaus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + bits)
    if bits == 32:         # positive float bit patterns, like the depth keys (many ties in the high bytes)
        keys = (0.2 + 20.0 * torch.rand(n, generator=g)).float().view(torch.int32)
    else:
        keys = torch.randint(0, 1 << bits, (n,), generator=g, dtype=torch.int32)
    keys = keys.to(device)
    vals = None if iota else torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32).to(device)
    ko, vo = torch.empty_like(keys), torch.empty_like(keys)
    nb = lib.vcr_sort_pairs_u32_scratch_bytes(n)
    scratch = torch.empty(nb, dtype=torch.uint8, device=device)
    _lib.check(lib.vcr_sort_pairs_u32(n, keys.data_ptr(), None if iota else vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), 0, bits,
                                      scratch.data_ptr(), nb, _lib.stream_of(keys)))
    order = torch.sort(keys.cpu().long(), stable=True).indices
    assert torch.equal(ko.cpu(), keys.cpu()[order])
    ref_v = order.int() if iota else vals.cpu()[order]
    assert torch.equal(vo.cpu(), ref_v)


def test_sh_adam_from_rgb_equals_dense_gradient_plus_adam(device):
    """vcr_sh_adam_from_rgb == (vcr_sh_grad_from_rgb -> vcr_adam_step) on the SH groups, for an inactive top degree too."""
    import ctypes as C
    from vcr_gaus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    n = 5000
    xyz = torch.randn(n, 3, generator=g).to(device)
    campos = torch.tensor([[0.3, -2.0, 4.0]], device=device)
    dirs = torch.nn.functional.normalize(xyz - campos, dim=1).contiguous()
    drgb = torch.randn(n, 3, generator=g).to(device) * (torch.rand(n, 1, generator=g).to(device) > 0.3)   # some zero rows
    for deg, step in ((3, 1), (2, 7)):
        dc0, rest0 = torch.randn(n, 1, 3, generator=g).to(device), torch.randn(n, 15, 3, generator=g).to(device)
        m0 = [0.01 * torch.randn(n, 1, 3, generator=g).to(device), 0.01 * torch.randn(n, 15, 3, generator=g).to(device)]
        v0 = [1e-4 * torch.rand(n, 1, 3, generator=g).to(device), 1e-4 * torch.rand(n, 15, 3, generator=g).to(device)]
        st = _lib.stream_of(xyz)
        # reference: dense gradient, then the multi-tensor Adam kernel
        gd, gr = torch.empty(n, 1, 3, device=device), torch.empty(n, 15, 3, device=device)
        _lib.check(lib.vcr_sh_grad_from_rgb(n, deg, 1, xyz.data_ptr(), campos.data_ptr(), drgb.contiguous().data_ptr(),
                                            gd.data_ptr(), gr.data_ptr(), st))
        p = [dc0.clone(), rest0.clone()]
        m, v = [t.clone() for t in m0], [t.clone() for t in v0]
        arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
        _lib.check(lib.vcr_adam_step(2, arr(p), arr([gd, gr]), arr(m), arr(v), (C.c_int64 * 2)(3 * n, 45 * n),
                                     (C.c_float * 2)(0.0025, 0.000125), 0.9, 0.999, 1e-15, step, 1.0, st))
        # fused
        q = [dc0.clone(), rest0.clone()]
        mq, vq = [t.clone() for t in m0], [t.clone() for t in v0]
        _lib.check(lib.vcr_sh_adam_from_rgb(n, deg, dirs.data_ptr(), drgb.contiguous().data_ptr(), q[0].data_ptr(), q[1].data_ptr(),
                                            mq[0].data_ptr(), vq[0].data_ptr(), mq[1].data_ptr(), vq[1].data_ptr(), 0.0025, 0.000125,
                                            0.9, 0.999, 1e-15, step, 1.0, st))
        for a, b in zip(p + m + v, q + mq + vq):
            assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), float((a - b).abs().max())


def test_sh_adam_from_rgb_views_equals_dense_gradient_plus_adam(device):
    """Data-parallel form: vcr_sh_adam_from_rgb_views == (vcr_sh_grad_from_rgb over 3 views -> vcr_adam_step with the
    1/world gradient scale)."""
    import ctypes as C
    from vcr_gaus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(12)
    n, V = 4097, 3
    xyz = torch.randn(n, 3, generator=g).to(device)
    campos = torch.tensor([[0.3, -2.0, 4.0], [3.0, 0.5, -2.0], [-4.0, 1.0, 0.2]], device=device)
    drgb = (torch.randn(V, n, 3, generator=g) * (torch.rand(V, n, 1, generator=g) > 0.4)).to(device).contiguous()
    dc0, rest0 = torch.randn(n, 1, 3, generator=g).to(device), torch.randn(n, 15, 3, generator=g).to(device)
    m0 = [0.01 * torch.randn(n, 1, 3, generator=g).to(device), 0.01 * torch.randn(n, 15, 3, generator=g).to(device)]
    v0 = [1e-4 * torch.rand(n, 1, 3, generator=g).to(device), 1e-4 * torch.rand(n, 15, 3, generator=g).to(device)]
    st = _lib.stream_of(xyz)
    gd, gr = torch.empty(n, 1, 3, device=device), torch.empty(n, 15, 3, device=device)
    _lib.check(lib.vcr_sh_grad_from_rgb(n, 3, V, xyz.data_ptr(), campos.data_ptr(), drgb.data_ptr(), gd.data_ptr(), gr.data_ptr(), st))
    p = [dc0.clone(), rest0.clone()]
    m, v = [t.clone() for t in m0], [t.clone() for t in v0]
    arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    _lib.check(lib.vcr_adam_step(2, arr(p), arr([gd, gr]), arr(m), arr(v), (C.c_int64 * 2)(3 * n, 45 * n),
                                 (C.c_float * 2)(0.0025, 0.000125), 0.9, 0.999, 1e-15, 5, 1.0 / V, st))
    q = [dc0.clone(), rest0.clone()]
    mq, vq = [t.clone() for t in m0], [t.clone() for t in v0]
    _lib.check(lib.vcr_sh_adam_from_rgb_views(n, 3, V, xyz.data_ptr(), campos.data_ptr(), drgb.data_ptr(), q[0].data_ptr(),
                                              q[1].data_ptr(), mq[0].data_ptr(), vq[0].data_ptr(), mq[1].data_ptr(), vq[1].data_ptr(),
                                              0.0025, 0.000125, 0.9, 0.999, 1e-15, 5, 1.0 / V, st))
    for a, b in zip(p + m + v, q + mq + vq):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), float((a - b).abs().max())


def test_weighted_total(device):
    from vcr_gaus_amd import _lib
    lib = _lib.load()
    res = torch.tensor([0.1, 0.8, 0.3, 0.0, 2.0, 0.5], device=device)
    w = torch.tensor([0.8, -0.2, 100.0, 0.0, 0.015, 0.05], device=device)
    out = torch.zeros(1, device=device)
    _lib.check(lib.vcr_weighted_total(6, res.data_ptr(), w.data_ptr(), 1, out.data_ptr(), _lib.stream_of(res)))
    ref = float((res.double() * w.double()).sum() - w[1].double())
    assert abs(float(out) - ref) < 1e-5 * abs(ref)


def test_colour_stream_render_is_identical(device):
    """SH -> RGB on a second stream (VcrRasterArgs.colour_stream, with a hook that enqueues foreign work first) gives the
    same image and gradients as the single-stream call."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.rasterizer import RasterOptions
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    from vcr_gaus_amd.gaussian_renderer import render
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    raw = synthetic.make_gaussians(20000, seed=9)
    cams = synthetic.make_cameras(2, 160, 120, 140.0, device=device)
    cfg = make_config("tnt")
    m = GaussianModel(cfg.model)
    m.create_from_params(raw, spatial_lr_scale=1.0, device=device)
    m.active_sh_degree = 3
    dirs = get_all_px_dir(cams[0].intr, 120, 160)
    bg = torch.zeros(3, device=device)
    outs = []
    side = torch.cuda.Stream(device=device)
    called = []
    for two in (False, True):
        for p in (m._features_dc, m._features_rest, m._xyz):
            p.grad = None
        hook = (lambda: called.append(torch.zeros(1 << 20, device=device).add_(1.0))) if two else None
        pkg = render(cams[1], m, cfg, bg, dirs=dirs, raster_options=RasterOptions("full", side if two else None, hook))
        (pkg["render"].sum() + pkg["depth"].sum()).backward()
        torch.cuda.synchronize()
        outs.append((pkg["render_out"].detach().clone(), m._features_rest.grad.clone(), m._xyz.grad.clone()))
    assert called, "the colour-stream hook was not invoked"
    for a, b in zip(*outs):
        assert torch.equal(a, b) or torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert torch.equal(outs[0][0], outs[1][0])          # the image is bit-identical


@pytest.mark.parametrize("active", [1, 2, 3, 4, 7, 6])
def test_fused_normal_losses_match_the_modular_operators(device, active):
    """vcr_normal_losses_forward/backward == normalize + compute_normals + normal_loss x3 (values and gradients w.r.t. the
    depth and normal planes), with the camera mask and the depth threshold, on a ragged image."""
    from vcr_gaus_amd import _lib
    from vcr_gaus_amd.loss_utils import normal_loss
    from vcr_gaus_amd.normal_utils import compute_normals, normalize_rendered_normal
    lib = _lib.load()
    g = torch.Generator().manual_seed(100 + active)
    H, W = 70, 93
    P = H * W
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    depth = (2.0 + 0.01 * xx + 0.02 * yy + 0.05 * torch.rand(H, W, generator=g)).to(device)
    nrm = torch.randn(3, H, W, generator=g).to(device)
    gt = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1).to(device).contiguous()
    mask = (torch.rand(H, W, generator=g) > 0.2).to(device)
    K = torch.tensor([[80.0, 0, 46.0], [0, 82.0, 35.5], [0, 0, 1.0]])
    intr = (80.0, 82.0, 46.0, 35.5)
    depth_max, exp_t = 3.6, 0.01
    seeds = torch.tensor([0.7, 1.3, 0.4], device=device)
    # reference through the modular autograd operators
    d_ref = depth.clone().requires_grad_(True)
    n_ref = nrm.clone().requires_grad_(True)
    n = normalize_rendered_normal(n_ref)
    est = compute_normals(d_ref[None], K, intr)
    vals = [torch.zeros((), device=device)] * 3
    if active & 1:
        vals[0] = normal_loss(n, gt.view(H, W, 3))
    if active & 2:
        vals[1] = normal_loss(est, gt.view(H, W, 3), weight_src=n.detach(), exp_t=exp_t, mask=mask, depth=d_ref.detach()[None],
                              depth_max=depth_max)
    if active & 4:
        vals[2] = normal_loss(est, n)
    (seeds[0] * vals[0] + seeds[1] * vals[1] + seeds[2] * vals[2]).backward()
    # fused
    n9 = lib.vcr_sums_elems(9)
    sums = torch.zeros(n9, dtype=torch.float64, device=device)
    res = torch.zeros(3, device=device)
    st = _lib.stream_of(depth)
    m8 = mask.view(-1).to(torch.uint8).contiguous()
    _lib.check(lib.vcr_normal_losses_forward(H, W, *intr, depth.data_ptr(), nrm.data_ptr(), gt.data_ptr(), m8.data_ptr(), depth_max,
                                             exp_t, active, sums.data_ptr(), res.data_ptr(), 1, st))
    for k in range(3):
        assert abs(float(res[k]) - float(vals[k])) < 2e-5 * max(1.0, abs(float(vals[k]))), (k, float(res[k]), float(vals[k]))
    dd, dn4 = torch.empty(H, W, device=device), torch.full((4, H, W), 7.0, device=device)
    dn = dn4[:3]
    scratch = torch.empty(P * 6, device=device)
    _lib.check(lib.vcr_normal_losses_backward(H, W, *intr, depth.data_ptr(), nrm.data_ptr(), gt.data_ptr(), m8.data_ptr(), depth_max,
                                              exp_t, active, sums.data_ptr(), seeds.data_ptr(), scratch.data_ptr(), dd.data_ptr(),
                                              dn.data_ptr(), st))
    ref_dd = d_ref.grad if d_ref.grad is not None else torch.zeros_like(dd)
    ref_dn = n_ref.grad if n_ref.grad is not None else torch.zeros_like(dn)
    assert float((dd - ref_dd).abs().max()) <= 1e-4 * float(ref_dd.abs().max()) + 1e-9
    assert float((dn - ref_dn).abs().max()) <= 1e-4 * float(ref_dn.abs().max()) + 1e-9
    assert float(dn4[3].abs().max()) == 0.0          # the alpha plane behind the normal planes is zeroed


@pytest.mark.parametrize("views", [0, 3])
def test_fused_sh_update_and_colour_equals_separate_steps(device, views):
    """VcrRasterArgs.sh_update (SH Adam step fused into the colour evaluation on the colour stream; single-view and
    data-parallel multi-view form) == the stand-alone update kernel followed by an ordinary render."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.rasterizer import RasterOptions
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    from vcr_gaus_amd.gaussian_renderer import render
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    raw = synthetic.make_gaussians(12000, seed=21)
    cams = synthetic.make_cameras(3, 128, 96, 110.0, device=device)
    cfg = make_config("tnt")
    dirs = get_all_px_dir(cams[0].intr, 96, 128)
    bg = torch.zeros(3, device=device)
    g = torch.Generator().manual_seed(5)
    n = 12000
    drgb = (torch.randn(max(views, 1), n, 3, generator=g) * 0.05).to(device).contiguous()
    campos_all = torch.stack([c.camera_center for c in cams]).float().to(device).contiguous()
    results = []
    for fused in (False, True):
        m = GaussianModel(cfg.model)
        m.create_from_params(raw, spatial_lr_scale=1.0, device=device)
        m.active_sh_degree = 3
        m.training_setup(cfg.optim)
        opt = m.optimizer
        opt.grad_scale = 1.0 / max(views, 1)
        xyz0 = m._xyz.detach().clone()
        vdirs = torch.nn.functional.normalize(xyz0 - campos_all[0], dim=1).contiguous()
        side = torch.cuda.Stream(device=device)
        if fused:
            provider = (lambda: opt.make_sh_update(drgb, 3, xyz=xyz0, campos_all=campos_all)) if views else \
                (lambda: opt.make_sh_update(drgb[0], 3, view_dirs=vdirs))
            with torch.no_grad():
                out = render(cams[1], m, cfg, bg, dirs=dirs, raster_options=RasterOptions("full", side, None, provider))["render_out"]
        else:
            if views:
                opt.step_sh_from_rgb_views(drgb, xyz0, campos_all, 3)
            else:
                opt.step_sh_from_rgb(drgb[0], vdirs, 3)
            with torch.no_grad():
                out = render(cams[1], m, cfg, bg, dirs=dirs)["render_out"]
        torch.cuda.synchronize()
        results.append((out.clone(), m._features_dc.detach().clone(), m._features_rest.detach().clone(),
                        opt.state["f_rest"]["exp_avg"].clone(), opt.state["f_dc"]["exp_avg_sq"].clone()))
        assert opt.state["f_dc"]["step"] == 1 and opt.state["f_rest"]["step"] == 1
    for a, b in zip(*results):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), float((a - b).abs().max())


def test_bench_priming_restores_the_workload(device):
    """BenchTrainer.prime() runs real training steps (allocator sizes, host warm-up) and must hand back the model,
    optimizer and trainer exactly as specified, so the timed steps measure the named workload."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import BenchTrainer
    raw = synthetic.make_gaussians(5000, seed=31)
    cams = synthetic.make_cameras(3, 96, 64, 90.0, device=device)
    bt = BenchTrainer(raw, cams, device)
    m, tr = bt.tr.model, bt.tr
    before = {k: getattr(m, k).detach().clone() for k in ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"]}
    it0, order0, rng0 = tr.current_iteration, list(tr.view_order), tr.rng.getstate()
    bt.prime(min_seconds=0.05)
    assert tr.current_iteration == it0 and tr.view_order == order0 and tr.rng.getstate() == rng0
    assert m.optimizer.state == {} and tr._pending_sh is None
    for k, v in before.items():
        assert torch.equal(getattr(m, k).detach(), v), k
    bt.step(0)                                      # and the trainer still steps
    assert float(tr.losses["total"]) == float(tr.losses["total"])


def test_misc_regularisers_vs_reference_vectors(device):
    """edge-aware distortion mean, normal-curvature loss and opacity entropy (`trainer.py:247-249,282-303`): HIP forward and
    backward against values / gradients produced by the reference's own functions (g5 / g7 fixtures)."""
    from vcr_gaus_amd.loss_utils import curv_loss, edge_aware_mean, entropy_regulariser, normal2curv
    g5, g7 = load("g5_misc.npz"), load("g7_misc_grads.npz")
    T = lambda a: torch.from_numpy(np.asarray(a)).to(device)
    # forward maps of g5 (the map form used for visualisation)
    assert torch.allclose(normal2curv(T(g5["nrm"]), T(g5["mask"])), T(g5["curv"]), atol=1e-6)
    # edge-aware mean + gradient to the map
    dist = T(g7["dist"]).clone().requires_grad_(True)
    l = edge_aware_mean(T(g7["img"]), dist)
    l.backward()
    assert abs(float(l) - float(g7["edge_loss"])) < 1e-6 * max(1.0, abs(float(g7["edge_loss"])))
    assert torch.allclose(dist.grad, T(g7["edge_grad"]), rtol=1e-5, atol=1e-9)
    # curvature loss + gradient to the normal map (sign function: compare away from exact zeros)
    nrm = T(g7["nrm"]).clone().requires_grad_(True)
    l = curv_loss(nrm, T(g7["mask"]))
    l.backward()
    assert abs(float(l) - float(g7["curv_loss"])) < 2e-6 * max(1.0, abs(float(g7["curv_loss"])))
    assert torch.allclose(nrm.grad, T(g7["curv_grad"]), rtol=1e-5, atol=1e-9)
    # entropy on raw opacities without the box mask == reference entropy_loss(sigmoid(raw)); chain rule for the gradient
    op = T(g7["op"]).double()
    raw = torch.log(op / (1 - op)).float().requires_grad_(True)
    l = entropy_regulariser(raw)
    l.backward()
    assert abs(float(l) - float(g7["entropy_loss"])) < 1e-5
    want = T(g7["entropy_grad"]).double() * op * (1 - op)
    assert torch.allclose(raw.grad.double(), want, rtol=2e-4, atol=1e-8)
    # with the bounding-box mask: only inside Gaussians count
    xyz = torch.randn(raw.shape[0], 3, device=device)
    trans, scale = torch.tensor([0.1, -0.2, 0.3], device=device), torch.tensor([1.5, 1.0, 0.8], device=device)
    inside = (((xyz - trans) / scale).abs() < 1).all(-1)
    l2 = entropy_regulariser(raw.detach().requires_grad_(True), xyz, trans, scale)
    p = torch.sigmoid(raw.detach()[inside]).double()
    ref = (-p * torch.log(p + 1e-6) - (1 - p) * torch.log(1 - p + 1e-6)).mean()
    assert abs(float(l2) - float(ref)) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tsdf_depth_input_vs_reference_vectors(device, tag):
    """depth2point + the depth masking of tsdf_fusion (`tools/graphics_utils.py:134-141`, `tools/depth2mesh.py:37-52`)
    against the reference's own outputs (g8 fixture)."""
    from vcr_gaus_amd import depth2mesh
    g = load("g8_tsdf_input.npz")
    K, w2c = g[f"{tag}_K"], g[f"{tag}_w2c"]
    depth = g[f"{tag}_depth"].to(device)
    cam, wld = depth2mesh.depth2point(depth[0], K, w2c)
    assert torch.allclose(cam.cpu(), g[f"{tag}_xyz_cam"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(wld.cpu(), g[f"{tag}_xyz_world"], rtol=1e-5, atol=2e-6)

    class V:
        intr_scalars = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
        world_view_transform = w2c.t().contiguous()
        gt_alpha_mask = g[f"{tag}_gt_alpha"].to(device)

    class M:
        trans, scale = g[f"{tag}_trans"].to(device), g[f"{tag}_scale"].to(device)

    out = depth2mesh.tsdf_depth_input({"depth": depth, "alpha": g[f"{tag}_alpha"].to(device)}, V, M, alpha_thres=0.5)
    want = g[f"{tag}_masked"]
    assert out.shape == want.shape
    diff = (out.cpu() != want)
    assert float(diff.double().mean()) < 2e-3            # a point within one ulp of the box face may fall on the other side
    assert 0.2 < float((want == 0).double().mean()) < 0.98
    assert torch.allclose(out.cpu()[~diff], want[~diff])


def test_get_covariance_and_checkpoint_roundtrip(device):
    """`GaussianModel.get_covariance` (`scene/gaussian_model.py:38-42,194`) vs the oracle's Sigma = (R S)(R S)^T, and
    `capture` / `restore` (`:88-123`) reproducing parameters, Adam moments, statistics and learning rates."""
    from oracle import raster_torch as OR
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    cfg = make_config("tnt")
    raw = synthetic.make_gaussians(777, seed=9)
    m = GaussianModel(cfg.model)
    m.create_from_params(raw, 2.5, device=device)
    m.training_setup(cfg.optim)
    cov = m.get_covariance(1.3).detach().cpu().double()
    q = torch.nn.functional.normalize(raw["rotation"].double())
    S3 = OR.cov3d_from_scale_rot(torch.exp(raw["scaling"].double()), 1.3, q)
    want = torch.stack([S3[:, 0, 0], S3[:, 0, 1], S3[:, 0, 2], S3[:, 1, 1], S3[:, 1, 2], S3[:, 2, 2]], 1)
    assert torch.allclose(cov, want, rtol=1e-5, atol=2e-6 * float(want.abs().max()))     # fp32 products of the rotation
    # a few Adam steps so that the optimizer state is non-trivial
    for i in range(3):
        for g in m.optimizer.param_groups:
            g["params"][0].grad = torch.randn_like(g["params"][0]) * 1e-3
        m.update_learning_rate(100 * (i + 1))
        m.optimizer.step()
    m.xyz_gradient_accum += 0.5
    m.denom += 2
    m.max_radii2D += 7
    m.active_sh_degree = 2
    import copy
    snap = copy.deepcopy(m.capture())          # (what torch.save / torch.load of the checkpoint tuple does)
    before = {k: getattr(m, k).detach().clone() for k in ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"]}
    m2 = GaussianModel(cfg.model)
    m2.restore(snap, cfg.optim)
    assert m2.active_sh_degree == 2 and m2.spatial_lr_scale == 2.5
    for k, v in before.items():
        assert torch.equal(getattr(m2, k).detach(), v)
    assert torch.equal(m2.xyz_gradient_accum, m.xyz_gradient_accum) and torch.equal(m2.denom, m.denom)
    assert torch.equal(m2.max_radii2D, m.max_radii2D)
    for name, st in m.optimizer.state.items():
        st2 = m2.optimizer.state[name]
        assert st2["step"] == st["step"] and torch.equal(st2["exp_avg"], st["exp_avg"]) and torch.equal(st2["exp_avg_sq"], st["exp_avg_sq"])
    assert [g["lr"] for g in m2.optimizer.param_groups] == [g["lr"] for g in m.optimizer.param_groups]
    # and the restored model keeps training identically
    for mm in (m, m2):
        torch.manual_seed(3)
        for g in mm.optimizer.param_groups:
            g["params"][0].grad = torch.randn_like(g["params"][0]) * 1e-3
        mm.optimizer.step()
    assert torch.equal(m._xyz, m2._xyz) and torch.equal(m._features_rest, m2._features_rest)


def test_ply_roundtrip_on_device(device, tmp_path):
    """save_ply / load_ply (`scene/gaussian_model.py:272-320,366-423`) with the model on the GPU: same field order as the
    reference's point_cloud.ply and a bit-exact parameter round trip."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    cfg = make_config("tnt")
    raw = synthetic.make_gaussians(513, seed=10, sem_channels=2)
    m = GaussianModel(cfg.model)
    m.create_from_params(raw, 1.0, device=device)
    path = str(tmp_path / "point_cloud.ply")
    m.save_ply(path)
    head = open(path, "rb").read(4096).split(b"end_header")[0].decode()
    props = [l.split()[-1] for l in head.splitlines() if l.startswith("property")]
    assert props[:6] == ["x", "y", "z", "nx", "ny", "nz"] and props[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[9 + 45:9 + 45 + 8] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert props[-2:] == ["obj_dc_0", "obj_dc_1"]
    m2 = GaussianModel(cfg.model)
    m2.load_ply(path, device=device)
    for k in ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_objects_dc"]:
        assert torch.equal(getattr(m, k).detach(), getattr(m2, k).detach()), k


def test_semantic_classifier_is_trained(device):
    """`scene/gaussian_model.py:254`: the 1x1-conv classifier is an Adam group at cls_lr: with the semantic
    loss on, its weights move, its gradients are released after the step, and densify / prune leave the group alone."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(3000, seed=12, sem_channels=2)
    raw["scaling"] = raw["scaling"] + 1.5
    cams = synthetic.make_cameras(3, 96, 64, 80.0, device=device)
    tr = make_synthetic_trainer(raw, cams, device, preset="tnt", overlap_sh=False,
                                optim={"loss_weight": {"semantic": 0.005}, "densify_from_iter": 2, "densification_interval": 2,
                                       "densify_until_iter": 100, "prune": {"iterations": []}})
    m = tr.model
    names = [g["name"] for g in m.optimizer.param_groups]
    assert "classifier.weight" in names and "classifier.bias" in names
    w0, b0 = m.classifier.weight.detach().clone(), m.classifier.bias.detach().clone()
    n0 = m._xyz.shape[0]
    for _ in range(5):
        tr.train_step()
    torch.cuda.synchronize()
    assert "semantic" in tr.losses and float(tr.losses["semantic"]) > 0
    assert float((m.classifier.weight - w0).abs().max()) > 0 and float((m.classifier.bias - b0).abs().max()) > 0
    assert m.classifier.weight.grad is None and m.classifier.bias.grad is None
    assert m._xyz.shape[0] != n0                                      # densification ran with the aux groups present
    st = m.optimizer.state["classifier.weight"]
    assert st["step"] == 5 and st["exp_avg"].shape == m.classifier.weight.shape
    assert any(g["params"][0] is m.classifier.weight for g in m.optimizer.param_groups)


def test_convert_shs_python_path_equals_native_sh(device):
    """`pipline.convert_SHs_python` (`gaussian_renderer/__init__.py:81-87`): colours from `eval_sh` in Python handed over as
    colors_precomp give the render of the native SH path, and the SH coefficients receive the same gradients."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.config import make_config
    from vcr_gaus_amd.gaussian_model import GaussianModel
    from vcr_gaus_amd.gaussian_renderer import render
    from vcr_gaus_amd.graphics_utils import get_all_px_dir
    raw = synthetic.make_gaussians(3000, seed=14)
    raw["scaling"] = raw["scaling"] + 1.8
    cam = synthetic.make_cameras(3, 96, 64, 80.0, device=device)[1]
    dirs = get_all_px_dir(cam.intr, 64, 96)
    outs, grads = [], []
    for py in (False, True):
        cfg = make_config("tnt")
        cfg.pipline.convert_SHs_python = py
        m = GaussianModel(cfg.model)
        m.create_from_params(raw, 1.0, device=device)
        m.active_sh_degree = 2
        pkg = render(cam, m, cfg, torch.tensor([0.1, 0.2, 0.3], device=device), dirs=dirs)
        (pkg["render"] * torch.linspace(0.5, 1.5, 96, device=device)).sum().backward()
        outs.append(pkg["render"].detach()); grads.append((m._features_dc.grad.clone(), m._features_rest.grad.clone(), m._xyz.grad.clone()))
    assert torch.allclose(outs[0], outs[1], atol=2e-6)
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9


@pytest.mark.parametrize("S,K", [(2, 2), (3, 5), (4, 8)])
def test_fused_semantic_loss_equals_conv_plus_cross_entropy(device, S, K):
    """`semantic_loss` = the reference's `F.cross_entropy(classifier(sem)...) / log(num_cls)`
    (`gaussian_renderer/__init__.py:146-148`, `trainer.py:304-307`): value and gradients to the feature planes, the 1x1-conv
    weight and the bias against the same torch ops (which ARE the reference's implementation)."""
    from vcr_gaus_amd.loss_utils import semantic_loss
    g = torch.Generator().manual_seed(S * 10 + K)
    H, W = 37, 53
    sem0 = torch.randn(S, H, W, generator=g)
    lab = torch.randint(0, K, (H, W), generator=g)
    cls = torch.nn.Conv2d(S, K, kernel_size=1)
    res = []
    for fused in (False, True):
        c = torch.nn.Conv2d(S, K, kernel_size=1).to(device)
        c.load_state_dict(cls.state_dict())
        sem = sem0.to(device).clone().requires_grad_(True)
        if fused:
            loss = semantic_loss(sem, c, lab.to(device))
        else:
            logits = c(sem[None])[0].permute(1, 2, 0)
            loss = torch.nn.functional.cross_entropy(logits.reshape(-1, K), lab.to(device).view(-1)) / torch.log(torch.tensor(float(K)))
        loss.backward()
        res.append((float(loss), sem.grad.clone(), c.weight.grad.clone(), c.bias.grad.clone()))
    assert abs(res[0][0] - res[1][0]) < 2e-6 * max(1.0, abs(res[0][0]))
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-8), float((a - b).abs().max())
    # a label outside [0, K) is refused, as F.cross_entropy refuses it (the kernel alone would count it as zero loss)
    bad = lab.clone()
    bad[3, 4] = K
    with pytest.raises(ValueError, match="labels span"):
        semantic_loss(sem0.to(device), c, bad.to(device))


def test_batched_visibility_equals_the_per_camera_passes(device):
    """`vcr_visibility_batch` (B cameras in one call: geometry-only projection, several cameras in flight on internal
    streams, buffer sets re-used, counters accumulated on the device) against the reference's form -- one f_count = 3 render
    per camera, `countlist`s summed (`tools/prune.py:51-69`): identical integers, for every number of cameras in flight; the
    flag form equals `sum > 0`; cameras of two resolutions in one list; and the sum against the oracle's counts."""
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.gaussian_renderer import visi_acc_render, visibility_counts
    from vcr_gaus_amd.rasterizer import visibility_batch
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(6000, seed=3)
    raw["scaling"] = raw["scaling"] + 1.1
    cams = synthetic.make_cameras(21, 112, 80, 95.0, device=device) + synthetic.make_cameras(5, 64, 96, 70.0, radius=2.2, device=device)
    tr = make_synthetic_trainer(raw, cams[:3], device)
    m, pipe = tr.model, tr.cfg.pipline
    ref = torch.zeros(6000, dtype=torch.int32, device=device)
    per_cam_R = []
    for cam in cams:
        pkg = visi_acc_render(cam, m, pipe, tr.background)
        ref += pkg["countlist"]
    assert int((ref > 0).sum()) > 1000
    got = visibility_counts(cams, m, pipe)
    assert got.dtype == torch.int32 and torch.equal(got, ref)
    flags = visibility_counts(cams, m, pipe, flags_only=True)
    assert torch.equal(flags, (ref > 0).int())
    # accumulation into a caller's tensor, and every in-flight depth (1 set ... more sets than cameras)
    from vcr_gaus_amd.gaussian_renderer import fused_activate, _cam_rotation
    import math as _m
    sc, ro, op = fused_activate(m, cams[0].camera_center, _cam_rotation(cams[0], device), False)
    big = cams[:21]
    vm = torch.stack([c.world_view_transform for c in big]); pm = torch.stack([c.full_proj_transform for c in big])
    cc = torch.stack([c.camera_center for c in big])
    tx, ty = [_m.tan(c.FoVx * 0.5) for c in big], [_m.tan(c.FoVy * 0.5) for c in big]
    want = torch.zeros_like(ref)
    for cam in big:
        want += visi_acc_render(cam, m, pipe, tr.background)["countlist"]
    for inflight in (1, 3, 4, 8, 16, 64):
        acc = torch.full((6000,), 7, dtype=torch.int32, device=device)
        cnt, R, V = visibility_batch(vm, pm, cc, tx, ty, 80, 112, m.get_xyz, op, sc, ro, inflight=inflight, count=acc)
        assert cnt is acc and torch.equal(acc - 7, want), inflight
        assert len(R) == 21 and all(r > 0 for r in R) and all(0 < v <= 6000 for v in V)
    # the plain forward accepts the flag mode too (f_count = 4: count SET to 1)
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = cams[2]
    rs = GaussianRasterizationSettings(image_height=80, image_width=112, tanfovx=_m.tan(cam.FoVx * 0.5), tanfovy=_m.tan(cam.FoVy * 0.5),
                                       bg=tr.background, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                       projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center, f_count=4)
    c3, _ = GaussianRasterizer(rs)(means3D=m.get_xyz, means2D=None, opacities=op, shs=m._features_dc, shs_rest=m._features_rest,
                                   scales=sc, rotations=ro)
    # against the oracle (single camera: the counts of f_count = 3 are those of the count render)
    from tests import util
    cam_c, inp, dirs = util.make_case(3000, 96, 64, 80.0, seed=9, scale_mult=6.0)
    (rc, _, _, _, _), _ = util.oracle_forward(cam_c, inp, dirs, torch.zeros(3), f_count=1, use_normals=False)
    mv = lambda t: t.float().to(device)
    cnt, _, _ = visibility_batch(mv(cam_c.world_view_transform)[None], mv(cam_c.full_proj_transform)[None], mv(cam_c.camera_center)[None],
                                 [_m.tan(cam_c.FoVx * 0.5)], [_m.tan(cam_c.FoVy * 0.5)], 64, 96, mv(inp["means3D"]), mv(inp["opac"]),
                                 mv(inp["scales"]), mv(inp["rots"]))
    assert float((cnt.cpu() != rc).double().mean()) < 1e-3
    assert int(c3.max()) == 1 and torch.equal(c3 > 0, visi_acc_render(cam, m, pipe, tr.background)["countlist"] > 0)
    # the trainer's own cameras (`sample_cameras`: rows of one stacked tensor, handed to the library without re-stacking)
    from vcr_gaus_amd.camera_utils import sample_cameras
    vc = sample_cameras(12, m.trans, m.scale, device=device, generator=torch.Generator().manual_seed(5), size=96, fov=1.2)
    assert vc[0]._stack[0].is_cuda and vc[3].world_view_transform.data_ptr() == vc[0]._stack[0][3].data_ptr()
    want = torch.zeros_like(ref)
    for c in vc:
        want += visi_acc_render(c, m, pipe, tr.background)["countlist"]
    assert int((want > 0).sum()) > 100
    assert torch.equal(visibility_counts(vc, m, pipe), want)
    assert torch.equal(visibility_counts(vc[::-1], m, pipe), want)            # (not the stack's order: re-stacked)
    assert torch.equal(tr.visibility_mask(vc), (want > 0) & m.get_inside_gaus_normalized()[0])
