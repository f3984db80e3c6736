"""Densification / pruning / opacity reset against fixtures produced by the REFERENCE's own GaussianModel
(`scene/gaussian_model.py:361-364,425-671`, executed on the CPU by tests/golden/make_golden.py::densify_cases):
parameters, Adam moments and densification statistics after every operation.  The CPU test pins the host logic; the
`gpu` test runs the same operations on the device (HIP compaction / statistics kernels)."""
import os

import numpy as np
import pytest
import torch

from vcr_gaus_amd.config import make_config
from vcr_gaus_amd.gaussian_model import GaussianModel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_densify.npz"))
NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "obj_dc"]
EXTENT = float(G["extent"])


def model_from(prefix, device):
    cfg = make_config("tnt", optim={"percent_dense": 0.01, "densify_large": {"percent_dense": 2e-3}})
    cfg.model.enable_semantic, cfg.model.ch_sem_feat, cfg.model.num_cls = True, 2, 2
    m = GaussianModel(cfg.model)
    m.create_from_params({k: torch.from_numpy(G[f"{prefix}_{k}"]) for k in NAMES}, 3.0, device=device)
    m.extent = EXTENT
    m.training_setup(cfg.optim)
    for g in m.optimizer.param_groups:
        if g.get("aux"):
            continue
        m.optimizer.state[g["name"]] = dict(step=7, exp_avg=torch.from_numpy(G[f"{prefix}_{g['name']}_m"]).to(device),
                                            exp_avg_sq=torch.from_numpy(G[f"{prefix}_{g['name']}_v"]).to(device))
    m.xyz_gradient_accum = torch.from_numpy(G[f"{prefix}_accum"]).to(device)
    m.denom = torch.from_numpy(G[f"{prefix}_denom"]).to(device)
    m.max_radii2D = torch.from_numpy(G[f"{prefix}_radii"]).to(device)
    return m


def check(m, tag, exact=True):
    tab = m._param_table()
    for name in NAMES:
        got = getattr(m, tab[name]).detach().cpu()
        want = torch.from_numpy(G[f"{tag}_{name}"])
        assert got.shape == want.shape, (tag, name, got.shape, want.shape)
        # gathered values are bit-exact; the split's new means / log-scales are fp32 arithmetic (rounding of exp/log)
        assert torch.allclose(got, want, rtol=0 if exact and name not in ("xyz", "scaling", "opacity") else 2e-6, atol=1e-7), (tag, name)
        st = m.optimizer.state[name]
        assert torch.equal(st["exp_avg"].cpu(), torch.from_numpy(G[f"{tag}_{name}_m"])), (tag, name, "exp_avg")
        assert torch.equal(st["exp_avg_sq"].cpu(), torch.from_numpy(G[f"{tag}_{name}_v"])), (tag, name, "exp_avg_sq")
        assert m.optimizer.param_groups[[g["name"] for g in m.optimizer.param_groups].index(name)]["params"][0] is getattr(m, tab[name])
    assert torch.allclose(m.xyz_gradient_accum.cpu(), torch.from_numpy(G[f"{tag}_accum"]), rtol=1e-6, atol=1e-9), tag
    assert torch.equal(m.denom.cpu(), torch.from_numpy(G[f"{tag}_denom"])), tag
    assert torch.equal(m.max_radii2D.cpu(), torch.from_numpy(G[f"{tag}_radii"])), tag


def grads_of(m):
    g = m.xyz_gradient_accum / m.denom
    g[g.isnan()] = 0.0
    return g


def run_ops(device):
    m = model_from("start", device)
    m.densify_and_clone(grads_of(m), 5e-4, EXTENT)
    check(m, "clone")
    m = model_from("start", device)
    m.densify_and_split_along_maxscaling(grads_of(m), 5e-4, EXTENT)
    check(m, "split")
    m = model_from("start", device)
    m.densify_and_split_along_maxscaling(grads_of(m), 5e-4, EXTENT, visi=torch.from_numpy(G["split_visi_mask"]).to(device))
    check(m, "split_visi")
    m = model_from("start", device)
    m.prune_points(torch.from_numpy(G["prune_mask"]).to(device))
    check(m, "prune")
    m = model_from("start", device)
    m.reset_opacity()
    check(m, "reset")
    m = model_from("start", device)
    m.prune_gaussians(0.3, torch.from_numpy(G["prune_gaussians_score"]).to(device))
    check(m, "prune_gaussians")
    m = model_from("start", device)
    m.densify_and_prune(5e-4, 0.005, EXTENT, None, torch.from_numpy(G["dap_visi"]).to(device))
    check(m, "densify_and_prune")
    m = model_from("start", device)
    m.densify_and_prune(5e-4, 0.005, EXTENT, 20, torch.from_numpy(G["daps_visi"]).to(device))
    check(m, "densify_and_prune_sized")


def test_fixture_is_non_trivial():
    n0 = G["start_xyz"].shape[0]
    assert G["clone_xyz"].shape[0] > n0 and G["split_xyz"].shape[0] > n0 and G["split_visi_xyz"].shape[0] < G["split_xyz"].shape[0]
    assert G["densify_and_prune_sized_xyz"].shape[0] < G["densify_and_prune_xyz"].shape[0]
    assert (G["start_denom"] == 0).any()             # NaN -> 0 branch of `:644-645` is exercised


def test_densify_prune_reset_match_reference_cpu():
    run_ops(torch.device("cpu"))


def test_densification_stats_match_reference_cpu():
    m = model_from("stats_in", torch.device("cpu"))
    vp = torch.zeros(G["stats_vpgrad"].shape)
    vp.grad = torch.from_numpy(G["stats_vpgrad"])
    radii = torch.from_numpy(G["stats_radii"])
    vis = radii > 0
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis].float())       # `trainer.py:345`
    m.add_densification_stats(vp, vis)
    check(m, "stats_out")


@pytest.mark.gpu
def test_densify_prune_reset_match_reference_gpu(device):
    run_ops(device)


@pytest.mark.gpu
def test_densification_stats_kernel_matches_reference(device):
    """vcr_densify_stats (accum += |grad_xy|, denom += 1, max_radii2D = max on radii > 0) vs `scene/gaussian_model.py:669-671`."""
    m = model_from("stats_in", device)
    vp = torch.zeros(G["stats_vpgrad"].shape, device=device)
    vp.grad = torch.from_numpy(G["stats_vpgrad"]).to(device)
    m.add_densification_stats(vp, None, radii=torch.from_numpy(G["stats_radii"]).to(device))
    check(m, "stats_out")


def test_ply_bytes_match_the_table_the_reference_writes(tmp_path):
    """`save_ply` (`scene/gaussian_model.py:272-311`): same property names in the same order and, row by row, the same
    float32 vertex table the reference hands to plyfile (captured by make_golden.py); `load_ply` restores the parameters."""
    m = model_from("ply", torch.device("cpu"))
    path = str(tmp_path / "pc" / "point_cloud.ply")
    m.save_ply(path)
    blob = open(path, "rb").read()
    head, payload = blob.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == f"element vertex {G['ply_table'].shape[0]}"
    names = [l.split()[-1] for l in lines if l.startswith("property")]
    assert names == [str(n) for n in G["ply_names"]]
    assert all(l.split()[1] == "float" for l in lines if l.startswith("property"))
    table = np.frombuffer(payload, dtype="<f4").reshape(G["ply_table"].shape)
    assert np.array_equal(table, G["ply_table"])
    cfg = make_config("tnt")
    cfg.model.enable_semantic, cfg.model.ch_sem_feat, cfg.model.num_cls = True, 2, 2
    m2 = GaussianModel(cfg.model)
    m2.load_ply(path, device="cpu")
    tab = m._param_table()
    for name in NAMES:
        assert torch.equal(getattr(m2, tab[name]).detach(), getattr(m, tab[name]).detach()), name
