"""World-size-2 data-parallel plumbing on CPU (gloo): camera partitioning, gradient all-reduce and the
densification-statistics reduction of vcr_gaus_amd.trainer.Trainer must reproduce what a single process
gets when it accumulates the same two views (SURVEY.md 8e semantics caveat).  The render / HIP kernels are
replaced by a deterministic per-view stub here: only the distributed logic is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vcr_gaus_amd.config import make_config
from vcr_gaus_amd.trainer import Trainer

N = 257


class StubOptim:
    def __init__(self, params):
        self.param_groups = [{"params": [p], "lr": 0.1, "name": str(i)} for i, p in enumerate(params)]
        self.grad_scale = None


class StubModel:
    """Holds parameters + densification buffers; statistics use the reference's masked torch ops."""

    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self._xyz = torch.nn.Parameter(torch.randn(N, 3, generator=g))
        self._opacity = torch.nn.Parameter(torch.randn(N, 1, generator=g))
        self.optimizer = StubOptim([self._xyz, self._opacity])
        self.xyz_gradient_accum = torch.zeros(N, 1)
        self.denom = torch.zeros(N, 1)
        self.max_radii2D = torch.zeros(N)
        self.extent = 1.0

    def add_densification_stats(self, vp, update_filter, radii=None):
        f = radii > 0
        self.xyz_gradient_accum[f] += torch.norm(vp.grad[f, :2], dim=-1, keepdim=True)
        self.denom[f] += 1
        self.max_radii2D[f] = torch.max(self.max_radii2D[f], radii[f].float())


def view_data(view, model):
    """Deterministic stand-in for render+loss+backward of camera `view`."""
    g = torch.Generator().manual_seed(100 + view)
    model._xyz.grad = torch.randn(N, 3, generator=g)
    model._opacity.grad = torch.randn(N, 1, generator=g)
    vp = torch.zeros(N, 3)
    vp.grad = torch.randn(N, 3, generator=g)
    radii = (torch.rand(N, generator=g) > 0.4).int() * torch.randint(1, 30, (N,), generator=g, dtype=torch.int32)
    return {"viewspace_points_densify": vp, "visibility_filter": radii > 0, "radii": radii}


def make_trainer(world, rank):
    cfg = make_config("tnt")
    tr = Trainer(cfg, StubModel(), list(range(8)), 1.0, torch.device("cpu"), world=world, rank=rank, seed=3)
    tr.factorised_sh = False          # the stub model has no SH groups; the factorised exchange is tested below
    return tr


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = make_trainer(world, rank)
    picks = []
    for step in range(3):
        cams = tr._next_cameras()
        picks.append(cams)
        data = view_data(cams[rank], tr.model)
        tr._allreduce_grads()
        tr._densify_stats(data)                    # rank-local deltas, no collective ...
        if step == 1:
            tr.sync_densify_stats()                # ... folded in on demand (here once mid-way and once at the end)
        if step < 2:
            tr.model._xyz.grad = None
    local_only = tr.model.denom.clone()
    tr.sync_densify_stats()
    assert tr._stats_delta is None and not torch.equal(local_only, tr.model.denom)
    tr.sync_densify_stats()                        # idempotent once clean
    if rank == 0:
        torch.save(dict(picks=picks, gx=tr.model._xyz.grad, go=tr.model._opacity.grad, acc=tr.model.xyz_gradient_accum,
                        den=tr.model.denom, mr=tr.model.max_radii2D, scale=tr.model.optimizer.grad_scale), out)
    dist.destroy_process_group()


def test_two_rank_dp_matches_single_process_accumulation(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single process: same seeded camera order, two views per step accumulated by hand
    tr = make_trainer(1, 0)
    order = make_trainer(2, 0)
    for step in range(3):
        cams = order._next_cameras()
        assert cams == got["picks"][step] and len(set(cams)) == 2
        gx = go = None
        for v in cams:
            data = view_data(v, tr.model)
            gx = tr.model._xyz.grad.clone() if gx is None else gx + tr.model._xyz.grad
            go = tr.model._opacity.grad.clone() if go is None else go + tr.model._opacity.grad
            tr.model.add_densification_stats(data["viewspace_points_densify"], data["visibility_filter"], radii=data["radii"])
    assert got["scale"] == 0.5                      # mean over the two views is applied inside the Adam kernel
    assert torch.allclose(got["gx"], gx) and torch.allclose(got["go"], go)
    assert torch.allclose(got["acc"], tr.model.xyz_gradient_accum, atol=1e-6)
    assert torch.equal(got["den"], tr.model.denom) and torch.equal(got["mr"], tr.model.max_radii2D)


def test_camera_batches_cover_every_view_once_per_epoch():
    tr = make_trainer(4, 1)
    seen = []
    for _ in range(2):
        seen += tr._next_cameras()
    assert sorted(seen) == list(range(8))


def _basis_outer(xyz, campos, drgb, deg):
    """torch stand-in for vcr_sh_grad_from_rgb: d/dshs of sum(eval_sh(shs, dir) * drgb) = basis_k(dir) x drgb."""
    from oracle import raster_torch as OR
    d = xyz - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    shs = torch.zeros(xyz.shape[0], 16, 3, requires_grad=True)
    (OR.eval_sh(deg, shs, d) * drgb).sum().backward()
    return shs.grad


class _Cam:
    def __init__(self, i):
        self.camera_center = torch.tensor([1.0 + i, -0.5 * i, 2.0])


def sh_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcr_gaus_amd.rasterizer import RasterRecord
    cfg = make_config("tnt")
    m = StubModel()
    m._features_dc = torch.nn.Parameter(torch.zeros(N, 1, 3))
    m._features_rest = torch.nn.Parameter(torch.zeros(N, 15, 3))
    m.active_sh_degree = 3
    m.optimizer.param_groups += [{"params": [m._features_dc], "lr": 0.1, "name": "f_dc"},
                                 {"params": [m._features_rest], "lr": 0.1, "name": "f_rest"}]
    tr = Trainer(cfg, m, [_Cam(i) for i in range(8)], 1.0, torch.device("cpu"), world=world, rank=rank, seed=3)
    assert tr.factorised_sh

    def rebuild(drgb_all, campos_all):
        g = sum(_basis_outer(m._xyz.detach(), campos_all[v], drgb_all[v], 3) for v in range(drgb_all.shape[0]))
        return g[:, :1].contiguous(), g[:, 1:].contiguous()

    tr._sh_grads_from_rgb = rebuild
    cams = tr._next_cameras()
    view_data(cams[rank], m)
    gen = torch.Generator().manual_seed(500 + cams[rank])
    rec = RasterRecord()                                            # what this rank's backward leaves on ITS render's record
    rec.drgb, rec.view_dirs = torch.randn(N, 3, generator=gen), torch.zeros(N, 3)
    tr._allreduce_grads(rec=rec)
    assert rec.drgb is None
    if rank == 0:
        torch.save(dict(cams=cams, dc=m._features_dc.grad, rest=m._features_rest.grad, gx=m._xyz.grad), out)
    dist.destroy_process_group()


def test_factorised_sh_exchange_two_ranks(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sh.pt")
    mp.spawn(sh_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    m = StubModel()
    ref = 0
    gx = 0
    for v in got["cams"]:
        view_data(v, m)
        gx = gx + m._xyz.grad
        drgb = torch.randn(N, 3, generator=torch.Generator().manual_seed(500 + v))
        ref = ref + _basis_outer(m._xyz.detach(), _Cam(v).camera_center, drgb, 3)
    assert torch.allclose(got["dc"], ref[:, :1], atol=1e-6) and torch.allclose(got["rest"], ref[:, 1:], atol=1e-6)
    assert torch.allclose(got["gx"], gx)


# ---- the two-stream ("deferred SH") data-parallel branch: `Trainer._exchange_grads(overlap=True, surgery=False)` ----------
class _ShStubOptim(StubOptim):
    """Records what the deferred SH update receives; applies a plain SGD step so that replicas can be compared."""

    def __init__(self, params):
        super().__init__(params)
        self.state = {}
        self.sh_calls = []

    def _state(self, g):
        return self.state.setdefault(g["name"], dict(step=0))

    def step_sh_from_rgb_views(self, drgb_all, xyz, campos_all, deg, stream=None):
        g = sum(_basis_outer(xyz, campos_all[v], drgb_all[v], deg) for v in range(drgb_all.shape[0])) * self.grad_scale
        self.sh_calls.append(g)
        by = {q["name"]: q["params"][0] for q in self.param_groups}
        with torch.no_grad():
            by["f_dc"].sub_(0.1 * g[:, :1])
            by["f_rest"].sub_(0.1 * g[:, 1:])

    def step(self):
        with torch.no_grad():
            for g in self.param_groups:
                p = g["params"][0]
                if p.grad is not None:
                    p.sub_(0.1 * self.grad_scale * p.grad)


def _sh_model():
    m = StubModel()
    m._features_dc = torch.nn.Parameter(torch.zeros(N, 1, 3))
    m._features_rest = torch.nn.Parameter(torch.zeros(N, 15, 3))
    m.active_sh_degree = 3
    m.optimizer = _ShStubOptim([m._xyz, m._opacity])
    m.optimizer.param_groups += [{"params": [m._features_dc], "lr": 0.1, "name": "f_dc"},
                                 {"params": [m._features_rest], "lr": 0.1, "name": "f_rest"}]
    return m


def deferred_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcr_gaus_amd.rasterizer import RasterRecord
    m = _sh_model()
    tr = Trainer(make_config("tnt"), m, [_Cam(i) for i in range(8)], 1.0, torch.device("cpu"), world=world, rank=rank, seed=3)
    hist = []
    for step in range(3):
        cams = tr._next_cameras()
        view_data(cams[rank], m)                                   # geometry gradients of THIS rank's view
        gen = torch.Generator().manual_seed(500 + cams[rank])
        rec = RasterRecord()
        rec.drgb, rec.view_dirs = torch.randn(N, 3, generator=gen), torch.zeros(N, 3)
        xyz_rendered = m._xyz.detach().clone()
        tr._exchange_grads(True, False, rec)                         # two-stream form, no surgery
        assert tr._pending_sh is not None and tr._pending_sh[0] == "views" and rec.drgb is None
        m.optimizer.step()                                           # geometry Adam runs BEFORE the deferred SH update ...
        for g in m.optimizer.param_groups:
            g["params"][0].grad = None
        tr.join_side()                                               # ... which must still see the means the views were rendered with
        hist.append((cams, xyz_rendered))
    torch.save(dict(hist=hist, xyz=m._xyz.detach(), op=m._opacity.detach(), dc=m._features_dc.detach(),
                    rest=m._features_rest.detach(), scale=m.optimizer.grad_scale), out + f".{rank}")
    dist.destroy_process_group()


def test_deferred_sh_exchange_two_ranks_matches_serial(tmp_path):
    """with world > 1 the two-stream branch must all-reduce the geometry bucket and queue the all-view SH
    update; replicas stay identical and equal the single-process accumulation over the same two cameras per step."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "def.pt")
    mp.spawn(deferred_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("xyz", "op", "dc", "rest"):
        assert torch.equal(r0[k], r1[k]), f"replicas diverged in {k}"
    assert r0["scale"] == 0.5
    m = _sh_model()                                                  # serial reference: accumulate both views by hand
    for cams, _ in r0["hist"]:
        gx = go = gsh = 0
        for v in cams:
            view_data(v, m)
            gx, go = gx + m._xyz.grad, go + m._opacity.grad
            drgb = torch.randn(N, 3, generator=torch.Generator().manual_seed(500 + v))
            gsh = gsh + _basis_outer(m._xyz.detach(), _Cam(v).camera_center, drgb, 3)
        with torch.no_grad():
            m._features_dc.sub_(0.1 * 0.5 * gsh[:, :1]); m._features_rest.sub_(0.1 * 0.5 * gsh[:, 1:])
            m._xyz.sub_(0.1 * 0.5 * gx); m._opacity.sub_(0.1 * 0.5 * go)
    assert torch.allclose(r0["xyz"], m._xyz, atol=1e-6) and torch.allclose(r0["op"], m._opacity, atol=1e-6)
    assert torch.allclose(r0["dc"], m._features_dc, atol=1e-6) and torch.allclose(r0["rest"], m._features_rest, atol=1e-6)
    assert float(r0["dc"].abs().max()) > 0


def visi_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcr_gaus_amd import trainer as T
    tr = make_trainer(world, rank)
    tr.model.get_inside_gaus_normalized = lambda: (torch.arange(N) % 5 != 0, None)
    tr.model._xyz = torch.nn.Parameter(torch.zeros(N, 3))
    seen = []

    def fake_visi(cam, model, pipe, bg):
        seen.append(cam)
        g = torch.Generator().manual_seed(900 + cam)
        return {"countlist": (torch.rand(N, generator=g) > 0.7).int()}

    def fake_count(cam, model, pipe, bg):
        g = torch.Generator().manual_seed(700 + cam)
        return {"gaussians_count": torch.randint(0, 9, (N,), generator=g, dtype=torch.int32),
                "important_score": torch.rand(N, generator=g)}

    T.visi_acc_render, T.count_render = fake_visi, fake_count
    cams = list(range(7))                                            # odd count: ranks get 4 / 3 cameras
    mask = tr.visibility_mask(cams)
    cnt, imp = tr.importance_scores(cams)
    torch.save(dict(mask=mask, cnt=cnt, imp=imp, seen=seen), out + f".{rank}")
    dist.destroy_process_group()


def test_visibility_and_importance_passes_shard_cameras(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "vis.pt")
    mp.spawn(visi_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert sorted(r0["seen"] + r1["seen"]) == list(range(7)) and not set(r0["seen"]) & set(r1["seen"])
    count = sum((torch.rand(N, generator=torch.Generator().manual_seed(900 + c)) > 0.7).int() for c in range(7))
    want = (count > 0) & (torch.arange(N) % 5 != 0)
    assert torch.equal(r0["mask"], want) and torch.equal(r1["mask"], want)
    cnt = imp = 0
    for c in range(7):
        g = torch.Generator().manual_seed(700 + c)
        cnt = cnt + torch.randint(0, 9, (N,), generator=g, dtype=torch.int32)
        imp = imp + torch.rand(N, generator=g)
    assert torch.equal(r0["cnt"], cnt) and torch.allclose(r0["imp"], imp, atol=1e-5) and torch.equal(r1["cnt"], cnt)


def test_bench_gpus_flag_relaunches_itself_as_n_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-executes the same command line through torch.distributed.run with N
    processes on the loop-back address and exits with the launcher's return code (host logic only: no GPU, nothing is run)."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **k):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a launcher that started a different number of ranks than --gpus is an error, not a silently smaller job
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(AssertionError, match="--gpus 4"):
        bench.main()


def rs_ag_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcr_gaus_amd.rasterizer import RasterRecord
    res = {}
    for algo in ("allreduce", "rs_ag"):
        m = _sh_model()
        tr = Trainer(make_config("tnt"), m, [_Cam(i) for i in range(9)], 1.0, torch.device("cpu"), world=world, rank=rank, seed=3,
                     exchange=algo)
        for mode in ("serial", "deferred"):
            cams = tr._next_cameras()
            view_data(cams[rank], m)
            rec = RasterRecord()
            rec.drgb, rec.view_dirs = torch.randn(N, 3, generator=torch.Generator().manual_seed(500 + cams[rank])), torch.zeros(N, 3)
            if mode == "serial":
                def rebuild(drgb_all, campos_all, m=m):
                    g = sum(_basis_outer(m._xyz.detach(), campos_all[v], drgb_all[v], 3) for v in range(drgb_all.shape[0]))
                    return g[:, :1].contiguous(), g[:, 1:].contiguous()
                tr._sh_grads_from_rgb = rebuild
                tr._exchange_grads(False, True, rec)                  # (surgery step: serial exchange, no early feature step)
            else:
                tr._exchange_grads(True, False, rec)
            res[(algo, mode)] = dict(gx=m._xyz.grad.clone(), go=m._opacity.grad.clone(), ex=tr.last_exchange,
                                     numel=m._xyz.grad.numel() + m._opacity.grad.numel())
            m.optimizer.step()
            for g in m.optimizer.param_groups:
                g["params"][0].grad = None
            tr.join_side()
        res[(algo, "params")] = dict(xyz=m._xyz.detach().clone(), op=m._opacity.detach().clone(), dc=m._features_dc.detach().clone())
    torch.save(res, out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_all_gather_exchange_equals_the_all_reduce(tmp_path, world):
    """`Trainer(exchange="rs_ag")`: the geometry bucket goes through reduce-scatter + all-gather (padded to a multiple of
    the ranks: world 3 exercises the padding) instead of one all-reduce -- same summed gradients on every rank, in the
    serial and in the deferred (two-stream) form, and the replicas end with the same parameters."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rsag.pt")
    mp.spawn(rs_ag_worker, args=(world, port, out), nprocs=world, join=True)
    rs = [torch.load(out + f".{r}") for r in range(world)]
    for mode in ("serial", "deferred"):
        a, b = rs[0][("allreduce", mode)], rs[0][("rs_ag", mode)]
        assert b["ex"].endswith("+rs_ag") and not a["ex"].endswith("+rs_ag")
        assert torch.allclose(a["gx"], b["gx"], rtol=1e-6, atol=1e-7) and torch.allclose(a["go"], b["go"], rtol=1e-6, atol=1e-7)
        for r in range(1, world):                                  # every rank holds the same sums
            assert torch.equal(rs[r][("rs_ag", mode)]["gx"], b["gx"]) and torch.equal(rs[r][("rs_ag", mode)]["go"], b["go"])
    if world == 3:
        assert rs[0][("rs_ag", "serial")]["numel"] % 3 != 0          # (the bucket really needed padding)
    for k in ("xyz", "op", "dc"):
        assert torch.allclose(rs[0][("allreduce", "params")][k], rs[0][("rs_ag", "params")][k], rtol=1e-6, atol=1e-7)
        for r in range(1, world):
            assert torch.equal(rs[r][("rs_ag", "params")][k], rs[0][("rs_ag", "params")][k])


# ---- round 5: the bucket of the ONE-KERNEL tail under data parallelism carries ACTIVATED-space gradients --------------------
class _NamedModel:
    """The four geometry groups under their real names + one more group (semantic features), as `Trainer._allreduce_grads(sink=...)`
    tells them apart."""

    def __init__(self):
        g = torch.Generator().manual_seed(1)
        self._xyz = torch.nn.Parameter(torch.randn(N, 3, generator=g))
        self._scaling = torch.nn.Parameter(torch.randn(N, 3, generator=g))
        self._rotation = torch.nn.Parameter(torch.randn(N, 4, generator=g))
        self._opacity = torch.nn.Parameter(torch.randn(N, 1, generator=g))
        self._objects_dc = torch.nn.Parameter(torch.randn(N, 1, 2, generator=g))
        self.optimizer = StubOptim([])
        self.optimizer.param_groups = [{"params": [p], "lr": 0.1, "name": n} for n, p in
                                       [("xyz", self._xyz), ("opacity", self._opacity), ("scaling", self._scaling),
                                        ("rotation", self._rotation), ("obj_dc", self._objects_dc)]]


def _activated_grads(view):
    g = torch.Generator().manual_seed(900 + view)
    return dict(xyz=torch.randn(N, 3, generator=g), scales=torch.randn(N, 3, generator=g), rots=torch.randn(N, 4, generator=g),
                opac=torch.randn(N, 1, generator=g), nworld=torch.randn(N, 3, generator=g), obj=torch.randn(N, 1, 2, generator=g))


def activated_worker(rank, world, port, out, with_normals):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcr_gaus_amd.gaussian_model import GeometrySink
    m = _NamedModel()
    tr = Trainer(make_config("tnt"), m, list(range(8)), 1.0, torch.device("cpu"), world=world, rank=rank, seed=3)
    tr.factorised_sh = False
    cams = tr._next_cameras()
    a = _activated_grads(cams[rank])
    m._xyz.grad, m._objects_dc.grad = a["xyz"].clone(), a["obj"].clone()
    sink = GeometrySink(armed=True)
    sink.grads = [a["scales"].clone(), a["rots"].clone(), a["opac"].clone(), a["nworld"].clone() if with_normals else None]
    tr._allreduce_grads(sink=sink)
    # the raw-parameter gradients of the geometry groups do not exist in this form of the step
    assert m._scaling.grad is None and m._rotation.grad is None and m._opacity.grad is None
    assert (sink.grads[3] is None) == (not with_normals)
    assert sink.grads[1].data_ptr() % 16 == 0                       # (the kernel reads the quaternion gradient as float4)
    torch.save(dict(cams=cams, xyz=m._xyz.grad.clone(), grads=[None if t is None else t.clone() for t in sink.grads],
                    obj=m._objects_dc.grad.clone(), scale=m.optimizer.grad_scale), out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,with_normals", [(2, True), (2, False), (3, True)])
def test_activated_space_bucket_equals_single_process_sum(tmp_path, world, with_normals):
    """the data-parallel step exchanges the gradients w.r.t. (mean, activated scales, unit quaternion, opacity,
    world-space axis column) -- 56 B per Gaussian -- and runs the one-kernel tail on the SUM; every rank must hold the same sums,
    equal to one process adding up the same views, with the other groups' raw gradients riding in the same bucket."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "act.pt")
    mp.spawn(activated_worker, args=(world, port, out, with_normals), nprocs=world, join=True)
    rs = [torch.load(out + f".{r}") for r in range(world)]
    tot = None
    for v in rs[0]["cams"]:
        a = _activated_grads(v)
        tot = a if tot is None else {k: tot[k] + a[k] for k in a}
    for r in rs:
        assert r["cams"] == rs[0]["cams"] and r["scale"] == 1.0 / world
        assert torch.equal(r["xyz"], rs[0]["xyz"]) and torch.equal(r["obj"], rs[0]["obj"])
        for x, y in zip(r["grads"], rs[0]["grads"]):
            assert (x is None and y is None) or torch.equal(x, y)
    assert torch.allclose(rs[0]["xyz"], tot["xyz"], atol=1e-6) and torch.allclose(rs[0]["obj"], tot["obj"], atol=1e-6)
    for got, k in zip(rs[0]["grads"], ("scales", "rots", "opac", "nworld")):
        if got is None:
            assert k == "nworld" and not with_normals
        else:
            assert got.shape == tot[k].shape and torch.allclose(got, tot[k], atol=1e-6), k
