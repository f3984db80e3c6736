"""Two data-parallel ranks sharing ONE GPU (gloo transport, CUDA tensors): the whole `Trainer.train_step` with the real HIP
kernels, both streams and the real collectives of the factorised / deferred exchange -- what a one-rank process group cannot
show: replicas that render DIFFERENT cameras must end every step -- ordinary, densification and opacity-reset steps -- with
bit-identical parameters and statistics.  (That the exchanged quantities equal the single-process accumulation over the same
cameras is the CPU test tests/test_dp_gloo_cpu.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import make_synthetic_trainer
    raw = synthetic.make_gaussians(5000, seed=21)
    raw["scaling"] = raw["scaling"] + 1.2
    cams = synthetic.make_cameras(4, 128, 96, 110.0, device=dev)
    tr = make_synthetic_trainer(raw, cams, dev, world=world, rank=rank, preset="tnt", overlap_sh=overlap, overlap_min_gaussians=0,
                                optim={"densify_from_iter": 3, "densification_interval": 4, "densify_until_iter": 100,
                                       "opacity_reset_interval": 7})
    losses, picks, exch = [], [], []
    for _ in range(9):                     # includes a densification (it 4, 8) and an opacity reset (it 7): surgery steps
        tr.train_step()
        losses.append(float(tr.losses["total"]))
        picks.append(list(tr._picked))
        exch.append(tr.last_exchange)
    tr.join_side()
    tr.sync_densify_stats()                # (statistics are rank-local between the points where they are read)
    torch.cuda.synchronize()
    m = tr.model
    torch.save(dict(losses=losses, picks=picks, exch=exch, n=m._xyz.shape[0],
                    params={k: getattr(m, k).detach().cpu() for k in ["_xyz", "_features_dc", "_features_rest", "_scaling",
                                                                      "_rotation", "_opacity"]},
                    accum=m.xyz_gradient_accum.cpu(), denom=m.denom.cpu()), out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_on_one_gpu_stay_identical(device, tmp_path, overlap):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out, overlap), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["picks"] == r1["picks"] and all(len(set(p)) == 2 for p in r0["picks"])      # same batch, different cameras
    assert r0["n"] == r1["n"] and r0["n"] != 5000                                            # densified in lock-step
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), f"replicas diverged in {k}"
    assert torch.equal(r0["accum"], r1["accum"]) and torch.equal(r0["denom"], r1["denom"])
    assert r0["losses"] != r1["losses"]                                                      # (each rank saw its own view)
    want = "factorised-deferred" if overlap else "factorised"
    assert want in r0["exch"] and r0["exch"] == r1["exch"], r0["exch"]
