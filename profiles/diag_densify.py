"""Where a densification event of the schedule-inclusive window goes (bench.py `schedule_inclusive`: 6.0 ms / iteration in
round 3 against 1.36 ms for the plain step).  One event = visibility passes + `densify_and_prune` + the steps that follow it
with a new N (new buffer sizes).  Wall clock around each phase with the device drained before and after.
  python profiles/diag_densify.py [--events 2] [--after 12]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=2)
    ap.add_argument("--after", type=int, default=12)
    a = ap.parse_args()
    from vcr_gaus_amd import synthetic
    from vcr_gaus_amd.trainer import BenchTrainer
    dev = torch.device("cuda:0")
    n, views, W, H, focal, sem, smult = synthetic.workload("metric_1m_1080p")
    raw = synthetic.make_gaussians(n, seed=0)
    cams = synthetic.make_cameras(8, W, H, focal, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bt = BenchTrainer(raw, cams, dev)
    torch.cuda.synchronize()
    setup = {"trainer_setup_ms": 1e3 * (time.perf_counter() - t0), "arena_GB": bt.arena_bytes / 2 ** 30}
    t0 = time.perf_counter()
    extra = bt.tr.reserve_arena(factor=1.0, min_gb=2.0)
    torch.cuda.synchronize()
    setup["a_second_2GB_arena_ms"] = 1e3 * (time.perf_counter() - t0)
    bt.prime()
    tr = bt.tr
    for i in range(20):
        bt.step(i)
    tr.join_side(); torch.cuda.synchronize()

    def wall(fn):
        tr.join_side(); torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        tr.join_side(); torch.cuda.synchronize()
        return r, 1e3 * (time.perf_counter() - t0)

    out = []
    o = tr.cfg.optim
    dl = o.densify_large
    from vcr_gaus_amd import _lib as _L
    _L.profile_enable(True); _L.profile_read()
    for k in range(16):
        bt.step(0)
    tr.join_side(); torch.cuda.synchronize()
    pr0 = _L.profile_read(); _L.profile_enable(False)
    setup["stage_ms_before"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in pr0.items()}
    setup["R_V_E_before"] = [bt.last_R, bt.last_V, bt.last_E]
    for ev in range(a.events):
        rec = {"N_before": tr.model._xyz.shape[0]}
        _, rec["steady_step_ms"] = wall(lambda: [bt.step(0) for _ in range(10)])
        rec["steady_step_ms"] /= 10
        vcams = tr._visibility_cameras(dl.sample_cams)
        visi, rec["visibility_ms"] = wall(lambda: tr.visibility_mask(vcams))
        st0 = torch.cuda.memory_stats(dev)
        if ev == 0:
            snap = torch.cuda.memory_snapshot()
            free = sorted((b["size"] for seg in snap for b in seg["blocks"] if b["state"] == "inactive"), reverse=True)
            rec["free_blocks_MB_before"] = [round(x / 2 ** 20) for x in free[:12]]
            rec["segments_MB"] = sorted((round(seg["total_size"] / 2 ** 20) for seg in snap), reverse=True)[:12]
            rec["segment_streams"] = sorted({seg["stream"] for seg in snap})
            torch.cuda.memory._record_memory_history(max_entries=20000)
        _, rec["densify_and_prune_ms"] = wall(lambda: tr.model.densify_and_prune(o.densify_grad_threshold, 0.005, tr.extent, None, visi))
        if ev == 0:
            hist = torch.cuda.memory._snapshot()
            torch.cuda.memory._record_memory_history(enabled=None)
            evs = [e for tr_ in hist.get("device_traces", []) for e in tr_]
            rec["segment_allocs_MB"] = [round(e["size"] / 2 ** 20) for e in evs if e["action"] == "segment_alloc"]
            rec["allocs_over_50MB"] = [round(e["size"] / 2 ** 20) for e in evs if e["action"] == "alloc" and e["size"] > 50 * 2 ** 20][:60]
            rec["alloc_streams"] = sorted({e.get("stream") for e in evs if e["action"] == "alloc"})
        st1 = torch.cuda.memory_stats(dev)
        rec["densify_hipmallocs"] = st1["num_device_alloc"] - st0["num_device_alloc"]
        rec["N_after"] = tr.model._xyz.shape[0]
        steps = []
        for k in range(a.after):
            s0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
            _, t = wall(lambda: bt.step(0))
            steps.append((round(t, 3), torch.cuda.memory_stats(dev)["num_device_alloc"] - s0))
        rec["steps_after_ms_and_hipmallocs"] = steps
        from vcr_gaus_amd import _lib
        _lib.profile_enable(True); _lib.profile_read()
        for k in range(16):
            bt.step(0)
        tr.join_side(); torch.cuda.synchronize()
        pr = _lib.profile_read(); _lib.profile_enable(False)
        rec["stage_ms_after"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items()}
        rec["R_V_E_after"] = [bt.last_R, bt.last_V, bt.last_E]
        rec["quad_lists_after"] = bool(tr._quad_on)
        rec["reserved_GB"] = torch.cuda.memory_reserved(dev) / 2 ** 30
        out.append(rec)
    print(json.dumps({"setup": setup, "events": out}))


if __name__ == "__main__":
    main()
