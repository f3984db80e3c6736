#!/bin/bash
# Round 4 experiment (VERDICT r3 item 4 i): the side stream (SH update + SH -> RGB) confined to M compute units.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4_cumask; mkdir -p $O
for wl in ${WLS:-metric_1m_1080p c5_360_5m_1600x1200}; do
  for m in ${MS:-0 64 96 128 160 192}; do
    python bench.py --steps 40 --warmup 10 --workload $wl --side-cus $m --no-cpu-baseline --no-context > $O/${wl}_m$m.json 2> $O/${wl}_m$m.err
    python - <<PY
import json
try:
    l=json.loads(open("$O/${wl}_m$m.json").read().strip().splitlines()[-1])
    print("$wl", "side_cus=$m", "ms/step %.3f" % l["ms_per_step"], {k: l["stage_ms"][k] for k in ("depth_sort_scan","binning","composite_fwd","composite_bwd")})
except Exception as e:
    print("$wl side_cus=$m FAILED", e); print(open("$O/${wl}_m$m.err").read()[-800:])
PY
  done
done
