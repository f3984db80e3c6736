#!/bin/bash
# Round 4: quad-granular binning on / off at the metric scene and the other workloads (bench lines only, no context).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4_quad_ab; mkdir -p $O
for wl in ${WLS:-metric_1m_1080p fullframe_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200}; do
  for q in 1 0; do
    VCR_QUAD_LISTS=$q python bench.py --steps 40 --warmup 10 --workload $wl --no-cpu-baseline --no-context > $O/${wl}_q$q.json 2> $O/${wl}_q$q.err
    python - <<PY
import json
try:
    l=json.loads(open("$O/${wl}_q$q.json").read().strip().splitlines()[-1])
    print("$wl", "quad=$q", "ms/step %.3f" % l["ms_per_step"], "E", l["config"]["emitted_instances"], "R", l["config"]["tile_instances_R"], {k: l["stage_ms"][k] for k in l["stage_ms"]})
except Exception as e:
    print("$wl quad=$q FAILED", e); print(open("$O/${wl}_q$q.err").read()[-1500:])
PY
  done
done
