#!/bin/bash
# Round 4: per-tile against per-quad binning with the kernels templated on the mode (the per-tile path is round 3's again).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in metric_1m_1080p c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 dense_1m_1080p; do
  for q in 0 100; do for i in 1 2; do
    python bench.py --steps 30 --warmup 8 --workload $wl --quad-below $q --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=l['config']
print('$wl', 'quad' if $q else 'tile', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], 'R/V %.2f' % (c['tile_instances_R']/max(c['visible_V'],1)), 'E', c['emitted_instances'], {k: l['stage_ms'][k] for k in l['stage_ms']})"
  done; done
done
