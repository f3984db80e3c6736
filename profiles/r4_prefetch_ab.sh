#!/bin/bash
# Round 4: activations of the next camera written by the static tail (default) against the stand-alone activation kernel (VCR_NO_ACT_PREFETCH=1).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in metric_1m_1080p c5_360_5m_1600x1200; do
for i in 1 2 3; do for b in "" 1; do
  VCR_NO_ACT_PREFETCH=$b python bench.py --steps 30 --warmup 8 --workload $wl --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl no_prefetch=${b:-0}', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('preprocess','preprocess_bwd')})"
done; done; done
