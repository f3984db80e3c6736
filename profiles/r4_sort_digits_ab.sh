#!/bin/bash
# Round 4: depth sort as three 9-bit passes over 27 key bits (default) against four 8-bit passes (VCR_SORT_DIGIT_BITS=8), in the step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in metric_1m_1080p c4_tnt_2m_1080p c5_360_5m_1600x1200; do
for i in 1 2 3; do for b in 9 8; do
  VCR_SORT_DIGIT_BITS=$b python bench.py --steps 30 --warmup 8 --workload $wl --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl digits=$b', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('depth_sort_scan','binning')})"
done; done; done
