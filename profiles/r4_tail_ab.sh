#!/bin/bash
# Round 4: static tail inside the rasterizer's backward (default) against the separate geometry-step kernel (VCR_NO_RASTER_TAIL=1).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in metric_1m_1080p c4_tnt_2m_1080p c5_360_5m_1600x1200 c2_dtu_300k_800x600; do
for i in 1 2 3; do for b in "" 1; do
  VCR_NO_RASTER_TAIL=$b python bench.py --steps 30 --warmup 8 --workload $wl --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl separate_tail=${b:-0}', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('composite_bwd','preprocess_bwd')})"
done; done; done
