cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in metric_1m_1080p c4_tnt_2m_1080p; do
for g in 0 320 384 512 768; do for i in 1 2; do
  VCR_SIDE_GRID=$g python bench.py --steps 30 --warmup 8 --workload $wl --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl side_grid=$g', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('depth_sort_scan','binning')})"
done; done; done
