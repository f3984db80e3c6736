cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in 0 640 768 1024; do for i in 1 2; do
  VCR_SIDE_GRID=$g python bench.py --steps 30 --warmup 8 --workload c5_360_5m_1600x1200 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 side_grid=$g', 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('depth_sort_scan','binning')})"
done; done
bash profiles/r4_vs_r3.sh 2>&1 | cut -c1-130
