#!/bin/bash
# Round 4: no event record behind the publish kernel (default) against the library built with it (VCR_LIB=.../libvcr_raster_ev.so).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4; do for lib in "" $PWD/vcr_gaus_amd/libvcr_raster_ev.so; do
  VCR_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with_event=%d' % bool('$lib'), 'ms/step %.4f' % l['ms_per_step'], 'median %.3f' % l['step_ms']['median'], {k: l['stage_ms'][k] for k in ('preprocess','depth_sort_scan')})"
done; done
