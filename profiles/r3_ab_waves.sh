#!/bin/bash
# waves-per-SIMD variants of the row-packed backward (libs built with EXTRA=-DVCR_ROWS_WAVES=n): bash profiles/r3_ab_waves.sh "WL .."
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in ${1:-metric_1m_1080p}; do for lib in libvcr_raster.so libvcr_w3.so; do VCR_LIB=$PWD/vcr_gaus_amd/$lib python bench.py --workload $wl --steps 30 --warmup 8 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$lib', round(d['ms_per_step'],4), d['stage_ms']['composite_bwd'])"; done; done
