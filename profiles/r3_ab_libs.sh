#!/bin/bash
# Same-box A/B of two builds of the library: bash profiles/r3_ab_libs.sh [workload] [steps]
# libvcr_prev.so = the previous commit (built from a git worktree), libvcr_raster.so = the working tree; interleaved runs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-metric_1m_1080p}; STEPS=${2:-60}
one() { tag=$1; lib=$2; VCR_LIB=$PWD/vcr_gaus_amd/$lib python bench.py --workload $WL --steps $STEPS --warmup 10 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '$WL', round(d['ms_per_step'],4), d['stage_ms'])"; }
for rep in 1 2 3; do one prev libvcr_prev.so; one new libvcr_raster.so; done
