#!/bin/bash
# Same box, alternating: the round-3 tree against this tree at the driver's settings (--steps 20 --warmup 5).
#   mkdir -p build/r3tree && git archive eb6122c | tar -x -C build/r3tree && make -C build/r3tree/vcr_gaus_amd/csrc -j8
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { (cd $1 && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-context $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 |', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['step_ms'].items() if k!='note'}, d['stage_ms'])"); }
for i in 1 2 3; do one $R/build/r3tree r3; one $R r4; done
