#!/bin/bash
# A/B of environment switches on the metric workload: bash profiles/r3_ab.sh "TAG ENV=.. ENV=.." ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
    set -- $spec; tag=$1; shift
    env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), d['stage_ms'], d['config'].get('emitted_instances'))"
done
