#!/bin/bash
# A/B of environment switches over several workloads: bash profiles/r3_ab_wl.sh "WL1 WL2 .." "TAG ENV=.." "TAG ENV=.." ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
WLS=$1; shift
SPECS=("$@")
for wl in $WLS; do
  for spec in "${SPECS[@]}"; do
    read -r tag envs <<< "$spec"
    env $envs python bench.py --workload $wl --steps 30 --warmup 8 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$wl', '$tag', round(d['ms_per_step'],4), 'fwd', s['composite_fwd'], 'bwd', s['composite_bwd'])"
  done
done
