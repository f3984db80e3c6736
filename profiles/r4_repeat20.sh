cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "" "--quad-below 0" "--no-arena" "--quad-below 0 --no-arena"; do
  for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-context $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg |', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['step_ms'].items() if k!='note'})"; done; done
