#!/bin/bash
# round 6, call 12: densification event with the next event's virtual cameras formed ahead on a worker thread
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run12
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_run12/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "value_steady", "ms_per_step", "ms_per_step_steady", "densify_event_ms", "stage_ms")}, d["roofline"]["frac"])
print(d["schedule_inclusive"])
PY
timeout 900 python -m pytest tests/test_train_loop_fullsize_gpu.py tests/test_densify.py tests/test_train_step_gpu.py -m gpu -x -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed" $OUT/pytest.txt
