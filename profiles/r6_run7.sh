#!/bin/bash
# round 6, call 7: the tree with the two-level emission scan, the schedule-inclusive headline and the exchange diagnostics --
# whole GPU suite, FETCH_SIZE calibration, the bench line as the driver runs it, kernel trace of the same command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run7
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt
bash profiles/r6_fetch_calib.sh > $OUT/fetch_calib.log 2>&1; cp gpurun_out/r6_fetch_calib.txt $OUT/ 2>/dev/null
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
tail -c 600 $OUT/bench_driver_form.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
cd $R
cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $OUT/bench_kernel_stats.csv
python profiles/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1) 3 > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/trace
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_run7/bench_driver_form.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "value_steady", "ms_per_step", "ms_per_step_steady", "densify_event_ms", "stage_ms")}, d["roofline"]["frac"])
PY
head -25 $OUT/bench_kernel_stats.csv | cut -c1-150
