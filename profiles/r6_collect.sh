#!/bin/bash
# Round-6 evidence, in parts (one gpurun call each: bash profiles/r6_collect.sh R|S|A|B|C).
#   R: the whole GPU suite twice in REPORT mode (every gradient comparison logged, nothing asserted) -> the known-miss list
#   S: the whole GPU suite three times as ONE command in strict mode
#   A: headline bench line as the driver runs it, kernel trace + timeline of the same command, smoke()
#   B: PMC passes of the headline command (one rocprofv3 run per counter set)
#   C: the other BASELINE workloads, the two-rank functional record
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_final
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
part=${1:-A}
case $part in
R)  for k in 1 2; do
        rm -f $OUT/grad_report_$k.txt
        VCR_GRAD_REPORT=$OUT/grad_report_$k.txt timeout 1100 python -m pytest tests -m gpu -q > $OUT/pytest_report_$k.txt 2>&1
        grep -n "passed\|failed" $OUT/pytest_report_$k.txt
    done ;;
S)  for k in 1 2 3; do timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$k.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu_$k.txt; done ;;
A)  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
    grep "bench " $OUT/bench_driver_form.err
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
    cd $R
    python profiles/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1) 3 > $OUT/step_timeline.txt 2>&1
    cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $OUT/bench_kernel_stats.csv
    rm -rf $OUT/trace
    tail -n 3 $OUT/step_timeline.txt
    timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt ;;
B)  VCR_ROUND=r6 bash profiles/collect_pmc.sh sq grbm fetch write > $OUT/pmc.log 2>&1
    ls $R/gpurun_out/r6_pmc_* ;;
C)  for wl in c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 fullframe_1m_1080p; do
        timeout 200 python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
    done
    VCR_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_gpus2_gloo_one_gpu.json 2> $OUT/bench_gpus2_gloo_one_gpu.err
    for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), round(d['value'], 1), d.get('value_steady'), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
    done ;;
esac
