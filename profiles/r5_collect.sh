#!/bin/bash
# Round-5 final evidence, in three parts (one gpurun call each: bash profiles/r5_collect.sh A|B|C).  Every part runs the WHOLE
# GPU suite once as ONE command on the final tree (three logs) and then collects its share of the rest.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_final
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
part=${1:-A}
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$part.txt 2>&1
tail -n 6 $OUT/pytest_gpu_$part.txt
case $part in
A)  # headline bench line as the driver runs it, kernel trace + timeline of the same command
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
    cd $R
    python profiles/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1) 3 > $OUT/step_timeline.txt 2>&1
    cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $OUT/bench_kernel_stats.csv
    rm -rf $OUT/trace
    tail -n 3 $OUT/step_timeline.txt ;;
B)  # PMC passes of the headline command (one rocprofv3 run per counter set)
    VCR_ROUND=r5 bash profiles/collect_pmc.sh sq grbm fetch write > $OUT/pmc.log 2>&1
    ls $R/gpurun_out/r5_pmc_* ;;
C)  # the other BASELINE workloads, the two-rank functional record, a second headline line
    for wl in c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 fullframe_1m_1080p; do
        timeout 150 python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
    done
    VCR_DIST_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-context > $OUT/bench_gpus2_gloo_one_gpu.json 2> $OUT/bench_gpus2_gloo_one_gpu.err
    timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_metric_50.json 2> $OUT/bench_metric_50.err
    for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), round(d['value'], 1), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
    done ;;
esac
