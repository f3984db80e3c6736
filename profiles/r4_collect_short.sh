#!/bin/bash
# Round-4 evidence refresh after the last kernel change (the static tail inside the rasterizer's backward): bench lines of every
# workload, kernel stats + timeline of the headline command and of c5, this tree against the round-3 tree on the same box.
# (Visibility passes and PMC counter passes: profiles/r4_collect.sh; the compositing kernels they describe did not change.)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4_final2
mkdir -p $OUT
cd $R
timeout 200 python bench.py --steps 50 --warmup 10 > $OUT/bench_metric.json 2> $OUT/bench_metric.err
for wl in c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p; do
    timeout 120 python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 120 python bench.py --workload c2_dtu_300k_800x600 --preset dtu --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_c2_preset_dtu.json 2> $OUT/bench_c2_preset_dtu.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python $R/profiles/step_timeline.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 3 > $OUT/step_timeline.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
wl=c5_360_5m_1600x1200
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o t -- python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-context > $OUT/traced_$wl.json 2> $OUT/traced_$wl.err
cp $(ls $OUT/trace_$wl/*kernel_stats.csv | head -1) $OUT/kernel_stats_$wl.csv
python $R/profiles/step_timeline.py $(ls $OUT/trace_$wl/*kernel_trace.csv | head -1) 3 > $OUT/step_timeline_$wl.txt 2>&1
rm -rf $OUT/trace_$wl
cd $R
timeout 150 bash profiles/r4_vs_r3.sh > $R/gpurun_out/r4_vs_r3.txt 2>&1
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), round(d['value'], 1), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4), d.get('schedule_inclusive', {}).get('ms_per_iter'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $R/gpurun_out/r4_vs_r3.txt
