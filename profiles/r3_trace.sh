#!/bin/bash
# Round-3 measurement pass on the GPU box: bash profiles/r3_trace.sh <tag>
# kernel trace + stats of the bench command, step timeline, A/B of the sort digit width, serial (one-stream) stage times.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/r3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stages() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4), 'R', d['config']['tile_instances_R'])
PY
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/bench_traced.err
KT=$(ls $OUT/trace/*kernel_trace.csv | head -1)
python $R/profiles/step_timeline.py $KT 3 > $OUT/step_timeline.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cd $R
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; stages $OUT/bench_default.json
VCR_SORT_DIGIT_BITS=8 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_digit8.json 2>> $OUT/bench_default.err; stages $OUT/bench_digit8.json
VCR_NO_OVERLAP=1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_serial.json 2>> $OUT/bench_default.err; stages $OUT/bench_serial.json
VCR_NO_OVERLAP=1 VCR_SORT_DIGIT_BITS=8 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_serial8.json 2>> $OUT/bench_default.err; stages $OUT/bench_serial8.json
cat $OUT/step_timeline.txt
