#!/bin/bash
# kernel trace of the bench command -> step timeline + per-kernel stats: bash profiles/r3_timeline.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
OUT=$R/gpurun_out/r3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context "$@" > $OUT/bench_traced.json 2> $OUT/bench_traced.err
KT=$(ls $OUT/trace/*kernel_trace.csv | head -1)
python $R/profiles/step_timeline.py $KT 3 > $OUT/step_timeline.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cat $OUT/step_timeline.txt
