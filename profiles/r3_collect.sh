#!/bin/bash
# Round-3 evidence pass on the GPU box: bench lines of every workload, kernel stats + timeline of the headline command,
# PMC counter passes.  bash profiles/r3_collect.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_final
mkdir -p $OUT
cd $R
for wl in c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p; do
    python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
python bench.py --workload c2_dtu_300k_800x600 --preset dtu --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_c2_preset_dtu.json 2> $OUT/bench_c2_preset_dtu.err
python bench.py --steps 50 --warmup 10 > $OUT/bench_metric.json 2> $OUT/bench_metric.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python $R/profiles/step_timeline.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 3 > $OUT/step_timeline.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
bash $R/profiles/collect_pmc.sh fetch write sq grbm lds
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), round(d['value'], 1), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4), 'R', d['config']['tile_instances_R'], 'E', d['config'].get('emitted_instances'))
PY
done
