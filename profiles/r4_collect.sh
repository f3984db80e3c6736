#!/bin/bash
# Round-4 evidence pass on the GPU box: bench lines of every workload, kernel stats + timeline of the headline command,
# kernel stats of c5 and the full-frame variant, the visibility passes (wall + kernel stats), PMC counter passes.
#   bash profiles/r4_collect.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4_final
mkdir -p $OUT
cd $R
for wl in c2_dtu_300k_800x600 c4_tnt_2m_1080p c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p; do
    python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
python bench.py --workload c2_dtu_300k_800x600 --preset dtu --steps 30 --warmup 10 --no-cpu-baseline --no-context > $OUT/bench_c2_preset_dtu.json 2> $OUT/bench_c2_preset_dtu.err
python bench.py --steps 50 --warmup 10 > $OUT/bench_metric.json 2> $OUT/bench_metric.err
python profiles/visi_profile.py --cams 200 --mode all > $OUT/visi_wall.json 2> $OUT/visi_wall.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python $R/profiles/step_timeline.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 3 > $OUT/step_timeline.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
for wl in c5_360_5m_1600x1200 fullframe_1m_1080p; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o t -- python $R/bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-context > $OUT/traced_$wl.json 2> $OUT/traced_$wl.err
    cp $(ls $OUT/trace_$wl/*kernel_stats.csv | head -1) $OUT/kernel_stats_$wl.csv
    python $R/profiles/step_timeline.py $(ls $OUT/trace_$wl/*kernel_trace.csv | head -1) 3 > $OUT/step_timeline_$wl.txt 2>&1
    rm -rf $OUT/trace_$wl
done
for mode in percam flags; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_visi_$mode -o t -- python $R/profiles/visi_profile.py --cams 40 --reps 1 --mode $mode > $OUT/visi_traced_$mode.json 2> $OUT/visi_traced_$mode.err
    cp $(ls $OUT/trace_visi_$mode/*kernel_stats.csv | head -1) $OUT/visi_kernel_stats_$mode.csv
    rm -rf $OUT/trace_visi_$mode
done
VCR_ROUND=r4 bash $R/profiles/collect_pmc.sh fetch write sq grbm lds
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 4), round(d['value'], 1), d['stage_ms'], 'frac', round(d['roofline']['frac'], 4), 'R', d['config']['tile_instances_R'], 'E', d['config'].get('emitted_instances'), d.get('schedule_inclusive'))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
