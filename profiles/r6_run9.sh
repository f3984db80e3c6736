#!/bin/bash
# round 6, call 9: persistent compositing, second form (first item of a wave dealt, the rest drawn one ahead; background batches
# dealt; plain longest-first order) against the block form -- times, bit comparison, the raster parity tests through it; the
# count kernel with one Gaussian per thread; the driver-form bench with the quota-aware CPU thread count
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run9
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
export AB_REPS=6 AB_CAMS=3
run() {  # tag, workloads, env...
    local tag=$1 wl=$2; shift 2
    env "$@" timeout 300 python profiles/r6_fwd_ab.py $tag $wl > $OUT/ab_$tag.txt 2>&1
    grep MEAN $OUT/ab_$tag.txt
}
WL="metric_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200"
run blk "$WL dense_1m_1080p" VCR_PERSIST_FWD=0
run blkns "$WL" VCR_NO_SNAKE=1
run f2b2 "$WL" VCR_PERSIST_FWD=2 VCR_PERSIST_BWD=2
run f3b3 "$WL" VCR_PERSIST_FWD=3 VCR_PERSIST_BWD=3
run f4b3 "$WL" VCR_PERSIST_FWD=4 VCR_PERSIST_BWD=3
run f5b4 "$WL" VCR_PERSIST_FWD=5 VCR_PERSIST_BWD=4
run d3 "dense_1m_1080p" VCR_PERSIST_FWD2=3
run d5 "dense_1m_1080p" VCR_PERSIST_FWD2=5
for t in f2b2 f3b3 f4b3 f5b4 d3 d5; do python profiles/r6_fwd_cmp.py blk $t > $OUT/cmp_$t.txt 2>&1; tail -1 $OUT/cmp_$t.txt; done
VCR_PERSIST_FWD=3 VCR_PERSIST_FWD2=3 VCR_PERSIST_BWD=3 timeout 900 python -m pytest tests/test_raster_parity_gpu.py tests/test_deterministic_bwd_gpu.py tests/test_ops_gpu.py -m gpu -q > $OUT/pytest_p3.txt 2>&1; tail -3 $OUT/pytest_p3.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
grep "bench " $OUT/bench_driver_form.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_run9/bench_driver_form.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "value_steady", "ms_per_step", "ms_per_step_steady", "densify_event_ms", "stage_ms")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
