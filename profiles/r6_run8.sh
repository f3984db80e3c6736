#!/bin/bash
# round 6, call 8: persistent wave-scheduled compositing (VCR_PERSIST_FWD / VCR_PERSIST_BWD = waves per SIMD, 0 = block form),
# with and without the folded launch order; bit comparison of every variant against the block form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run8
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
WL="metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200"
run() {  # tag, env...
    local tag=$1; shift
    env "$@" timeout 300 python profiles/r6_fwd_ab.py $tag $WL > $OUT/ab_$tag.txt 2>&1
    grep MEAN $OUT/ab_$tag.txt
}
run blk VCR_PERSIST_FWD=0 VCR_PERSIST_BWD=0
run f2b2 VCR_PERSIST_FWD=2 VCR_PERSIST_BWD=2
run f3b3 VCR_PERSIST_FWD=3 VCR_PERSIST_BWD=3
run f4b3 VCR_PERSIST_FWD=4 VCR_PERSIST_BWD=3
run f5b4 VCR_PERSIST_FWD=5 VCR_PERSIST_BWD=4
run f2b2ns VCR_PERSIST_FWD=2 VCR_PERSIST_BWD=2 VCR_NO_SNAKE=1
run f3b3ns VCR_PERSIST_FWD=3 VCR_PERSIST_BWD=3 VCR_NO_SNAKE=1
run f4b3ns VCR_PERSIST_FWD=4 VCR_PERSIST_BWD=3 VCR_NO_SNAKE=1
run f6b3ns VCR_PERSIST_FWD=6 VCR_PERSIST_BWD=3 VCR_NO_SNAKE=1
for t in f2b2 f3b3 f4b3 f3b3ns; do python profiles/r6_fwd_cmp.py blk $t > $OUT/cmp_$t.txt 2>&1; tail -1 $OUT/cmp_$t.txt; done
timeout 600 python -m pytest tests/test_raster_parity_gpu.py -m gpu -x -q > $OUT/pytest_raster_blk.txt 2>&1; tail -2 $OUT/pytest_raster_blk.txt
VCR_PERSIST_FWD=3 VCR_PERSIST_BWD=3 timeout 600 python -m pytest tests/test_raster_parity_gpu.py tests/test_deterministic_bwd_gpu.py -m gpu -x -q > $OUT/pytest_raster_p3.txt 2>&1; tail -2 $OUT/pytest_raster_p3.txt
# why did the driver-form bench not finish in 400 s in call 7?  progress marks on stderr
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
grep "bench " $OUT/bench_driver_form.err
