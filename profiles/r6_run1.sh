#!/bin/bash
# round 6, GPU call 1: two-phase forward -- bit-identity against the v2 loop, timing of the variants, raster parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run1
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=$R/vcr_gaus_amd
for t in v2 tp1 tp0 tp0w6 tp1w5; do
    lib=$L/libvcr_raster_$t.so; [ $t == tp1 ] && lib=$L/libvcr_raster.so
    VCR_LIB=$lib timeout 400 python profiles/r6_fwd_ab.py $t metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 > $OUT/ab_$t.txt 2>&1
    grep MEAN $OUT/ab_$t.txt
done
for t in tp1 tp0 tp0w6 tp1w5; do python profiles/r6_fwd_cmp.py v2 $t > $OUT/cmp_$t.txt 2>&1; tail -1 $OUT/cmp_$t.txt; done
timeout 900 python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_sampled_gpu.py -x -q > $OUT/pytest_raster.txt 2>&1
tail -5 $OUT/pytest_raster.txt
