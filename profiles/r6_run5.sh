#!/bin/bash
# round 6, GPU call: third form of the two-phase forward (deferred colour halves, head prefetch depth, uniform walk for dense groups)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run5
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=$R/vcr_gaus_amd
for t in v2 tp d1 d2w5 nodense; do
    lib=$L/libvcr_raster_$t.so; [ $t == tp ] && lib=$L/libvcr_raster.so
    VCR_LIB=$lib timeout 400 python profiles/r6_fwd_ab.py $t metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 fullframe_1m_1080p > $OUT/ab_$t.txt 2>&1
    grep MEAN $OUT/ab_$t.txt
done
for t in tp d1 d2w5 nodense; do python profiles/r6_fwd_cmp.py v2 $t > $OUT/cmp_$t.txt 2>&1; tail -1 $OUT/cmp_$t.txt; done
