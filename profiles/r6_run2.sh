#!/bin/bash
# round 6, GPU call 2: second form of the two-phase forward (batched phase 1, late colour gather) -- variants vs v2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run2
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=$R/vcr_gaus_amd
for t in v2 tp tpw6 tpp0 tpp0w6; do
    lib=$L/libvcr_raster_$t.so; [ $t == tp ] && lib=$L/libvcr_raster.so
    VCR_LIB=$lib timeout 400 python profiles/r6_fwd_ab.py $t metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 > $OUT/ab_$t.txt 2>&1
    grep MEAN $OUT/ab_$t.txt
done
for t in tp tpw6 tpp0 tpp0w6; do python profiles/r6_fwd_cmp.py v2 $t > $OUT/cmp_$t.txt 2>&1; tail -1 $OUT/cmp_$t.txt; done
