#!/bin/bash
# round 6, call 17: two-phase forward with a dummy slot for exhausted lanes and a fast path for empty quads (libvcr_raster_micro.so)
# against the committed kernel (libvcr_raster.so): times, bit comparison of the dumps
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run17
mkdir -p $OUT
cd $R
export TMPDIR=/tmp AB_REPS=10 AB_CAMS=4
WL="metric_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200"
for k in 1 2; do
timeout 300 python profiles/r6_fwd_ab.py base $WL > $OUT/ab_base_$k.txt 2>&1; grep MEAN $OUT/ab_base_$k.txt
VCR_LIB=$R/vcr_gaus_amd/libvcr_raster_micro.so timeout 300 python profiles/r6_fwd_ab.py micro $WL > $OUT/ab_micro_$k.txt 2>&1; grep MEAN $OUT/ab_micro_$k.txt
done
python profiles/r6_fwd_cmp.py base micro > $OUT/cmp.txt 2>&1; tail -1 $OUT/cmp.txt
