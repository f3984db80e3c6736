#!/bin/bash
# round 6, call 14: compositing backward with the survivors grouped across chunks (VCR_BWD_GROUP=1) against the per-chunk form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run14
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
export AB_REPS=8 AB_CAMS=4 AB_DUMP=0
WL="metric_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p"
VCR_BWD_GROUP=0 timeout 400 python profiles/r6_fwd_ab.py rows $WL > $OUT/ab_rows.txt 2>&1; grep MEAN $OUT/ab_rows.txt
VCR_BWD_GROUP=1 timeout 400 python profiles/r6_fwd_ab.py group $WL > $OUT/ab_group.txt 2>&1; grep MEAN $OUT/ab_group.txt
VCR_BWD_GROUP=1 timeout 1000 python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_sampled_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q > $OUT/pytest_group.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_group.txt; grep -n "^FAILED\|^E  " $OUT/pytest_group.txt | head -20
