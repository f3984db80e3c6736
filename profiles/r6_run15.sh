#!/bin/bash
# round 6, call 15: grouped backward WITH two-level culling against the per-chunk form; run-to-run spread of the c1 gradient figures
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run15
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
export AB_REPS=8 AB_CAMS=4 AB_DUMP=0
WL="metric_1m_1080p c2_dtu_300k_800x600 c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p"
VCR_BWD_GROUP=0 timeout 400 python profiles/r6_fwd_ab.py rows $WL > $OUT/ab_rows.txt 2>&1; grep MEAN $OUT/ab_rows.txt
VCR_BWD_GROUP=1 timeout 400 python profiles/r6_fwd_ab.py group2 $WL > $OUT/ab_group2.txt 2>&1; grep MEAN $OUT/ab_group2.txt
VCR_BWD_GROUP=1 timeout 1000 python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_sampled_gpu.py tests/test_fullsize_properties_gpu.py tests/test_deterministic_bwd_gpu.py -m gpu -q > $OUT/pytest_group2.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_group2.txt; grep -n "^FAILED\|^E  " $OUT/pytest_group2.txt | head -20
timeout 300 python profiles/r6_c1_noise.py 15 > $OUT/c1_noise.txt 2>&1; grep "scales\|rots\|normals" $OUT/c1_noise.txt
