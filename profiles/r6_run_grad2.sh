#!/bin/bash
# round 6: final forward (two-phase, per-frame form choice) against the uniform loop, then the whole GPU suite in REPORT mode on the
# tree's defaults (mixed-precision projection backward, FRAGILE_K = 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_grad2
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
VCR_LIB=$R/vcr_gaus_amd/libvcr_raster_v2.so timeout 400 python profiles/r6_fwd_ab.py v2 metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 fullframe_1m_1080p c5_360_5m_1600x1200 > $OUT/ab_v2.txt 2>&1
timeout 400 python profiles/r6_fwd_ab.py tp metric_1m_1080p dense_1m_1080p c2_dtu_300k_800x600 fullframe_1m_1080p c5_360_5m_1600x1200 > $OUT/ab_tp.txt 2>&1
grep MEAN $OUT/ab_v2.txt $OUT/ab_tp.txt
python profiles/r6_fwd_cmp.py v2 tp > $OUT/cmp_tp.txt 2>&1; tail -1 $OUT/cmp_tp.txt
rm -f $OUT/grad_report_k8.txt
VCR_GRAD_REPORT=$OUT/grad_report_k8.txt timeout 1100 python -m pytest tests -m gpu -q > $OUT/pytest_report_k8.txt 2>&1
tail -n 4 $OUT/pytest_report_k8.txt
