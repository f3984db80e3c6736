#!/bin/bash
# round 6: the two constants of the grouped backward's choice between the row-packed and the whole-quad loop (rows_bias, rows_pair_cost;
# (20, 2) from round 3's per-chunk sweep), swept again now that a group holds ~4x the survivors of a chunk.  Variant build -DVCR_ROWS_ENV.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_rows_sweep
mkdir -p $OUT
cd $R
export TMPDIR=/tmp AB_REPS=6 AB_CAMS=4 AB_DUMP=0 VCR_LIB=$R/vcr_gaus_amd/libvcr_raster_rowsenv.so
for cfg in "20 2" "12 2" "16 2" "24 2" "28 2" "36 2" "20 0" "20 4" "28 4" "64 0" "0 0"; do
  set -- $cfg
  VCR_ROWS_BIAS=$1 VCR_ROWS_PAIR=$2 timeout 300 python profiles/r6_fwd_ab.py b$1p$2 metric_1m_1080p c5_360_5m_1600x1200 dense_1m_1080p fullframe_1m_1080p 2>&1 | grep MEAN
done | tee $OUT/sweep.txt
