#!/bin/bash
# round 6, call 11: the visibility batch under its two knobs (cameras in flight, 8x8-cell lists)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run11
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for cfg in "0 0" "16 0" "4 0" "0 1" "16 1"; do
  set -- $cfg
  echo "== inflight $1 quad_lists $2"
  VCR_VIS_INFLIGHT=$1 VCR_VIS_QUAD=$2 timeout 200 python profiles/r6_visibility_trace.py 3 2>&1 | grep "visibility batch\|first 16" | tail -4
done | tee $OUT/visibility_ab.txt
