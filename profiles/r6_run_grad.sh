#!/bin/bash
# round 6: the whole GPU suite in REPORT mode (tests/util.py::_report logs every gradient comparison instead of asserting) on the
# mixed-precision projection backward, with the oracle's fragility constant K = 16 (round 5) and K = 4 (round 6 sweep)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_grad
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for K in 4 16; do
    rm -f $OUT/grad_report_mixed_k$K.txt
    VCR_LIB=$R/vcr_gaus_amd/libvcr_raster_mixed.so VCR_TEST_FRAGILE_K=$K VCR_GRAD_REPORT=$OUT/grad_report_mixed_k$K.txt \
        timeout 1100 python -m pytest tests -m gpu -q > $OUT/pytest_mixed_k$K.txt 2>&1
    tail -n 4 $OUT/pytest_mixed_k$K.txt
done
