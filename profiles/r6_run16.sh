#!/bin/bash
# round 6, call 16: l1 + ssim kernels with the conflict-free LDS layout (halo stride 45, row-sum planes stored [o][j]) against the old layout
# (libvcr_raster_ssimold.so: the default objects with losses.o compiled from the committed losses.hip; libvcr_raster.so: the re-laid-out
#  kernels, which were reverted after this call -- profiles/r6_ssim_lds_layout_negative.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run16
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for k in 1 2; do
VCR_LIB=$R/vcr_gaus_amd/libvcr_raster_ssimold.so timeout 100 python profiles/time_ssim.py
timeout 100 python profiles/time_ssim.py
done | tee $OUT/time_ssim.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_train_step_gpu.py -m gpu -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed" $OUT/pytest.txt
cd /tmp
for lib in ssimold new; do
  L=$R/vcr_gaus_amd/libvcr_raster.so; [ $lib == ssimold ] && L=$R/vcr_gaus_amd/libvcr_raster_ssimold.so
  VCR_LIB=$L timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pl_$lib -o pmc -- python $R/profiles/time_ssim.py > /tmp/pl_$lib.log 2>&1
  python $R/profiles/summarize.py counters $(ls /tmp/pl_$lib/*counter_collection.csv /tmp/pl_$lib/*/*counter_collection.csv 2>/dev/null | head -1) $OUT/pmc_lds_$lib.csv ssim > /dev/null
  cat $OUT/pmc_lds_$lib.csv
done
