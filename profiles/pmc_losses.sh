#!/bin/bash
# PMC counters of the image-loss and per-Gaussian tail kernels (not covered by collect_pmc.sh): bash profiles/pmc_losses.sh
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$name -o pmc -- \
        python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-context > $R/gpurun_out/pmcl_$name.log 2>&1
    python $R/profiles/summarize.py counters $(ls $R/gpurun_out/pmcl_$name/*counter_collection.csv | head -1) \
        $R/gpurun_out/r3_pmc_losses_$name.csv ssim normal_losses depth_normal geometry_step activate_ scale_reg > /dev/null
    rm -rf $R/gpurun_out/pmcl_$name
}
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM
run grbm GRBM_GUI_ACTIVE
cat $R/gpurun_out/r3_pmc_losses_*.csv | grep -i "ssim"
