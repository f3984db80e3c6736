#!/bin/bash
# PMC passes over the compositing kernels with the forward A/B driver as the workload (metric scene, 4 cameras, both list forms):
#   bash profiles/r6_pmc_fwd.sh <tag> <lib.so> [sq|lds|mem ...]     -> gpurun_out/r6_pmc_<tag>_<set>.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    AB_DUMP=0 AB_REPS=3 AB_CAMS=2 VCR_LIB=$lib rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$name -o pmc -- \
        python $R/profiles/r6_fwd_ab.py $tag metric_1m_1080p > /tmp/pmc_${tag}_$name.log 2>&1
    python $R/profiles/summarize.py counters $(ls /tmp/pmc_${tag}_$name/*counter_collection.csv /tmp/pmc_${tag}_$name/*/*counter_collection.csv 2>/dev/null | head -1) \
        $R/gpurun_out/r6_pmc_${tag}_$name.csv composite_ > /dev/null
    rm -rf /tmp/pmc_${tag}_$name
}
for set in "$@"; do
    case $set in
        sq)   run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY ;;
        lds)  run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM ;;
        thr)  run thr SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH ;;
        grbm) run grbm GRBM_GUI_ACTIVE ;;
    esac
done
ls $R/gpurun_out/r6_pmc_${tag}_*
