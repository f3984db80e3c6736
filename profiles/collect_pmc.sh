#!/bin/bash
# PMC passes for the compositing / sort kernels on the metric workload (run on the GPU box: bash profiles/collect_pmc.sh sq lds grbm fetch write).
# One rocprofv3 run per counter set (--pmc is never combined with API/runtime tracing); summaries land in gpurun_out/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${VCR_ROUND:-r3}
run() {  # name counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- \
        python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-context > $R/gpurun_out/pmc_$name.log 2>&1
    python $R/profiles/summarize.py counters $(ls $R/gpurun_out/pmc_$name/*counter_collection.csv | head -1) \
        $R/gpurun_out/${RND}_pmc_$name.csv composite_ adam_kernel sh_ rs_ duplicate_ preprocess_ > /dev/null
    # which frame the counters saw: R / R' / V of the bench line printed by the same command
    python - "$R/gpurun_out/pmc_$name.log" "$R/gpurun_out/${RND}_pmc_meta.json" "$name" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)["config"]
try:
    meta = json.load(open(sys.argv[2]))
except Exception:
    meta = {}
meta[sys.argv[3]] = {k: d.get(k) for k in ("workload", "tile_instances_R", "emitted_instances", "visible_V")}
json.dump(meta, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
    rm -rf $R/gpurun_out/pmc_$name
}
for set in "$@"; do
    case $set in
        sq)    run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY ;;
        lds)   run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM ;;
        grbm)  run grbm GRBM_GUI_ACTIVE ;;
        fetch) run fetch FETCH_SIZE ;;
        write) run write WRITE_SIZE ;;
    esac
done
