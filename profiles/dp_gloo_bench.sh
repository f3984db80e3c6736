#!/bin/bash
# Functional smoke of the multi-rank bench path on a ONE-GPU box: two ranks share the GPU, collectives travel over gloo
# (VCR_DIST_BACKEND=gloo).  The printed line exercises the N > 1 branch of bench.py -- exchange "factorised-deferred",
# views_per_s = 2 x iters_per_s -- its timing means nothing (two processes time-slice one GPU).
#   bash profiles/dp_gloo_bench.sh [workload] [steps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-c2_dtu_300k_800x600}; STEPS=${2:-10}
VCR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps $STEPS --warmup 3 --workload $WL --no-cpu-baseline --no-context
