#!/bin/bash
# Kernel stats and matrix-core counters of another BASELINE config (run through gpurun from the repo root):
#   bash profiles/collect_config.sh <tag> <workload> <frontier>     e.g. r02_c4 random_mld 1024 ; r02_n40 cart_pole_n40 2048
set -e -o pipefail
TAG=${1:-cfg}; WL=${2:-random_mld}; FR=${3:-1024}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
# (the first-use check of the compiled kernels would add one 6-node launch per kernel to the per-kernel averages)
export HMPC_JIT_SELFCHECK=0
B="python3 bench.py --workload $WL --frontier $FR --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rm -rf $O/${TAG}_*
timeout -k 10 300 $B > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_mfma -- $B > $O/${TAG}_mfma.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/${TAG}_sq -- $B > $O/${TAG}_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $B > $O/${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $B > $O/${TAG}_write.log 2>&1
tail -c 400 $O/${TAG}_bench.json
