#!/bin/bash
# Instruction-cache counters of the headline kernel for one or more build variants (run through gpurun from the repo root):
#   bash profiles/collect_icache.sh <tag> libhmpc.so [libhmpc_X.so ...]
set -e -o pipefail
TAG=${1:-ic}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
for L in "$@"; do
    export HMPC_LIBRARY_NAME=$L
    N=${L%.so}
    rm -rf $O/${TAG}_${N}_*
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/${TAG}_${N}_ic1 -- $B > $O/${TAG}_${N}_ic1.log 2>&1
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_${N}_ic2 -- $B > $O/${TAG}_${N}_ic2.log 2>&1
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/${TAG}_${N}_ic3 -- $B > $O/${TAG}_${N}_ic3.log 2>&1 || echo "ic3 group not available"
    python3 profiles/summarise.py pmc $O/${TAG}_${N}_icache.json $O/${TAG}_${N}_ic1 $O/${TAG}_${N}_ic2 $O/${TAG}_${N}_ic3 > /dev/null || python3 profiles/summarise.py pmc $O/${TAG}_${N}_icache.json $O/${TAG}_${N}_ic1 $O/${TAG}_${N}_ic2 > /dev/null
    echo "$L done"
done
