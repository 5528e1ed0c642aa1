#!/bin/bash
# Instruction-fetch and wait counters of the streaming kernel on BASELINE configs[4] (run through gpurun from the repo root):
#   bash profiles/collect_c4_icache.sh <tag> <library name>
set -e -o pipefail
TAG=${1:-c4ic}; L=${2:-libhmpc.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export HMPC_LIBRARY_NAME=$L DBG_PARITY=8
B="python3 tests/gpu_dev_cfg4.py"
rm -rf $O/${TAG}_ic*
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/${TAG}_ic1 -- $B > $O/${TAG}_ic1.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_ic2 -- $B > $O/${TAG}_ic2.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_ic3 -- $B > $O/${TAG}_ic3.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/${TAG}_ic4 -- $B > $O/${TAG}_ic4.log 2>&1 || echo "ic4 group not available"
python3 profiles/summarise.py pmc $O/${TAG}_icache.json --grid 65536 $O/${TAG}_ic1 $O/${TAG}_ic2 $O/${TAG}_ic3 $O/${TAG}_ic4 > /dev/null || python3 profiles/summarise.py pmc $O/${TAG}_icache.json --grid 65536 $O/${TAG}_ic1 $O/${TAG}_ic2 $O/${TAG}_ic3 > /dev/null
echo "$L done"
