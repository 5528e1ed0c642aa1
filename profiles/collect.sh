#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>      e.g. r01_v5
# rocprofv3 kernel stats and each PMC group in separate passes (the pool forbids mixing --pmc with
# other trace domains); outputs under gpurun_out/<tag>_*; condense afterwards with profiles/summarise.py.
set -e -o pipefail
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rm -rf $O/${TAG}_*
timeout -k 10 300 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $B > $O/${TAG}_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $B > $O/${TAG}_write.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_sq1 -- $B > $O/${TAG}_sq1.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${TAG}_sq2 -- $B > $O/${TAG}_sq2.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_sq3 -- $B > $O/${TAG}_sq3.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${TAG}_sq4 -- $B > $O/${TAG}_sq4.log 2>&1 || echo "sq4 group not available"
tail -c 600 $O/${TAG}_bench.json
