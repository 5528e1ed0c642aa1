#!/bin/bash
# Round 3 evidence for the default bench line (real-tree frontier, each node solved cold), run through gpurun from the
# repo root:   bash profiles/collect_r03.sh <tag>      e.g. r03_v16
# rocprofv3 kernel stats and each PMC group in separate passes (the pool forbids mixing --pmc with other trace domains);
# outputs under gpurun_out/<tag>_*; condense afterwards with profiles/summarise.py (pmc ... --grid 65536).
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rm -rf $O/${TAG}_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B > $O/${TAG}_stats.log 2>&1; echo "stats rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_handdown -- $B --handdown > $O/${TAG}_stats_handdown.log 2>&1; echo "stats handdown rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $B > $O/${TAG}_fetch.log 2>&1; echo "fetch rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $B > $O/${TAG}_write.log 2>&1; echo "write rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_sq1 -- $B > $O/${TAG}_sq1.log 2>&1; echo "sq1 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${TAG}_sq2 -- $B > $O/${TAG}_sq2.log 2>&1; echo "sq2 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_sq3 -- $B > $O/${TAG}_sq3.log 2>&1; echo "sq3 rc $?"
timeout -k 10 300 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc $?"
tail -c 400 $O/${TAG}_bench.json
