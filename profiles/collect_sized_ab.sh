#!/bin/bash
# Instruction-fetch / wait counters and the launch time of one bench workload with the kernels compiled with the problem's sizes
# (default) against the shipped kernels (HMPC_JIT_SIZED=0) -- run through gpurun from the repo root:
#   bash profiles/collect_sized_ab.sh <tag> <bench args...>      e.g. n40p05 --workload cart_pole_n40 --frontier 2048 --frontier-kind random_prefix --p-one 0.5
set -e -o pipefail
TAG=${1:-ab}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*"
for V in sized shipped; do
    if [ $V = shipped ]; then export HMPC_JIT_SIZED=0; fi
    rm -rf $O/${TAG}_${V}_*
    timeout -k 10 240 $B > $O/${TAG}_${V}_bench.json 2> $O/${TAG}_${V}_bench.err
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/${TAG}_${V}_ic1 -- $B > $O/${TAG}_${V}_ic1.log 2>&1
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_${V}_ic2 -- $B > $O/${TAG}_${V}_ic2.log 2>&1
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O/${TAG}_${V}_ic3 -- $B > $O/${TAG}_${V}_ic3.log 2>&1
    python3 profiles/summarise.py pmc $O/${TAG}_${V}_pmc.json $O/${TAG}_${V}_ic1 $O/${TAG}_${V}_ic2 $O/${TAG}_${V}_ic3 > /dev/null
    python3 - <<PY
import json
d = json.load(open("$O/${TAG}_${V}_pmc.json"))
b = json.loads(open("$O/${TAG}_${V}_bench.json").read().strip().splitlines()[-1])
print("$V", "ms", round(b["ms_per_step"], 3), {k: (round(v["mean_per_launch"]) if isinstance(v, dict) and "mean_per_launch" in v else None) for k, v in d.items() if k != "_kernel"}, d["_kernel"]["kernel"][:60], "scratch", d["_kernel"]["scratch"])
PY
done
