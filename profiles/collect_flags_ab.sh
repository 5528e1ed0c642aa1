#!/bin/bash
# Compiler flags of the kernels compiled at hmpc_create (each argument: one value of HMPC_JIT_FLAGS), three workloads:
#   bash profiles/collect_flags_ab.sh "<flags>" ...        (through gpurun from the repo root)
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
for V in "" "$@"; do
  export HMPC_JIT_FLAGS="$V"
  for W in "" "--workload cart_pole_n40 --frontier 2048" "--workload random_mld --frontier 4096 --steps 2 --warmup 1"; do
    S=$(date +%s.%N)
    timeout -k 10 500 $B $W 2> gpurun_out/t.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else {'value':0,'ms_per_step':0}
print('%-60s %-24s %9.0f QP/s %8.3f ms' % ('''$V'''[-60:] or '(default flags)', '''$W'''[11:35] or 'headline', d['value'], d['ms_per_step']), end='')"
    E=$(date +%s.%N)
    echo "   (run $(python -c "print(round($E-$S,1))") s)"
  done
done
