#!/bin/bash
# Round 5, session 33: the end-game fraction to the boundary -- early check on the headline problem's ILP-scheduled binaries
# (validation against the oracle, then the rate) before the whole cache is rebuilt
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s33; mkdir -p $O
VAL_ONE="('cart_pole_with_walls', 20, True)" timeout -k 10 400 python tests/gpu_validate_ilp.py 2>&1 | grep RESULT | cut -c1-600 | tee $O/validate_headline.txt
HMPC_JIT_SCHED=iterative-ilp timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_ilp.json 2> $O/bench_ilp.err; echo "bench rc $?"
python - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_s33/bench_ilp.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'nodes', d['nodes'], 'kinds', d['roofline']['kernel_kinds_1_2_4_waves'], 'ilp', d['roofline']['ilp_schedule_1_2_4_waves'], 'parity', d.get('parity_flags'))
PY
