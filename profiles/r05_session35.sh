#!/bin/bash
# Round 5, GPU session 35 (the tree with the end-game fraction, its VALIDATED manifest in place): the whole -m gpu suite, the default bench
# line, the closed-loop study, the N > 1 line rehearsed with two ranks on this one GPU.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s35; mkdir -p $O
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=600 ) > $O/pytest_gpu.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "pytest running: $(tail -c 120 $O/pytest_gpu.txt | tr '\n' ' ')"; done
wait $PID; rc=$?; echo "pytest gpu: $rc"; tail -40 $O/pytest_gpu.txt | cut -c1-250
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench: $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s35/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'nodes', d['nodes'], 'kinds', d['roofline']['kernel_kinds_1_2_4_waves'], 'ilp', d['roofline']['ilp_schedule_1_2_4_waves'])
print('parity_flags', d['parity_flags'])
print('shift', d.get('warm_start_shift', {}).get('achieved_GBs'))
for k, v in d.get('frontiers', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'), v.get('ipm_iters_mean'))
for k, v in d.get('other_configs', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'), v.get('statuses_equal'), v.get('speedup'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'), v.get('per_warm_step'))
PY
