#!/bin/bash
# Round 5, GPU session 40: the default bench line with the fleets' warm steps timed inside one run
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s40; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fleet.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
( timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench: $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s40/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_flags']['ok'], 'shift', d.get('warm_start_shift', {}).get('achieved_GBs'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'), v.get('warm_step_ms'), v.get('cold_step_ms'))
PY
