#!/bin/bash
# Round 5, GPU session 39: the tree as committed last -- the whole -m gpu suite and the default bench line once more; 1 / 2 / 4 / 8 fleets twice
# under HMPC_WAVES=1 (the sequence of the one unexplained core dump of session 20)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s39; mkdir -p $O
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=600 ) > $O/pytest_gpu.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "pytest running: $(tail -c 120 $O/pytest_gpu.txt | tr '\n' ' ')"; done
wait $PID; rc=$?; echo "pytest gpu: $rc"; tail -5 $O/pytest_gpu.txt | cut -c1-250
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench: $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s39/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'kinds', d['roofline']['kernel_kinds_1_2_4_waves'], 'ilp', d['roofline']['ilp_schedule_1_2_4_waves'])
print('parity_flags', d['parity_flags'], 'shift', d.get('warm_start_shift', {}).get('achieved_GBs'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'))
PY
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
for rep in 1 2; do HMPC_WAVES=1 timeout -k 10 200 python -X faulthandler tests/gpu_dev_fleet_parts.py 1024 > $O/parts_$rep.txt 2>&1; echo "parallel fleets rep $rep rc $? ($(grep -c 'steps/s' $O/parts_$rep.txt) lines)"; done
