#!/bin/bash
# Round 5, GPU session 9 (the tree with its VALIDATED manifest): the whole -m gpu suite, A/B of the paired solve and of the two schedules on
# the same box, the default bench line, the N > 1 line rehearsed with two ranks on this one GPU.
set -o pipefail
mkdir -p gpurun_out/r05_s9
( timeout -k 10 1500 python -m pytest tests -m gpu -x -q --timeout=600 ) > gpurun_out/r05_s9/pytest_gpu.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "pytest running: $(tail -c 120 gpurun_out/r05_s9/pytest_gpu.txt | tr '\n' ' ')"; done
wait $PID; rc=$?; echo "pytest gpu: $rc"; tail -6 gpurun_out/r05_s9/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
one() { # label, workload, env...
    local label=$1 wl=$2; shift 2
    ( env "$@" timeout -k 10 400 python bench.py --workload $wl --no-secondary --no-cpu-baseline --steps 20 --warmup 3 ) > gpurun_out/r05_s9/$label.json 2> gpurun_out/r05_s9/$label.err
    python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r05_s9/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-30s %9.0f QP/s  kernel %.3f ms  its %.2f  undecided %d  kinds %s ilp %s' % (sys.argv[1], d['value'], d['roofline']['kernel_ms_avg'], d['nodes']['ipm_iters_mean'], d['nodes']['not_converged'],
          d['roofline']['kernel_kinds_1_2_4_waves'], d['roofline']['ilp_schedule_1_2_4_waves']), flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200], flush=True)
PY
}
one n20_default cart_pole_n20 X=1
one n20_single_solves cart_pole_n20 HMPC_JIT_FLAGS=-DHMPC_PAIR=0 HMPC_JIT_SCHED=iterative-ilp
one n20_default_schedule cart_pole_n20 HMPC_JIT_SCHED=default
one n20_default_again cart_pole_n20 X=1
one n40_default cart_pole_n40 X=1
one n40_single_solves cart_pole_n40 HMPC_JIT_FLAGS=-DHMPC_PAIR=0 HMPC_JIT_SCHED=iterative-ilp
one n40_default_schedule cart_pole_n40 HMPC_JIT_SCHED=default
( timeout -k 10 900 python bench.py ) > gpurun_out/r05_s9/bench.json 2> gpurun_out/r05_s9/bench.err
echo "bench: $?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_s9/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'nodes', d['nodes'], 'kinds', d['roofline']['kernel_kinds_1_2_4_waves'], 'ilp', d['roofline']['ilp_schedule_1_2_4_waves'])
print('parity_flags', d['parity_flags'])
for k, v in d.get('frontiers', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'))
for k, v in d.get('other_configs', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('qp_per_s'), v.get('not_converged'), v.get('statuses_equal'), v.get('speedup'))
for k, v in d.get('mpc_steps_per_sec', {}).items():
    if isinstance(v, dict): print(' ', k, v.get('value'))
PY
( timeout -k 10 900 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 5 --warmup 1 --no-cpu-baseline ) > gpurun_out/r05_s9/rehearsal_two_ranks.json 2> gpurun_out/r05_s9/rehearsal_two_ranks.err
echo "rehearsal: $?"; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05_s9/rehearsal_two_ranks.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'n_gpus', 'scaling', 'rccl_ranks')}, d.get('mpc_steps_per_sec'), d.get('configs2_strong_scaling_1024', {}).get('qp_per_s'), d.get('parity_flags'))
except Exception as e:
    print('rehearsal FAILED', repr(e)[:300]); print(open('gpurun_out/r05_s9/rehearsal_two_ranks.err').read()[-1500:])
PY
