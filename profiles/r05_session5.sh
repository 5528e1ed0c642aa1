#!/bin/bash
# Round 5, GPU session 5: A/B of compile-time switches on the headline (N = 20, one wave per node) and on N = 40 (two waves per node),
# same box, bench.py's own timed region (kernel_ms_avg by HIP events).
set -o pipefail
mkdir -p gpurun_out/r05_s5
one() { # label, workload, env...
    local label=$1 wl=$2; shift 2
    ( env "$@" timeout -k 10 400 python bench.py --workload $wl --no-secondary --no-cpu-baseline --steps 20 --warmup 3 ) > gpurun_out/r05_s5/$label.json 2> gpurun_out/r05_s5/$label.err
    python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r05_s5/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-28s %9.0f QP/s  kernel %.3f ms  its %.2f  undecided %d' % (sys.argv[1], d['value'], d['roofline']['kernel_ms_avg'], d['nodes']['ipm_iters_mean'], d['nodes']['not_converged']), flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200], flush=True)
PY
}
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "frontier_parity or every_kernel or register_kernel or golden or boundary" ) > gpurun_out/r05_s5/pytest_subset.txt 2>&1
rc=$?; echo "parity subset with the paired solve: $rc"; tail -5 gpurun_out/r05_s5/pytest_subset.txt
[ $rc -eq 0 ] || exit $rc
one n20_base cart_pole_n20 X=1
one n20_nopair cart_pole_n20 HMPC_JIT_FLAGS=-DHMPC_PAIR=0
one n20_base_again cart_pole_n20 X=1
one n40_base cart_pole_n40 X=1
one n40_nopair cart_pole_n40 HMPC_JIT_FLAGS=-DHMPC_PAIR=0
# the N > 1 line, rehearsed: two ranks on this one GPU under gloo (not a measurement): both halves of the metric, rccl_ranks
( timeout -k 10 900 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 5 --warmup 1 --no-cpu-baseline ) > gpurun_out/r05_s5/rehearsal_two_ranks.json 2> gpurun_out/r05_s5/rehearsal_two_ranks.err
echo "rehearsal: $?"; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05_s5/rehearsal_two_ranks.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'n_gpus', 'scaling', 'rccl_ranks')}, d.get('mpc_steps_per_sec'), d.get('configs2_strong_scaling_1024', {}).get('qp_per_s'), d.get('parity_flags'))
except Exception as e:
    print('rehearsal FAILED', repr(e)[:300]); print(open('gpurun_out/r05_s5/rehearsal_two_ranks.err').read()[-1500:])
PY
