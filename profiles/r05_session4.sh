#!/bin/bash
# Round 5, GPU session 4: A/B of compile-time switches on the headline (N = 20, one wave per node) and on N = 40 (two waves per node),
# same box, bench.py's own timed region (kernel_ms_avg by HIP events).
set -o pipefail
mkdir -p gpurun_out/r05_s4
one() { # label, workload, env...
    local label=$1 wl=$2; shift 2
    ( env "$@" timeout -k 10 400 python bench.py --workload $wl --no-secondary --no-cpu-baseline --steps 20 --warmup 3 ) > gpurun_out/r05_s4/$label.json 2> gpurun_out/r05_s4/$label.err
    python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r05_s4/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-28s %9.0f QP/s  kernel %.3f ms  its %.2f  undecided %d' % (sys.argv[1], d['value'], d['roofline']['kernel_ms_avg'], d['nodes']['ipm_iters_mean'], d['nodes']['not_converged']), flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200], flush=True)
PY
}
one n20_base cart_pole_n20 X=1
one n20_unsafe cart_pole_n20 HMPC_JIT_SAFE=0
one n20_nofence cart_pole_n20 HMPC_JIT_FLAGS=-DHMPC_NO_FENCE
one n20_base_again cart_pole_n20 X=1
one n40_base cart_pole_n40 X=1
one n40_dppfew cart_pole_n40 HMPC_JIT_FLAGS=-DHMPC_DPP_FEW
one n40_dppfew_nofence cart_pole_n40 "HMPC_JIT_FLAGS=-DHMPC_DPP_FEW -DHMPC_NO_FENCE"
one n40_nofence cart_pole_n40 HMPC_JIT_FLAGS=-DHMPC_NO_FENCE
