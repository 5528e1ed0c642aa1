#!/bin/bash
# Round 5, GPU session 11: can the compiler's DEFAULT scheduler be told that these kernels run one wave per SIMD (amdgpu_waves_per_eu(1,1))?
set -o pipefail
mkdir -p gpurun_out/r05_s11
one() { # label, workload, env...
    local label=$1 wl=$2; shift 2
    ( env "$@" timeout -k 10 400 python bench.py --workload $wl --no-secondary --no-cpu-baseline --steps 20 --warmup 3 ) > gpurun_out/r05_s11/$label.json 2> gpurun_out/r05_s11/$label.err
    python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r05_s11/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print('%-34s %9.0f QP/s  kernel %.3f ms  its %.2f  undecided %d  ilp %s' % (sys.argv[1], d['value'], d['roofline']['kernel_ms_avg'], d['nodes']['ipm_iters_mean'], d['nodes']['not_converged'],
          d['roofline']['ilp_schedule_1_2_4_waves']), flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200], flush=True)
PY
}
one n20_default_schedule cart_pole_n20 HMPC_JIT_SCHED=default
one n20_default_schedule_w1 cart_pole_n20 HMPC_JIT_SCHED=default "HMPC_JIT_FLAGS=-DHMPC_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(1,1)))"
one n20_default_schedule_bias0 cart_pole_n20 HMPC_JIT_SCHED=default "HMPC_JIT_FLAGS=-mllvm -amdgpu-schedule-metric-bias=0"
one n20_maxilp cart_pole_n20 HMPC_JIT_SCHED=max-ilp
one n20_validated_ilp cart_pole_n20 X=1
