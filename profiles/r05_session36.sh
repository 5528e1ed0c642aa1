#!/bin/bash
# Round 5, GPU session 36: evidence on the kernels with the end-game fraction -- the randomized kernel <-> oracle sweeps (N = 20,
# N = 40), the shape sweeps under the default recipe, the closed-loop study without the hand-down, fleet steps, the N > 1 line rehearsed
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s36; mkdir -p $O gpurun_out/mc_r05
( DBG_REPS=24 timeout -k 10 500 python tests/gpu_parity_sweep.py ) > $O/parity_sweep_n20.txt 2>&1; echo "sweep n20: $?"; tail -6 $O/parity_sweep_n20.txt | cut -c1-250
( DBG_REPS=8 DBG_T=40 DBG_SKIP_WIDE=1 timeout -k 10 400 python tests/gpu_parity_sweep.py ) > $O/parity_sweep_n40.txt 2>&1; echo "sweep n40: $?"; tail -4 $O/parity_sweep_n40.txt | cut -c1-250
( timeout -k 10 600 python tests/gpu_sized_shapes.py ) > $O/sized_shapes.txt 2>&1; echo "shapes: $?"; tail -3 $O/sized_shapes.txt | cut -c1-200
( DBG_EDGE=1 timeout -k 10 600 python tests/gpu_sized_shapes.py ) > $O/sized_shapes_edge.txt 2>&1; echo "edge shapes: $?"; tail -3 $O/sized_shapes_edge.txt | cut -c1-200
for SD in 0.000 0.001 0.003; do
  timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd $SD --width 1 --no-handdown --out gpurun_out/mc_r05 > gpurun_out/mc_r05/summary_sd_$SD.txt 2>&1; echo "mc $SD rc $?"
done
timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 109 --steps 50 --sd 0.010 --width 1 --no-handdown --out gpurun_out/mc_r05 > gpurun_out/mc_r05/summary_sd_0.010.txt 2>&1; echo "mc 0.010 rc $?"
grep -h "warm/cold cost\|solves/step\|left the feasible" gpurun_out/mc_r05/summary_sd_*.txt | cut -c1-200
for rep in 1 2 3; do timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -o "host phases.*\|warm steps/s over.*" | tail -2 | tr '\n' ' '; echo; done | tee $O/fleet_steps.txt
( timeout -k 10 900 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 5 --warmup 1 --no-cpu-baseline ) > $O/rehearsal_two_ranks.json 2> $O/rehearsal_two_ranks.err
echo "rehearsal: $?"; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05_s36/rehearsal_two_ranks.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'n_gpus', 'scaling', 'rccl_ranks')}, d.get('mpc_steps_per_sec'), d.get('configs2_strong_scaling_1024', {}).get('qp_per_s'), d.get('parity_flags'))
except Exception as e:
    print('rehearsal FAILED', repr(e)[:300]); print(open('gpurun_out/r05_s36/rehearsal_two_ranks.err').read()[-1500:])
PY
