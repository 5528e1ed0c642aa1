#!/bin/bash
# Round 5, GPU session 41: more of the same evidence on the final kernels -- the closed-loop study WITH the parent -> child hand-down (the
# product's default), and the randomized kernel <-> oracle sweep at four times the size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s41; mkdir -p $O gpurun_out/mc_r05_handdown
for SD in 0.001 0.003 0.010; do
  S=100; [ $SD = 0.010 ] && S=109
  timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims $S --steps 50 --sd $SD --width 1 --out gpurun_out/mc_r05_handdown > gpurun_out/mc_r05_handdown/summary_sd_$SD.txt 2>&1; echo "mc (hand-down) $SD rc $?"
done
grep -h "hand-down\|warm/cold cost\|solves/step\|left the feasible\|cover size" gpurun_out/mc_r05_handdown/summary_sd_*.txt | cut -c1-200
( DBG_REPS=96 DBG_SKIP_WIDE=1 timeout -k 10 900 python tests/gpu_parity_sweep.py ) > $O/parity_sweep_n20_96.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "sweep running: $(tail -c 160 $O/parity_sweep_n20_96.txt | tr '\n' ' ')"; done
wait $PID; echo "sweep: $?"; tail -3 $O/parity_sweep_n20_96.txt | cut -c1-260
