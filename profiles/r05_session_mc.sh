#!/bin/bash
# Round 5: the four-sigma closed-loop study on the fleet driver on the FINAL kernels of round 5 (paired solve, one feature set, validated
# schedules), WITHOUT the parent -> child hand-down (the setting of the published runs) -> gpurun_out/mc_r05 (copied to profiles/mc_r05).
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
mkdir -p gpurun_out/mc_r05
for SD in 0.000 0.001 0.003; do
  timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd $SD --width 1 --no-handdown --out gpurun_out/mc_r05 > gpurun_out/mc_r05/summary_sd_$SD.txt 2>&1; echo "mc $SD rc $?"
done
timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 109 --steps 50 --sd 0.010 --width 1 --no-handdown --out gpurun_out/mc_r05 > gpurun_out/mc_r05/summary_sd_0.010.txt 2>&1; echo "mc 0.010 rc $?"
cat gpurun_out/mc_r05/summary_sd_*.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ilp_schedule_is_only" -p no:cacheprovider > gpurun_out/mc_r05/pytest_recipe.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/mc_r05/pytest_recipe.txt
