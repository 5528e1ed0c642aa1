#!/bin/bash
# Round 4: the four-sigma closed-loop study on the fleet driver, final kernels, WITHOUT the parent -> child hand-down (the
# setting of the reference's published runs) -> gpurun_out/mc_r04 (copied to profiles/mc_r04).
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
mkdir -p gpurun_out/mc_r04
for SD in 0.000 0.001 0.003; do
  timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd $SD --width 1 --no-handdown --out gpurun_out/mc_r04 > gpurun_out/mc_r04/summary_sd_$SD.txt 2>&1; echo "mc $SD rc $?"
done
timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 109 --steps 50 --sd 0.010 --width 1 --no-handdown --out gpurun_out/mc_r04 > gpurun_out/mc_r04/summary_sd_0.010.txt 2>&1; echo "mc 0.010 rc $?"
cat gpurun_out/mc_r04/summary_sd_*.txt
