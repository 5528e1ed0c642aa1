#!/bin/bash
# Round 3: the four-sigma closed-loop study once more on the FINAL tree (fleet driver with the parent -> child hand-down,
# terminal-set path, weak exit) -> gpurun_out/mc_r03 (copied to profiles/mc_r03).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/mc_r03
for SD in 0.000 0.001 0.003; do
  timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd $SD --width 1 --out gpurun_out/mc_r03 > gpurun_out/mc_r03/summary_sd_$SD.txt 2>&1; echo "mc $SD rc $?"
done
timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 109 --steps 50 --sd 0.010 --width 1 --out gpurun_out/mc_r03 > gpurun_out/mc_r03/summary_sd_0.010.txt 2>&1; echo "mc 0.010 rc $?"
grep -h -v amdgpu gpurun_out/mc_r03/summary_sd_*.txt
