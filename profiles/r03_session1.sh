#!/bin/bash
# Round 3, GPU session 1 (run through gpurun from the repo root: bash profiles/r03_session1.sh):
# the -m gpu suite (with the replay of the whole published study on the fleet driver), the default bench line, and the
# four-sigma closed-loop study on the fleet driver (monte_carlo.py -> gpurun_out/mc_r03, copied to profiles/mc_r03).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O gpurun_out/mc_r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu_1.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_1.log
tail -4 $O/pytest_gpu_1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_1.json 2> $O/bench_1.err; echo "bench rc $?"
for SD in 0.000 0.001 0.003; do
  timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd $SD --width 1 --out gpurun_out/mc_r03 > gpurun_out/mc_r03/summary_sd_$SD.txt 2>&1; echo "mc $SD rc $?"
done
timeout -k 10 400 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 109 --steps 50 --sd 0.010 --width 1 --out gpurun_out/mc_r03 > gpurun_out/mc_r03/summary_sd_0.010.txt 2>&1; echo "mc 0.010 rc $?"
cat gpurun_out/mc_r03/summary_sd_*.txt
