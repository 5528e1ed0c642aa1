#!/bin/bash
# Round 4, GPU session 5: the node on the boundary of feasibility, the sd = .003 study without hand-down again, the
# bounds-checked build (gpu_check_build.py), the whole -m gpu suite.
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r04
mkdir -p $O gpurun_out/mc_r04
timeout -k 10 300 python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz --sims 100 --steps 50 --sd 0.003 --width 1 --no-handdown --out gpurun_out/mc_r04 > gpurun_out/mc_r04/summary_sd_0.003.txt 2>&1; echo "mc 0.003 rc $?"; tail -8 gpurun_out/mc_r04/summary_sd_0.003.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_5.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest_gpu_5.log
HMPC_LIBRARY_NAME=libhmpc_check.so timeout -k 10 900 python tests/gpu_check_build.py > $O/check_build.txt 2>&1; echo "check rc $?"; tail -5 $O/check_build.txt
