#!/bin/bash
# Round 5, session 19: the fleet of 1024 loops over frontier width and speculation (the step is bound by the longest node of each launch)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s19; mkdir -p $O
timeout -k 10 900 python tests/gpu_dev_fleet_sweep.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/fleet_sweep_1024.txt
