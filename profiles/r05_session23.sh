#!/bin/bash
# Round 5, session 23: wall time of every step of the fleet of 1024 loops
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s23; mkdir -p $O
timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/fleet_steps_1024.txt
