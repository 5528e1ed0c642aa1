#!/bin/bash
# Round 5, session 30: MPC steps/s against the number of closed loops in the fleet
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s30; mkdir -p $O; rm -f $O/fleet_steps_by_K.txt
for K in 1024 4096; do
  timeout -k 10 500 python tests/gpu_dev_fleet_steps.py $K 2>&1 | grep -v amdgpu.ids | tail -5 | sed "s/^/K $K: /" | tee -a $O/fleet_steps_by_K.txt
done
