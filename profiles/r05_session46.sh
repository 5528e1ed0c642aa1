#!/bin/bash
# Round 5, session 46: does the hand-out order (hmpc_order_kernel: 0.1 ms of one workgroup per launch) pay in the fleet's launches?
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s46; mkdir -p $O; rm -f $O/fleet_order.txt
for rep in 1 2 3; do
  timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -o "host phases.*\|warm steps/s over.*" | tail -2 | tr '\n' ' ' | sed "s/^/ordered:   /" | tee -a $O/fleet_order.txt; echo | tee -a $O/fleet_order.txt
  HMPC_NO_ORDER=1 timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -o "host phases.*\|warm steps/s over.*" | tail -2 | tr '\n' ' ' | sed "s/^/unordered: /" | tee -a $O/fleet_order.txt; echo | tee -a $O/fleet_order.txt
done
