#!/bin/bash
# Round 5, session 20: 1024 closed loops as 1 / 2 / 4 / 8 fleets on handles and host threads of their own
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s20; mkdir -p $O
timeout -k 10 600 python tests/gpu_dev_fleet_parts.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/fleet_parts_1024.txt
HMPC_WAVES=1 timeout -k 10 600 python tests/gpu_dev_fleet_parts.py 1024 2>&1 | grep -v amdgpu.ids | tee -a $O/fleet_parts_1024.txt
