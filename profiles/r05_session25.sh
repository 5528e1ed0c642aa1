#!/bin/bash
# Round 5, session 25: 1024 closed loops as 1 - 4 fleets on handles and host threads of their own; the warm step from runs of two lengths
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s25; mkdir -p $O
timeout -k 10 600 python -X faulthandler tests/gpu_dev_fleet_parts.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/fleet_parts_1024.txt
