#!/bin/bash
# Round 5, session 21: the core dump of session 20 (8 fleets on 8 host threads, HMPC_WAVES=1, after 30 earlier calls in the process)
# once more with the fault handler on
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s21; mkdir -p $O
HMPC_WAVES=1 timeout -k 10 300 python -X faulthandler tests/gpu_dev_fleet_parts.py 1024 > $O/parts_w1.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/parts_w1.txt | tail -60
