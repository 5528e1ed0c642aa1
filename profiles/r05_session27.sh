#!/bin/bash
# Round 5, session 27: how far the HIP path's searches are from the reference's traces (tests/test_bb_traces.py allows +-3)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s27; mkdir -p $O
timeout -k 10 600 python tests/gpu_dev_bb_deviation.py 2>&1 | grep -v amdgpu.ids | tee $O/bb_deviation.txt
HMPC_JIT_SCHED=default timeout -k 10 600 python tests/gpu_dev_bb_deviation.py 2>&1 | grep -v amdgpu.ids | sed 's/^/default schedule: /' | tee -a $O/bb_deviation.txt
