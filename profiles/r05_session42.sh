#!/bin/bash
# Round 5, session 42: the run-time option polish_tol (from which residual the polish is attempted) against the rates
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s42; mkdir -p $O
timeout -k 10 800 python tests/gpu_dev_ptol.py 2>&1 | grep -v amdgpu.ids | tee $O/ptol.txt
