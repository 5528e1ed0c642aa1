#!/bin/bash
# Round 5, session 18: both shift kernels on odd shapes
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s18; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "shift" -p no:cacheprovider > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest_shift.txt
