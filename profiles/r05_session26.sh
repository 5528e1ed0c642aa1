#!/bin/bash
# Round 5, session 26: the fleet driver's per-tree loops on the host pool (hmpc_pool.h): test_fleet, then the steps of 1024 loops by thread count
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s26; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fleet.py tests/test_reference_replay.py -q -m gpu -p no:cacheprovider -x > $O/pytest_fleet.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_fleet.txt
[ $rc -ne 0 ] && exit 1
for N in 1 2 4 8 12; do
  echo "HMPC_HOST_THREADS=$N" | tee -a $O/fleet_steps_threads.txt
  HMPC_HOST_THREADS=$N timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/fleet_steps_threads.txt
done
