#!/bin/bash
# Round 5, session 17: hmpc_shift_row_kernel, M_mu laid out by hmpc_set_shift_maps -- the suites that go through the shift, its rate
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s17; mkdir -p $O; rm -f $O/shift_time.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fleet.py tests/test_reference_replay.py tests/test_capi.py -q -m gpu -k "shift or warm_start or fleet or replay or closed" -p no:cacheprovider -x > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_shift.txt
[ $rc -ne 0 ] && exit 1
HMPC_SHIFT_ROWS=0 timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | sed 's/^/registers: /' | tee $O/shift_time.txt
timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | sed "s/^/rows in LDS: /" | tee -a $O/shift_time.txt
