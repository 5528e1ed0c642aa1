#!/bin/bash
# Round 5, session 13: the shift kernel with rows staged in LDS by global_load_lds (hmpc_shift_row_kernel) -- parity tests
# that go through the shift, then its rate against the register-staged kernel (HMPC_SHIFT_ROWS=0) and over waves per CU.
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s13; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fleet.py tests/test_reference_replay.py -q -m gpu -k "shift or warm_start or fleet or replay or closed" -p no:cacheprovider -x > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_shift.txt
[ $rc -ne 0 ] && exit 1
HMPC_SHIFT_ROWS=0 timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | sed 's/^/registers: /' | tee $O/shift_time.txt
for W in 16 12 8 6 4; do
  HMPC_SHIFT_ROW_WAVES=$W timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | sed "s/^/rows, at most $W waves: /" | tee -a $O/shift_time.txt
done
