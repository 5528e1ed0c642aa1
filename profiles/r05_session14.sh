#!/bin/bash
# Round 5, session 14: hmpc_shift_row_kernel, second form (identifier stores after the wait, mat-vec in batches of eight)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s14; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "shift or warm_start" -p no:cacheprovider -x > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_shift.txt
[ $rc -ne 0 ] && exit 1
for W in 16 8 4; do
  HMPC_SHIFT_ROW_WAVES=$W timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | sed "s/^/rows, at most $W waves: /" | tee -a $O/shift_time.txt
done
