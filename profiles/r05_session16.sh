#!/bin/bash
# Round 5, session 16: hmpc_shift_row_kernel with 16-byte stores; non-temporal row fetch (nt1), stores (nt2), both (nt3)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s16; mkdir -p $O; rm -f $O/shift_nt.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fleet.py -q -m gpu -k "shift or warm_start or fleet" -p no:cacheprovider -x > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_shift.txt
[ $rc -ne 0 ] && exit 1
for L in libhmpc.so libhmpc_nt1.so libhmpc_nt2.so libhmpc_nt3.so; do
for W in 16 12 8; do
  HMPC_LIB=$L HMPC_JIT_SELFCHECK=0 HMPC_SHIFT_ROW_WAVES=$W timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | grep -v " 4096 " | sed "s/^/waves $W: /" | tee -a $O/shift_nt.txt
done; done
