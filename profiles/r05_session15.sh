#!/bin/bash
# Round 5, session 15/16: hmpc_shift_row_kernel with the per-tree pass -- parity, then the phases knocked out one at a time
# (diagnostic builds libhmpc_ko<bits>.so: -DSHIFT_KO, 2 = no mapped block, 4 = no copy, 7 = neither, and no row fetch)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s15; mkdir -p $O; rm -f $O/shift_ko.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fleet.py -q -m gpu -k "shift or warm_start or fleet" -p no:cacheprovider -x > $O/pytest_shift.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_shift.txt
[ $rc -ne 0 ] && exit 1
for L in libhmpc.so libhmpc_ko2.so libhmpc_ko4.so libhmpc_ko6.so libhmpc_ko7.so; do
for W in 16 4; do
  HMPC_LIB=$L HMPC_JIT_SELFCHECK=0 HMPC_SHIFT_ROW_WAVES=$W timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | grep -v " 4096 " | sed "s/^/waves $W: /" | tee -a $O/shift_ko.txt
done; done
