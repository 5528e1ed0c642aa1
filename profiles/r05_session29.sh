#!/bin/bash
# Round 5, session 29: does the alignment of the rows bound the shift kernel?  Rows of 4480 B (T = 10) and 10 240 B (T = 26) are whole
# 128-byte lines, the headline's 8080 B (T = 20) and 9 630 B (T = 24... 45 T + 110 entries) are not.
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests HMPC_JIT=0
O=gpurun_out/r05_s29; mkdir -p $O
for T in 20 26 25 10 11; do
  SHIFT_T=$T timeout -k 10 300 python tests/gpu_shift_time.py 2>/dev/null | grep -v " 4096 " | tee -a $O/shift_alignment.txt
done
