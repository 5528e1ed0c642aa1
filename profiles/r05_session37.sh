#!/bin/bash
# Round 5, GPU session 37: every problem of tests/jit_problems.py under the DEFAULT recipe (what a problem nobody has validated gets), on the
# kernels with the end-game fraction: the run of tests/gpu_validate_ilp.py in report mode (binaries compiled on the box)
set -o pipefail
mkdir -p gpurun_out/r05_s37
( VAL_SCHED=default timeout -k 10 1150 python tests/gpu_validate_ilp.py ) > gpurun_out/r05_s37/validate_default.txt 2>&1
echo "default recipe: $?"; tail -26 gpurun_out/r05_s37/validate_default.txt | cut -c1-220
