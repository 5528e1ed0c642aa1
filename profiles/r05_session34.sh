#!/bin/bash
# Round 5, GPU session 34: the kernels with the end-game fraction to the boundary: validation of the ILP-scheduled binaries
# (tests/gpu_validate_ilp.py -> gpurun_out/VALIDATED)
set -o pipefail
mkdir -p gpurun_out/r05_s34
( timeout -k 10 1150 python tests/gpu_validate_ilp.py ) > gpurun_out/r05_s34/validate.txt 2>&1
echo "validation: $?"; tail -32 gpurun_out/r05_s34/validate.txt | cut -c1-260
