#!/bin/bash
# Round 5, GPU session 8: validation of the ILP-scheduled binaries of the final tree (tests/gpu_validate_ilp.py -> gpurun_out/VALIDATED).
set -o pipefail
mkdir -p gpurun_out/r05_s8
( timeout -k 10 1150 python tests/gpu_validate_ilp.py ) > gpurun_out/r05_s8/validate.txt 2>&1
echo "validation: $?"; tail -32 gpurun_out/r05_s8/validate.txt | cut -c1-260
