#!/bin/bash
# Round 5, GPU session 7: validation of the ILP-scheduled binaries (tests/gpu_validate_ilp.py -> gpurun_out/VALIDATED), and the one
# default-schedule kernel the sweep found wrong, with the shipped kernels' broadcast rule.
set -o pipefail
mkdir -p gpurun_out/r05_s7
( timeout -k 10 300 python tests/gpu_dev_t40.py ) > gpurun_out/r05_s7/t40_b.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_s7/t40_b.txt | tail -9
( timeout -k 10 1100 python tests/gpu_validate_ilp.py ) > gpurun_out/r05_s7/validate.txt 2>&1
echo "validation: $?"; tail -40 gpurun_out/r05_s7/validate.txt
