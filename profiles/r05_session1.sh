#!/bin/bash
# Round 5, GPU session 1 (diagnostics): nz = 16 register kernels after the LDS-carve fix; the undecided node of the bench's
# generic_vs_specialised workload; which compiled kernels of the known-bad shapes are wrong with the final flags; pass bisection.
set -o pipefail
mkdir -p gpurun_out/r05_s1
export HMPC_JIT_VERBOSE=1
( DBG_SHAPES="9,3,4,6,23;8,4,4,6,23;10,2,4,6,23;9,3,4,10,23" DBG_ONLY="sized;per shape" timeout -k 10 600 python tests/gpu_dev_selfcheck_case2.py ) > gpurun_out/r05_s1/nz16.txt 2>&1
echo "nz16 done: $?"; tail -12 gpurun_out/r05_s1/nz16.txt
( timeout -k 10 600 python tests/gpu_dev_gvs.py ) > gpurun_out/r05_s1/gvs.txt 2>&1
echo "gvs done: $?"; grep -v "^hip ph\|^hip stamps" gpurun_out/r05_s1/gvs.txt | tail -30
( DBG_SHAPES="3,3,6,12,38;8,5,2,12,55;4,4,7,10,61" DBG_ONLY="sized" timeout -k 10 600 python tests/gpu_dev_selfcheck_case2.py ) > gpurun_out/r05_s1/bad_variants.txt 2>&1
echo "variants done: $?"; tail -5 gpurun_out/r05_s1/bad_variants.txt
( DBG_SHAPE=3,3,6,12,38 DBG_WAVES=1 timeout -k 10 1500 python tests/gpu_dev_bisect.py ) > gpurun_out/r05_s1/bisect_3_3_6.txt 2>&1
echo "bisect done: $?"; tail -25 gpurun_out/r05_s1/bisect_3_3_6.txt
