#!/bin/bash
# Round 5, GPU session 2 (diagnostics): the hard nodes of the generic_vs_specialised workload without the nets; the compiled
# kernels that come out wrong, with stack-slot sharing off (the pass the bisection of session 1 ended at); bisection of the second one.
set -o pipefail
mkdir -p gpurun_out/r05_s2
export HMPC_JIT_VERBOSE=1
( timeout -k 10 600 python tests/gpu_dev_gvs.py ) > gpurun_out/r05_s2/gvs.txt 2>&1
echo "gvs done: $?"; grep -v "amdgpu.ids" gpurun_out/r05_s2/gvs.txt | tail -70
( HMPC_JIT_FLAGS="-mllvm -no-stack-slot-sharing" DBG_NO_TRACE=1 timeout -k 10 600 python tests/gpu_dev_gvs.py ) > gpurun_out/r05_s2/gvs_noshare.txt 2>&1
echo "gvs noshare done: $?"; grep -v "amdgpu.ids" gpurun_out/r05_s2/gvs_noshare.txt | tail -12
( DBG_SHAPES="3,3,6,12,38;8,5,2,12,55" DBG_ONLY="sized, no stack slot sharing;sized, stack slot colouring off" timeout -k 10 600 python tests/gpu_dev_selfcheck_case2.py ) > gpurun_out/r05_s2/bad_variants_noshare.txt 2>&1
echo "variants done: $?"; grep -v "amdgpu.ids" gpurun_out/r05_s2/bad_variants_noshare.txt | tail -6
( DBG_SHAPE=8,5,2,12,55 DBG_WAVES=4 timeout -k 10 1200 python tests/gpu_dev_bisect.py ) > gpurun_out/r05_s2/bisect_8_5_2.txt 2>&1
echo "bisect done: $?"; tail -22 gpurun_out/r05_s2/bisect_8_5_2.txt
