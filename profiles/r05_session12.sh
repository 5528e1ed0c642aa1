#!/bin/bash
# Round 5, GPU session 12: evidence on the final kernels -- the randomized kernel <-> oracle sweeps (N = 20, N = 40) and the shape sweeps
# (problems nobody has validated: the default recipe, nets on).
set -o pipefail
mkdir -p gpurun_out/r05_s12
( DBG_REPS=24 timeout -k 10 500 python tests/gpu_parity_sweep.py ) > gpurun_out/r05_s12/parity_sweep_n20.txt 2>&1; echo "sweep n20: $?"; tail -6 gpurun_out/r05_s12/parity_sweep_n20.txt
( DBG_REPS=8 DBG_T=40 DBG_SKIP_WIDE=1 timeout -k 10 400 python tests/gpu_parity_sweep.py ) > gpurun_out/r05_s12/parity_sweep_n40.txt 2>&1; echo "sweep n40: $?"; tail -4 gpurun_out/r05_s12/parity_sweep_n40.txt
( timeout -k 10 600 python tests/gpu_sized_shapes.py ) > gpurun_out/r05_s12/sized_shapes.txt 2>&1; echo "shapes: $?"; tail -3 gpurun_out/r05_s12/sized_shapes.txt | cut -c1-200
( DBG_EDGE=1 timeout -k 10 600 python tests/gpu_sized_shapes.py ) > gpurun_out/r05_s12/sized_shapes_edge.txt 2>&1; echo "edge shapes: $?"; tail -3 gpurun_out/r05_s12/sized_shapes_edge.txt | cut -c1-200
