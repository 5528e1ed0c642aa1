#!/bin/bash
# Round 4, GPU session 3: the -m gpu suite (register kernels compiled for arbitrary shapes, fleet pools, sanitizer-split
# fleet driver) and the full default bench line.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_4.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_4.log
tail -8 $O/pytest_gpu_4.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_3.json 2> $O/bench_3.err; echo "bench rc $?"; tail -3 $O/bench_3.err
