#!/bin/bash
# Round 3, GPU session 2: -m gpu suite on the tree with the parent -> child hand-down, the new default bench line.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_2.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_2.log
grep -E "passed|failed|margins|^sd |FAILED|Error" $O/pytest_gpu_2.log | tail -20
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_2.json 2> $O/bench_2.err; echo "bench rc $?"; tail -c 300 $O/bench_2.err
