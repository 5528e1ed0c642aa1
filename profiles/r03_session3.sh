#!/bin/bash
# Round 3, GPU session 3: -m gpu suite (cold kernels without the hand-down code, terminal-set path reduced over lanes), bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_3.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_3.log
grep -E "passed|failed|margins|^sd |FAILED|Error" $O/pytest_gpu_3.log | tail -20
(cd tests && timeout -k 10 200 python gpu_nu_diff.py) > $O/nu_diff.txt 2>&1; cat $O/nu_diff.txt | grep -v amdgpu.ids
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_3.json 2> $O/bench_3.err; echo "bench rc $?"; tail -c 300 $O/bench_3.err
