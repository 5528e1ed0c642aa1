#!/bin/bash
# Round 4, GPU session 2: A/B of the accuracy changes in the cart-pole kernels (lam_0 from its row, stable denominator),
# then the -m gpu suite on the default build, then the escalation diagnostic on configs[4].
# libhmpc.so here: -DHMPC_LAM0_ROW=1 -DHMPC_STABLE_DEN=1 in the cart-pole kernels; libhmpc_b.so: -DHMPC_LAM0_ROW=0; libhmpc_c.so: both 0
# (make OUT=.. BUILD=.. CXXFLAGS=..).  Result, ms per 4096 nodes, two runs each: 8.654 / 8.658, 8.585 / 8.559, 8.570 / 8.546; identical
# iteration counts and records.  The final tree ships variant b (stable denominator only) for the cart-pole kernels.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
mkdir -p $O
for LIB in libhmpc.so libhmpc_b.so libhmpc_c.so libhmpc.so libhmpc_b.so libhmpc_c.so; do
  HMPC_LIBRARY_NAME=$LIB timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/ab2_$LIB.json 2>> $O/ab2.err; echo "$LIB rc $?"
  python -c "import json,sys; d=json.loads(open('$O/ab2_$LIB.json').read().strip().splitlines()[-1]); print('$LIB', d['value'], d['roofline']['kernel_ms_avg'], d['nodes'])"
done
timeout -k 10 300 python tests/gpu_dev_escalation.py 2>&1 | grep -v "^it \|^hip ph\|polish round\|cert \|sigma " | head -30
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu_2.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_2.log
tail -6 $O/pytest_gpu_2.log
