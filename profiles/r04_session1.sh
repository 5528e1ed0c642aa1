#!/bin/bash
# Round 4, GPU session 1 (through gpurun from the repo root: bash profiles/r04_session1.sh): the -m gpu suite with the
# tolerance escalation (configs[4] at 4096 nodes, one RTOL), A/B of the escalation compiled into the register kernels,
# bench.py starting its own ranks (rehearsal on one GPU).
# libhmpc_esc.so: the same tree built with -DHMPC_ESC_ALL (make OUT=../libhmpc_esc.so BUILD=build_esc CXXFLAGS="... -DHMPC_ESC_ALL"):
# the escalation compiled into the cart-pole kernels as well.  Result: 8.602 / 8.603 ms against 8.639 / 8.639 (+0.4 %), same records.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu_1.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu_1.log
tail -6 $O/pytest_gpu_1.log
for LIB in libhmpc.so libhmpc_esc.so libhmpc.so libhmpc_esc.so; do
  HMPC_LIBRARY_NAME=$LIB timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/ab_$LIB.json 2>> $O/ab.err; echo "$LIB rc $?"
  python -c "import json,sys; d=json.loads(open('$O/ab_$LIB.json').read().strip().splitlines()[-1]); print('$LIB', d['value'], d['roofline']['kernel_ms_avg'], d['nodes'])"
done
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $O/rehearse.json 2> $O/rehearse.err; echo "rehearse rc $?"; tail -c 1500 $O/rehearse.json
