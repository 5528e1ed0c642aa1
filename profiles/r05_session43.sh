#!/bin/bash
# Round 5, session 43: 8 fleets on 8 host threads, over and over in one process (the configuration of the core dump), native backtrace on;
# with the first-use check of the compiled kernels off, then on
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s43; mkdir -p $O
for run in 1 2 3; do
HMPC_JIT_SELFCHECK=0 HMPC_BACKTRACE=1 timeout -k 10 200 python tests/gpu_dev_fleet_parts8.py 60 > $O/parts8_nocheck_$run.txt 2>&1; echo "check off, run $run: rc $? ($(grep -c 'steps/s' $O/parts8_nocheck_$run.txt) of 60)"
HMPC_BACKTRACE=1 timeout -k 10 200 python tests/gpu_dev_fleet_parts8.py 60 > $O/parts8_check_$run.txt 2>&1; echo "check on,  run $run: rc $? ($(grep -c 'steps/s' $O/parts8_check_$run.txt) of 60)"; grep -A6 "fatal signal" $O/parts8_check_$run.txt | cut -c1-150
done
