#!/bin/bash
# Round 5, session 44: the first-use check with a pinned host mirror -- 8 fleets on 8 host threads, 4 x 60 calls in processes of their own
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s44; mkdir -p $O
for run in 1 2 3 4; do
HMPC_BACKTRACE=1 timeout -k 10 200 python tests/gpu_dev_fleet_parts8.py 60 > $O/parts8_check_$run.txt 2>&1; echo "check on (pinned mirror), run $run: rc $? ($(grep -c 'steps/s' $O/parts8_check_$run.txt) of 60)"; grep -A8 "fatal signal" $O/parts8_check_$run.txt | cut -c1-150
done
