#!/bin/bash
# Round 5, session 45: 8 fleets on 8 host threads with the handles and fleets created ONCE (no creation / destruction while threads work), 4 x 60 calls
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s45; mkdir -p $O
for run in 1 2 3 4; do
PARTS_KEEP=1 HMPC_BACKTRACE=1 timeout -k 10 200 python tests/gpu_dev_fleet_parts8.py 60 > $O/parts8_keep_$run.txt 2>&1; echo "handles kept, run $run: rc $? ($(grep -c 'steps/s' $O/parts8_keep_$run.txt) of 60)"; grep -A12 "fatal signal" $O/parts8_keep_$run.txt | cut -c1-150
done
