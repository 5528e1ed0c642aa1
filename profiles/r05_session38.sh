#!/bin/bash
# Round 5, session 38: the fleet of 1024 loops with the waves per node forced (its launches hold ~1900 nodes of which most are cheap hand-downs)
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s38; mkdir -p $O; rm -f $O/fleet_waves.txt
for W in default 2 4 default 2; do
  if [ $W = default ]; then unset HMPC_WAVES; else export HMPC_WAVES=$W; fi
  timeout -k 10 300 python tests/gpu_dev_fleet_steps.py 1024 2>&1 | grep -o "host phases.*\|warm steps/s over.*" | tail -2 | tr '\n' ' ' | sed "s/^/waves $W: /" | tee -a $O/fleet_waves.txt; echo | tee -a $O/fleet_waves.txt
done
