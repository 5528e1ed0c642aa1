#!/bin/bash
# Round 5, session 22: hunting the one core dump of session 20 (8 fleets x HMPC_WAVES=1): the same sequence three times with each shift kernel
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s22; mkdir -p $O
for rep in 1 2 3; do for rows in 1 0; do
  HMPC_SHIFT_ROWS=$rows HMPC_WAVES=1 timeout -k 10 200 python -X faulthandler tests/gpu_dev_fleet_parts.py 1024 > $O/parts_rows${rows}_$rep.txt 2>&1; rc=$?
  echo "rows $rows rep $rep rc $rc: $(grep -c 'steps/s' $O/parts_rows${rows}_$rep.txt) lines"
  if [ $rc -ne 0 ]; then grep -v amdgpu.ids $O/parts_rows${rows}_$rep.txt | tail -40; fi
done; done
