#!/bin/bash
# Round 5, session 31: what a node costs by how its solve goes (cold / hand-down verified / hand-down dropped); mixes; hand-out order
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=warm-start-hybrid-mpc_amd:.:tests
O=gpurun_out/r05_s31; mkdir -p $O
timeout -k 10 500 python tests/gpu_dev_handdown_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/handdown_cost.txt
HMPC_NO_ORDER=1 timeout -k 10 500 python tests/gpu_dev_handdown_cost.py 2>&1 | grep "array order" | tee -a $O/handdown_cost.txt
