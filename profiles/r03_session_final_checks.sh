#!/bin/bash
# Round 3, evidence on the final kernel: the HMPC_CHECK build over every instantiation, the randomized parity sweep,
# the multi-rank rehearsal of bench.py on one GPU (gloo, all ranks on cuda:0 -- not a measurement).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
(cd tests && timeout -k 10 400 python gpu_check_build.py) 2>&1 | grep -v amdgpu.ids | tee $O/check_build.txt | tail -4
(cd tests && DBG_REPS=40 timeout -k 10 500 python gpu_parity_sweep.py) 2>&1 | grep -v amdgpu.ids | tee $O/parity_sweep.txt | tail -12
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-on-one-gpu 2> $O/rehearse2.err | tee $O/rehearse2.json | cut -c1-400
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-on-one-gpu --frontier-total 1024 2> $O/rehearse2_strong.err | tee $O/rehearse2_strong.json | cut -c1-400
