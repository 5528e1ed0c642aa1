"""Development check of the sized kernels (diagnostic, not a test): the run-time-sized kernel compiled with a problem's sizes
(csrc/hmpc_jit.h) against the shipped build, with and without the row state in registers, on a random MLD beyond the
static row map (nx + nu = 18) -- rate of a 2048-node launch per waves per node.

    python tests/gpu_dev_sized.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
import torch
from helpers import random_prefix_frontier
from jit_problems import problem, SIZED
from warm_start_hmpc_amd.qp_backend import HipBatchedQP

spec = tuple(int(v) for v in os.environ.get('DBG_SPEC', ','.join(str(v) for v in SIZED[1])).split(','))
nx, nuc, nub, seed, T = spec
data, mld, objective, x0 = problem(*spec)
fix = random_prefix_frontier(T, nub, 2048, p_one=0.3)
fix[::2, :] = -1
fix[::2, :T * nub // 2] = fix[1::2, :T * nub // 2] * 0          # (half of the nodes: prefixes of zeros, mostly feasible)
dev = torch.device('cuda')
for label, env in (('sized, rows in registers', {}), ('sized, rows in the slab', {'HMPC_JIT_SIZED_ROWS': '0'}), ('shipped run-time-sized', {'HMPC_JIT_SIZED': '0'})):
    os.environ.update(env)
    t0 = time.perf_counter()
    qp = HipBatchedQP(data)
    tc = time.perf_counter() - t0
    for k in env:
        del os.environ[k]
    B = len(fix)
    out = dict(obj=torch.empty(B, dtype=torch.float64, device=dev), dual_obj=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev),
               primal=torch.empty(B, qp.n_primal, dtype=torch.float64, device=dev), dual=torch.empty(B, qp.n_dual, dtype=torch.float64, device=dev))
    xd, fd = torch.from_numpy(x0).to(dev), torch.from_numpy(fix).to(dev)
    line = '%-28s kinds %s create %.1f s;' % (label, qp.kernel_info(), tc)
    for waves in ('1', '2', '4'):
        os.environ['HMPC_WAVES'] = waves
        qp.solve_batch_device(xd, fd, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            qp.solve_batch_device(xd, fd, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        del os.environ['HMPC_WAVES']
        st = out['status'].cpu().numpy()
        line += ' w%s %.2f ms (%d opt %d inf)' % (waves, 1e3 * dt, (st == 0).sum(), (st == 1).sum())
    print(line, flush=True)
