"""Diagnostic (GPU box): kernels compiled per shape with two waves per SIMD (HMPC_JIT_FLAGS) against the default (one):
random MLDs small enough that LDS holds eight nodes per CU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import torch
from helpers import random_mld, _NoBackend, random_prefix_frontier
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _device_rate
dev = torch.device('cuda', 0)
for (nx, nuc, nub, T, seed) in ((6, 2, 3, 12, 3), (8, 3, 4, 10, 2)):
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
    c = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    f = random_prefix_frontier(T, nub, 4096, p_one=0.3)
    f[0, :] = -1
    for flags in ('', '-DHMPC_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))'):
        if flags:
            os.environ['HMPC_JIT_FLAGS'] = flags
        else:
            os.environ.pop('HMPC_JIT_FLAGS', None)
        t0 = time.time()
        qp = HipBatchedQP(c.problem_data())
        r, _ = _device_rate(qp, x0, f, dev)
        print('shape', (nx, nuc, nub, T), 'flags', repr(flags), 'create %.1f s' % (time.time() - t0), 'kinds', qp.kernel_info(), 'grid/lds', qp.launch_info(),
              '%.0f QP/s, %.3f ms' % (r['qp_per_s'], r['kernel_ms_avg']), 'optimal', r['optimal'], 'infeasible', r['infeasible'], flush=True)
