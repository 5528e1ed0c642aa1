"""Diagnostic (hand-run on the GPU box): the tolerance from which the polish is attempted (hmpc_options.polish_tol, a run-time option)
against the rate of the headline frontier, of the trees of distinct states and of the N = 40 frontier.  python tests/gpu_dev_ptol.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import torch
from helpers import make_controller, load_fixture
import bench

dev = torch.device('cuda', 0)
x_max = load_fixture('cart_pole_with_walls')['x_max']
base = {}
for T, B in ((20, 4096), (40, 2048)):
    ref = make_controller('cart_pole_with_walls', T=T, backend='hip')
    x0n, fixn, _ = bench.real_tree_frontier(ref, B, 0, x_max, spread=0.)
    x0t, fixt, _ = bench.real_tree_frontier(ref, B, 0, x_max) if T == 20 else (None, None, None)
    for ptol in (1e-4, 3e-4, 1e-3, 3e-3):
        ctrl = make_controller('cart_pole_with_walls', T=T, backend='hip', polish_tol=ptol)
        r, st = bench._device_rate(ctrl.qp, x0n, fixn, dev, reps=8)
        line = 'N %d polish_tol %.0e: replay frontier %.0f k QP/s (%.3f ms, iterations %.2f, optimal %.2f, polished %d of %d, undecided %d)' % (
            T, ptol, r['qp_per_s'] / 1e3, r['kernel_ms_avg'], r['ipm_iters_mean'], r['ipm_iters_mean_optimal'], r['polished'], r['optimal'], r['not_converged'])
        if x0t is not None:
            r2, st2 = bench._device_rate(ctrl.qp, x0t, fixt, dev, reps=8)
            line += '; distinct states %.0f k QP/s (iterations %.2f, polished %d of %d, undecided %d)' % (r2['qp_per_s'] / 1e3, r2['ipm_iters_mean'], r2['polished'], r2['optimal'], r2['not_converged'])
            key = (T, 'd')
            if key not in base: base[key] = st2
            line += ', statuses as at 1e-4: %s' % np.array_equal(st2, base[key])
        if (T, 'r') not in base: base[(T, 'r')] = st
        print(line + '; replay statuses as at 1e-4: %s' % np.array_equal(st, base[(T, 'r')]), flush=True)
