"""Diagnostic (GPU box): where the time of the real-tree frontier goes -- kernel time of subsets of its nodes.
    python tests/gpu_frontier_cost.py"""
import sys

import numpy as np
import torch
from conftest import ROOT  # noqa: F401
sys.path.insert(0, ROOT)
import bench
from helpers import make_controller, load_fixture

dev = torch.device('cuda')
hip = make_controller('cart_pole_with_walls', backend='hip')
orc = make_controller('cart_pole_with_walls', backend='oracle', threads=16)
x_max = load_fixture('cart_pole_with_walls')['x_max']
x0, fix, par = bench.real_tree_frontier(hip, 4096, 0, x_max)
o = orc.qp.solve_batch(x0, fix)
second, opt = o['second'] > 0, o['status'] == 0
depth = (fix >= 0).sum(axis=1)


def rate(name, m, parent=None):
    m = np.flatnonzero(m)
    if m.size < 64:
        print('%-40s only %d nodes' % (name, m.size)); return
    reps = int(np.ceil(4096 / m.size))
    idx = np.tile(m, reps)[:4096]
    r, _ = bench._device_rate(hip.qp, x0[idx], fix[idx], dev)
    print('%-40s %5d distinct nodes, tiled to 4096: %7.3f ms, iters %.2f (oracle %.2f)' % (name, m.size, r['kernel_ms_avg'], r['ipm_iters_mean'], o['iters'][idx].mean()), flush=True)


rate('all', np.ones(4096, bool))
rate('no second solve', ~second)
rate('second solve only', second)
rate('optimal, no second', opt & ~second)
rate('infeasible', ~opt)
rate('nominal state only', np.all(x0 == x0[0], axis=1))
rate('perturbed states, no second', ~np.all(x0 == x0[0], axis=1) & ~second)
for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 81)):
    rate('depth %d..%d, no second' % (lo, hi - 1), (depth >= lo) & (depth < hi) & ~second)
its = o['iters']
rate('iters <= 12', its <= 12)
rate('iters 13..16', (its > 12) & (its <= 16))
rate('iters > 16, no second', (its > 16) & ~second)
