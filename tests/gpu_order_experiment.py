"""Diagnostic (GPU box): how much of a launch of real-tree nodes is its tail -- the same frontier handed out in the
kernel's own order (shallow first), in index order, and longest-first by the oracle's iteration counts (a bound: the
kernel cannot know them in advance).
    python tests/gpu_order_experiment.py"""
import os
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
print('node-iterations %d: ideal makespan at 1024 nodes in flight and 165 us per iteration %.2f ms; longest node %d iterations = %.2f ms'
      % (o['iters'].sum(), o['iters'].sum() / 1024 * 0.165, o['iters'].max(), o['iters'].max() * 0.165))


def run(tag, perm, env=None):
    if env:
        os.environ[env] = '1'
    r, _ = bench._device_rate(hip.qp, x0[perm], fix[perm], dev)
    if env:
        del os.environ[env]
    print('%-46s %.3f ms' % (tag, r['kernel_ms_avg']), flush=True)


ident = np.arange(4096)
run('kernel order (shallow first)', ident)
run('index order (HMPC_NO_ORDER)', ident, 'HMPC_NO_ORDER')
lpt = np.argsort(-o['iters'], kind='stable')
run('longest first by oracle iterations', lpt, 'HMPC_NO_ORDER')
run('shortest first by oracle iterations', lpt[::-1], 'HMPC_NO_ORDER')
rng = np.random.RandomState(0)
run('random order', rng.permutation(4096), 'HMPC_NO_ORDER')
