"""Diagnostic: phase stamps (HMPC_TRACE=1, a stamps build of the kernel) of one infeasible node of the N = 40 frontier, two
waves per node.  Run once per build:
    HMPC_LIBRARY_NAME=libhmpc_stamps.so HMPC_JIT_SIZED=0 HMPC_TRACE=1 python tests/gpu_dev_n40_stamps.py
    HMPC_JIT_FLAGS=-DHMPC_STAMPS HMPC_TRACE=1 python tests/gpu_dev_n40_stamps.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier
T = 40
fix = random_prefix_frontier(T, 4, 2048, p_one=0.5)
os.environ['HMPC_WAVES'] = '2'
hip = make_controller('cart_pole_with_walls', T=T, backend='hip')
k = int(os.environ.get('DBG_NODE', '880'))
r = hip.qp.solve_batch(np.array([0., 0., 1., 0.]), fix[k:k + 1])
print('kinds', hip.qp.kernel_info(), 'status', r['status'], 'iters', r['iters'], 'launch', hip.qp.launch_info())
