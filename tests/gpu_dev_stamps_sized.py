"""Diagnostic (GPU box): cycle stamps per phase of the headline kernel AS COMPILED FOR THE PROBLEM (sizes as constants; the stamps
go in through the flags of the run-time compilation, a cache entry of its own), node 0 of a full launch at one wave per node.

    DBG_B=4096 python tests/gpu_dev_stamps_sized.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
os.environ['HMPC_TRACE'] = '1'
os.environ['HMPC_JIT_FLAGS'] = (os.environ.get('HMPC_JIT_FLAGS', '') + ' -DHMPC_STAMPS').strip()
os.environ['HMPC_WAVES'] = os.environ.get('DBG_WAVES', '1')
from helpers import make_controller, load_fixture
from bench import real_tree_frontier
T = int(os.environ.get('DBG_T', 20))
ch = make_controller(T=T, backend='hip')
B = int(os.environ.get('DBG_B', 4096))
x0, fix, _ = real_tree_frontier(ch, B, 0, load_fixture('cart_pole_with_walls')['x_max'], spread=0.)
print('kinds', ch.qp.kernel_info(), flush=True)
sys.stderr.flush()
r = ch.qp.solve_batch(x0, fix)
sys.stderr.flush()
print('node 0: status', r['status'][0], 'iters', r['iters'][0] & 0xffff, 'polished', r['polished'][0], 'batch time', r['time'])
