"""Statuses / iterations of a library variant against the oracle (diagnostic; HMPC_LIB selects the variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import warm_start_hmpc_amd.qp_backend as qb
if os.environ.get('HMPC_LIB'):
    qb.LIBRARY_PATH = qb.LIBRARY_PATH.replace('libhmpc.so', os.environ['HMPC_LIB'])
from helpers import make_controller, random_prefix_frontier
hip = make_controller('cart_pole_with_walls', T=20, backend='hip')
orc = make_controller('cart_pole_with_walls', T=20, backend='oracle', threads=8)
fix = random_prefix_frontier(20, 4, 40, p_one=0.1, seed0=9000)
fix[0, :] = -1
x0 = np.array([0., 0., 1., 0.])
a, b = hip.qp.solve_batch(x0, fix), orc.qp.solve_batch(x0, fix)
bad = np.flatnonzero(a['status'] != b['status'])
print('mismatching nodes', bad, 'hip status', a['status'][bad], 'oracle', b['status'][bad], 'hip iters', a['iters'][bad], 'oracle iters', b['iters'][bad])
print('iters hip   ', a['iters'])
print('iters oracle', b['iters'])
print('depth', (fix >= 0).sum(axis=1))
