import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
import warm_start_hmpc_amd.qp_backend as qb
qb.LIBRARY_PATH = qb.LIBRARY_PATH.replace('libhmpc.so', 'libhmpc_stamps.so')
from helpers import make_controller, random_prefix_frontier
ch = make_controller(T=20, backend='hip')
fix = random_prefix_frontier(20, 4, int(os.environ.get('DBG_B', 1)), p_one=0.1)
fix[0, :] = -1
r = ch.qp.solve_batch(np.array([0., 0., 1., 0.]), fix)
print('iters', r['iters'][:4], 'time', r['time'])
