"""One small batch through the C ABI, checked against the oracle (first thing to run after a kernel change)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import warm_start_hmpc_amd.qp_backend as qb
if os.environ.get('HMPC_LIB'):
    qb.LIBRARY_PATH = qb.LIBRARY_PATH.replace('libhmpc.so', os.environ['HMPC_LIB'])
from helpers import make_controller, random_prefix_frontier
T = int(os.environ.get('DBG_T', 20))
name = os.environ.get('DBG_FIXTURE', 'cart_pole_with_walls')
hip = make_controller(name, T=T, backend='hip')
orc = make_controller(name, T=T, backend='oracle', threads=8)
fix = random_prefix_frontier(T, hip.mld.nub, int(os.environ.get('DBG_B', 40)), p_one=0.1, seed0=9000)
fix[0, :] = -1
x0 = np.array([0., 0., 1., 0.])
a, b = hip.qp.solve_batch(x0, fix), orc.qp.solve_batch(x0, fix)
print('waves', os.environ.get('HMPC_WAVES'), 'status equal', np.array_equal(a['status'], b['status']),
      'max obj diff', np.nanmax(np.abs(np.where(np.isfinite(b['obj']), a['obj'] - b['obj'], 0.))), hip.qp.launch_info(), flush=True)
