"""Diagnostic (GPU box): the node of the published sd = .003 replay on which the kernel did not converge (infeasible by about
the accuracy of the arithmetic): kernel trace (HMPC_TRACE) and statuses for 1 / 2 / 4 waves, beside the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller
d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hard_node_sd003.npz'))
x0, fix = d['x0'][0], d['fix']
orc = make_controller('cart_pole_with_walls', backend='oracle', threads=1)
b = orc.qp.solve_batch(x0, fix)
print('oracle: status %d iters %d weak %d' % (b['status'][0], b['iters'][0], b['weak'][0]))
os.environ['HMPC_TRACE'] = '1'
hip = make_controller('cart_pole_with_walls', backend='hip')
for waves in ('4', '2', '1'):
    os.environ['HMPC_WAVES'] = waves
    sys.stderr.write('WAVES %s\n' % waves); sys.stderr.flush()
    a = hip.qp.solve_batch(x0, np.repeat(fix, 1, axis=0))
    sys.stderr.flush()
    print('kernel %s waves: status %d iters %d weak %d' % (waves, a['status'][0], a['iters'][0], a['weak'][0]), flush=True)
