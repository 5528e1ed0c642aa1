import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
g = load_fixture('qp_golden')
fix, x0 = g['n20dive_fix'], g['n20dive_x0']
co = make_controller(backend='oracle'); ch = make_controller(backend='hip')
ro = co.qp.solve_batch(x0, fix); rh = ch.qp.solve_batch(x0, fix)
nx = 4; T = 20
xa, xb = rh['primal'][:, :84], ro['primal'][:, :84]
dev = np.max(np.abs(xa - xb), axis=1)
print('iters oracle', ro['iters']); print('iters hip   ', rh['iters'])
print('dev', np.array2string(dev, precision=2))
k = int(os.environ.get("DBG_K", np.argmax(dev))); print('worst node', k, 'obj', ro['obj'][k], rh['obj'][k], 'golden', g['n20dive_obj'][k])
print('dev vs golden hip', np.max(np.abs(xa[k] - g['n20dive_x'][k])), 'oracle', np.max(np.abs(xb[k] - g['n20dive_x'][k])))
if os.environ.get('HMPC_TRACE'):
    ch.qp.solve_batch(x0, fix[k:k+1])
    os.environ['ORACLE_QP_TRACE'] = '1'
    co.qp.solve_batch(x0, fix[k:k+1])
