"""GPU <-> oracle deviation on the nodes of a real branch-and-bound tree (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller
X0 = np.array([0., 0., 1., 0.])
orc = make_controller('cart_pole_with_walls', backend='oracle', threads=16)
seen = []
inner = orc.solve_frontier
def recording(identifiers, x0):
    seen.extend(orc._fix_vector(i) for i in identifiers)
    return inner(identifiers, x0)
orc.solve_frontier = recording
sol, leaves, solves, _ = orc.feedforward(X0, printing_period=None)
orc.solve_frontier = inner
fix = np.array(seen, dtype=np.int8)
hip = make_controller('cart_pole_with_walls', backend='hip')
for waves in ('1', '2', '4'):
    os.environ['HMPC_WAVES'] = waves
    a, b = hip.qp.solve_batch(X0, fix), orc.qp.solve_batch(X0, fix)
    assert np.array_equal(a['status'], b['status'])
    fin = a['status'] == 0
    xa, xb = a['primal'][fin][:, :84], b['primal'][fin][:, :84]
    scale = np.maximum(1e-2, np.max(np.abs(xb), axis=1))
    dev = np.max(np.abs(xa - xb), axis=1) / scale
    frac = (fix[fin] >= 0).mean(axis=1)
    order = np.argsort(-dev)[:6]
    print('waves', waves, 'feasible', fin.sum(), 'iters equal %.0f%%' % (100 * np.mean(a['iters'] == b['iters'])),
          'dev median %.1e max %.1e' % (np.median(dev), dev.max()), 'count > 1e-5:', int((dev > 1e-5).sum()))
    print('   worst:', [(round(float(frac[i]), 2), '%.1e' % dev[i], int(a['iters'][fin][i]), int(b['iters'][fin][i])) for i in order])
    print('   objective dev max %.1e' % np.max(np.abs(a['obj'][fin] - b['obj'][fin]) / (1 + np.abs(b['obj'][fin]))))
