"""Distribution of the GPU <-> oracle trajectory deviation over a large frontier (diagnostic, feeds DESIGN.md)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier
T = 20
hip = make_controller('cart_pole_with_walls', T=T, backend='hip')
orc = make_controller('cart_pole_with_walls', T=T, backend='oracle', threads=16)
x0 = np.array([0., 0., 1., 0.])
for p_one in (0.1, 0.02):
    fix = random_prefix_frontier(T, 4, 4096, p_one=p_one, seed0=50000)
    a, b = hip.qp.solve_batch(x0, fix), orc.qp.solve_batch(x0, fix)
    assert np.array_equal(a['status'], b['status'])
    fin = a['status'] == 0
    xa, xb = a['primal'][fin][:, :(T + 1) * 4], b['primal'][fin][:, :(T + 1) * 4]
    scale = np.maximum(1e-2, np.max(np.abs(xb), axis=1))
    dev = np.max(np.abs(xa - xb), axis=1) / scale
    frac = (fix[fin] >= 0).mean(axis=1)
    obj = np.abs(a['obj'][fin] - b['obj'][fin]) / (1 + np.abs(b['obj'][fin]))
    print('p_one %.2f: %d feasible of 4096; iterations equal on %.1f%%' % (p_one, fin.sum(), 100 * np.mean(a['iters'] == b['iters'])))
    print('  trajectory deviation: median %.1e, 99%% %.1e, max %.1e; > 1e-5 on %d nodes (all with >= %.0f%% binaries fixed)'
          % (np.median(dev), np.percentile(dev, 99), dev.max(), (dev > 1e-5).sum(), 100 * (frac[dev > 1e-5].min() if (dev > 1e-5).any() else 1)))
    print('  objective deviation: max %.1e' % obj.max())
    inf = a['status'] == 1
    print('  Farkas rays: max |difference| %.1e over %d infeasible nodes' % (np.max(np.abs(a['dual'][inf] - b['dual'][inf])), inf.sum()))
