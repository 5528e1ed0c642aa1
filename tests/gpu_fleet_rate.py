"""Closed-loop MPC steps/s of the C++ fleet driver for several fleet sizes (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import FleetMPC
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')
X0 = np.array([0., 0., 1., 0.])
steps = 10
for width, spec in ((8, 0), (8, 4), (8, 2)):
    for K in ((1, 16, 64, 256, 1024) if spec == 0 else (1, 4, 16, 64)):
        errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
        fl = FleetMPC(ctrl, K)
        fl.closed_loop(X0, 2, errs[:, :2], frontier_width=width, speculation=spec)
        cold = fl.closed_loop(X0, 1, errs[:, :1], frontier_width=width, speculation=spec)
        s0 = fl.stats()
        st = fl.closed_loop(X0, steps + 1, errs, frontier_width=width, speculation=spec)
        s1 = fl.stats()
        dt = st['wall'] - cold['wall']
        print('width %d spec %d K %4d: %.0f steps/s, warm solves/step %.2f, launches/step %.2f, nodes/launch %.0f, cold step %.1f ms'
              % (width, spec, K, K * steps / dt, st['nodes_ws'][:, 1:].mean(), (s1['rounds'] - s0['rounds']) / (steps + 1.0),
                 (s1['launched'] - s0['launched']) / max(1, s1['rounds'] - s0['rounds']), 1e3 * cold['wall']), flush=True)
        del fl
