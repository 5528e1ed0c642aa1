"""Diagnostic (hand-run on the GPU box): MPC steps/s of a fleet of K closed loops over frontier width and speculation depth,
with the time of a step by phase (hmpc_fleet_timing).  python tests/gpu_dev_fleet_sweep.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import FleetMPC

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = 10
ctrl = make_controller('cart_pole_with_walls', backend='hip')
x_max = load_fixture('cart_pole_with_walls')['x_max']
errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
x0 = np.array([0., 0., 1., 0.])
ref = None
for fw, spec in ((8, 0), (16, 0), (32, 0), (64, 0), (4, 0), (8, 1), (16, 1), (8, 2), (8, -1)):
    fl = FleetMPC(ctrl, K, handdown=True)
    kw = dict(frontier_width=fw, speculation=spec, cold_speculation=0, cold_frontier_width=8)
    fl.closed_loop(x0, 2, errs[:, :2], **kw)
    cold = fl.closed_loop(x0, 1, errs[:, :1], **kw)
    s0 = fl.stats()
    st = fl.closed_loop(x0, steps + 1, errs, **kw)
    s1 = fl.stats()
    dt = st['wall'] - cold['wall']
    costs = st['costs']
    same = ref is None or np.allclose(costs, ref, rtol=1e-6, atol=1e-9, equal_nan=True)
    if ref is None:
        ref = costs
    sec = {k: s1['seconds'][k] - s0['seconds'][k] for k in s1['seconds']}
    print('K %d width %2d speculation %2d: %7.0f steps/s, warm step %.2f ms, launches/step %.1f, solves/step %.2f, nodes launched/step %.0f, costs as the first: %s, seconds %s'
          % (K, fw, spec, K * steps / dt, 1e3 * dt / steps, (s1['rounds'] - s0['rounds']) / (steps + 1.0), st['nodes_ws'][:, 1:].mean(),
             (s1['launched'] - s0['launched']) / (steps + 1.0), same, {k: round(v, 3) for k, v in sec.items()}), flush=True)
    del fl
