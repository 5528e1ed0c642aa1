"""Diagnostic (GPU box): where a fleet step spends its host wall time (hmpc_fleet_timing).
    python tests/gpu_fleet_phases.py [loops] [steps]"""
import sys

import numpy as np
from conftest import ROOT  # noqa: F401
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import FleetMPC

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')
errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
X0 = np.array([0., 0., 1., 0.])
for hand in (True, False):
    fl = FleetMPC(ctrl, K, handdown=hand)
    fl.closed_loop(X0, 2, errs[:, :2], frontier_width=8)
    cold = fl.closed_loop(X0, 1, errs[:, :1], frontier_width=8)
    s0 = fl.stats()
    st = fl.closed_loop(X0, steps + 1, errs, frontier_width=8)
    s1 = fl.stats()
    # warm steps only: subtract one cold step measured just before
    c0 = fl.stats()
    d = {k: s1['seconds'][k] - s0['seconds'][k] for k in s1['seconds']}
    wall = st['wall']
    print('%d loops, hand-down %s: %d steps in %.3f s (incl. one cold step of %.3f s); rounds %d, nodes launched %d'
          % (K, hand, steps + 1, wall, cold['wall'], s1['rounds'] - s0['rounds'], s1['launched'] - s0['launched']))
    print('   host wall by phase: ' + ', '.join('%s %.3f s' % kv for kv in d.items()) + ', python %.3f s' % (wall - sum(d.values())))
