"""Diagnostic (GPU box): where a fleet step spends its host wall time (hmpc_fleet_timing), cold step and warm steps apart.
    python tests/gpu_fleet_phases.py [loops] [steps]"""
import sys
from time import perf_counter

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


def show(tag, wall, a, b):
    d = {k: b['seconds'][k] - a['seconds'][k] for k in b['seconds']}
    print('   %-10s %.3f s wall; rounds %d, nodes %d (hand-down verified %d); host wall by phase: ' % (tag, wall, b['rounds'] - a['rounds'], b['launched'] - a['launched'], b['handed'] - a['handed'])
          + ', '.join('%s %.3f' % kv for kv in d.items()) + ', python %.3f' % (wall - sum(d.values())))


for hand in (True, False):
    fl = FleetMPC(ctrl, K, handdown=hand)
    fl.closed_loop(X0, 2, errs[:, :2], frontier_width=8)      # warm-up (allocations)
    fl.reset()
    xs = np.repeat(X0[None], K, axis=0)
    print('%d loops, hand-down %s' % (K, hand))
    s0, t0 = fl.stats(), perf_counter()
    r = fl.solve(xs, 8)
    fl.shift(errs[:, 0])
    xs = r['x1'] + errs[:, 0]
    s1, t1 = fl.stats(), perf_counter()
    show('cold step', t1 - t0, s0, s1)
    for t in range(1, steps + 1):
        r = fl.solve(xs, 8)
        fl.shift(errs[:, t])
        xs = r['x1'] + errs[:, t]
    s2, t2 = fl.stats(), perf_counter()
    show('%d warm' % steps, t2 - t1, s1, s2)
    print('   => %.0f warm steps/s' % (K * steps / (t2 - t1)))
