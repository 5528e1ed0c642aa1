"""Diagnostic (hand-run on the GPU box): wall time of every step of a fleet of K closed loops (solve and shift apart), a fresh
fleet and the same fleet once more after reset.  python tests/gpu_dev_fleet_steps.py [K]"""
import os, sys
from time import perf_counter
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import FleetMPC

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = 12
ctrl = make_controller('cart_pole_with_walls', backend='hip')
x_max = load_fixture('cart_pole_with_walls')['x_max']
errs = np.array([0.001 * np.random.RandomState(s).randn(steps, 4) * x_max for s in range(K)])
x0 = np.array([0., 0., 1., 0.])
fl = FleetMPC(ctrl, K, handdown=True)
for run in range(3):
    fl.reset()
    xs = np.repeat(x0[None], K, axis=0)
    ts, tsh, nodes, marks = [], [], [], []
    s_prev = fl.stats()
    for t in range(steps):
        a = perf_counter()
        r = fl.solve(xs, 8)
        b = perf_counter()
        fl.shift(errs[:, t])
        c = perf_counter()
        xs = r['x1'] + errs[:, t]
        ts.append(1e3 * (b - a)); tsh.append(1e3 * (c - b)); nodes.append(int(r['solves'].sum()))
        if t in (0, steps - 1):
            marks.append(fl.stats())
    s = fl.stats()
    sec = {k: round(s['seconds'][k] - s_prev['seconds'][k], 3) for k in s['seconds']}
    print('run %d: solve ms per step %s' % (run, ' '.join('%.1f' % v for v in ts)))
    print('        shift ms per step %s' % ' '.join('%.1f' % v for v in tsh))
    print('        host phases of the %d warm steps, ms per step: %s; rounds per warm step %.1f' % (steps - 1, {k: round(1e3 * (marks[1]['seconds'][k] - marks[0]['seconds'][k]) / (steps - 1), 2) for k in marks[0]['seconds']},
                                                                                                      (marks[1]['rounds'] - marks[0]['rounds']) / (steps - 1.0)))
    print('        nodes launched per warm step %.0f, of them verified hand-downs %.0f' % ((marks[1]['launched'] - marks[0]['launched']) / (steps - 1.0), (marks[1]['handed'] - marks[0]['handed']) / (steps - 1.0)))
    print('        solves per step %s; rounds %d; phases (s) %s; warm steps/s over steps 1.. : %.0f'
          % (' '.join(str(v) for v in nodes), s['rounds'] - s_prev['rounds'], sec, K * (steps - 1) / (1e-3 * (sum(ts[1:]) + sum(tsh[1:])))), flush=True)
