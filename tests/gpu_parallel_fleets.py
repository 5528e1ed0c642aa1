"""Diagnostic (GPU box): several fleets on several handles and host threads (fleet.closed_loop_parallel), with progress.
    python tests/gpu_parallel_fleets.py [loops] [steps]"""
import sys
import time

import numpy as np
from conftest import ROOT  # noqa: F401
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import FleetMPC, closed_loop_parallel

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')
errs = np.array([0.001 * np.random.RandomState(s).randn(steps, 4) * x_max for s in range(K)])
X0 = np.array([0., 0., 1., 0.])
t = time.perf_counter()
one = FleetMPC(ctrl, K).closed_loop(X0, steps, errs, frontier_width=8)
print('one fleet: %.2f s, %.0f steps/s' % (time.perf_counter() - t, one['steps_per_sec']), flush=True)
for parts in (2, 4):
    t = time.perf_counter()
    st = closed_loop_parallel(ctrl, X0, steps, errs, parts=parts, frontier_width=8)
    print('%d fleets: %.2f s, %.0f steps/s' % (parts, time.perf_counter() - t, st['steps_per_sec']), flush=True)
    assert np.allclose(st['costs'], one['costs'], rtol=1e-9, atol=1e-12) and np.array_equal(st['len_ws'], one['len_ws'])
print('ok')
