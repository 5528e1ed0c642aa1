"""Diagnostic (hand-run on the GPU box): K closed loops as `parts` fleets on handles and host threads of their own
(tests/parallel_fleets.py): the kernel of one fleet overlaps the host bookkeeping of the others.  The time of a warm step
is taken from two runs of different length (fresh fleets each: allocations and the cold step cancel).
python tests/gpu_dev_fleet_parts.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from parallel_fleets import closed_loop_parallel

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_short, n_long = 6, 26
ctrl = make_controller('cart_pole_with_walls', backend='hip')
x_max = load_fixture('cart_pole_with_walls')['x_max']
errs = np.array([0.001 * np.random.RandomState(s).randn(n_long, 4) * x_max for s in range(K)])
x0 = np.array([0., 0., 1., 0.])
ref = None
kw = dict(frontier_width=8, speculation=0, cold_speculation=0, cold_frontier_width=8)
closed_loop_parallel(ctrl, x0, 2, errs[:, :2], parts=1, **kw)
for rep in range(2):
    for parts in (1, 2, 3, 4):
        a = closed_loop_parallel(ctrl, x0, n_short, errs[:, :n_short], parts=parts, **kw)
        b = closed_loop_parallel(ctrl, x0, n_long, errs, parts=parts, **kw)
        dt = (b['wall'] - a['wall']) / (n_long - n_short)
        same = ref is None or np.allclose(b['costs'], ref, rtol=1e-6, atol=1e-9, equal_nan=True)
        if ref is None:
            ref = b['costs']
        print('K %d in %d fleets (HMPC_WAVES %s): warm step %.2f ms = %7.0f steps/s (runs of %d and %d steps: %.1f and %.1f ms), costs as the first: %s'
              % (K, parts, os.environ.get('HMPC_WAVES', '-'), 1e3 * dt, K / dt, n_short, n_long, 1e3 * a['wall'], 1e3 * b['wall'], same), flush=True)
