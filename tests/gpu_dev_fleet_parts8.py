"""Diagnostic (hand-run on the GPU box): tests/parallel_fleets.py: closed_loop_parallel with 8 fleets on 8 host threads, over and over -- the configuration of
the one unexplained core dump of round 5 (profiles/r05_fleet_trace.txt).  python -X faulthandler tests/gpu_dev_fleet_parts8.py [repeats]"""
import os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from parallel_fleets import closed_loop_parallel
K, steps = 1024, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctrl = make_controller('cart_pole_with_walls', backend='hip')
x_max = load_fixture('cart_pole_with_walls')['x_max']
errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
x0 = np.array([0., 0., 1., 0.])
kw = dict(frontier_width=8, speculation=0, cold_speculation=0, cold_frontier_width=8)
ref = None
keep = {} if os.environ.get('PARTS_KEEP') else None          # (PARTS_KEEP=1: the eight handles and fleets are created once)
for rep in range(reps):
    st = closed_loop_parallel(ctrl, x0, steps + 1, errs, parts=8, keep=keep, **kw)
    same = ref is None or np.allclose(st['costs'], ref, rtol=1e-6, atol=1e-9, equal_nan=True)
    if ref is None:
        ref = st['costs']
    print('rep %d: %.0f steps/s incl. cold, costs as the first: %s' % (rep, st['steps_per_sec'], same), flush=True)
