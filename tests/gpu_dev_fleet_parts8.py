import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.fleet import closed_loop_parallel
K, steps = 1024, 10
ctrl = make_controller('cart_pole_with_walls', backend='hip')
x_max = load_fixture('cart_pole_with_walls')['x_max']
errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
x0 = np.array([0., 0., 1., 0.])
kw = dict(frontier_width=8, speculation=0, cold_speculation=0, cold_frontier_width=8)
for rep in range(3):
    for parts in (4, 8):
        st = closed_loop_parallel(ctrl, x0, steps + 1, errs, parts=parts, **kw)
        print('rep', rep, 'parts', parts, 'steps/s incl cold', st['steps_per_sec'], flush=True)
