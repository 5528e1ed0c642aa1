"""Closed-loop MPC steps/s as a function of frontier_width (diagnostic, run by hand on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.batched import BatchedMPC
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')
bm = BatchedMPC(ctrl)
for sims in (1, 64):
    for fw in (4, 8, 16, 32, 77):
        seeds = tuple(range(sims))
        warm = bm.closed_loop(np.array([0., 0., 1., 0.]), 1, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=fw)
        t0 = time.perf_counter()
        st = bm.closed_loop(np.array([0., 0., 1., 0.]), 11, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=fw)
        dt = time.perf_counter() - t0 - warm['wall']
        ws = np.array([v[1:] for v in st['nodes_ws']])
        print('sims %d width %d: %.1f steps/s, warm solves/step %.1f' % (sims, fw, sims * 10 / dt, ws.mean()), flush=True)
