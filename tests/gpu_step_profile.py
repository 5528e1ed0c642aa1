"""Where a closed-loop MPC step spends its time (diagnostic, run by hand on the GPU box)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')

def run(n, depth):
    np.random.seed(0)
    x, ws = np.array([0., 0., 1., 0.]), None
    for k in range(n):
        e = 0.001 * np.random.randn(4) * x_max
        u, ws, info = ctrl.feedback(x, warm_start=ws, e0=e, frontier_width=8, speculation_depth=depth)
        x = info['x1']

run(3, 4)
pr = cProfile.Profile()
pr.enable()
run(21, 4)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
