"""Where the lockstep Monte-Carlo driver spends its time (diagnostic, run by hand on the GPU box)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.batched import BatchedMPC
x_max = load_fixture('cart_pole_with_walls')['x_max']
ctrl = make_controller('cart_pole_with_walls', backend='hip')
bm = BatchedMPC(ctrl)
seeds = tuple(range(64))
bm.closed_loop(np.array([0., 0., 1., 0.]), 2, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=8)
pr = cProfile.Profile()
pr.enable()
st = bm.closed_loop(np.array([0., 0., 1., 0.]), 11, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=8)
pr.disable()
print('wall', st['wall'])
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
