"""ms per batch with 1 / 2 / 4 waves per node for several batch sizes (diagnostic: the thresholds of hmpc_waves_for)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import time
import numpy as np
from helpers import make_controller, random_prefix_frontier
X0 = np.array([0., 0., 1., 0.])
ctrl = make_controller('cart_pole_with_walls', backend='hip')
for nb in (8, 77, 256, 384, 512, 768, 1024, 1536, 2048, 4096):
    fix = random_prefix_frontier(20, 4, nb, p_one=0.1)
    row = []
    for w in ('1', '2', '4'):
        os.environ['HMPC_WAVES'] = w
        ts = []
        for _ in range(6):
            t = time.perf_counter(); ctrl.qp.solve_batch(X0, fix, want_primal=False, want_dual=False); ts.append(time.perf_counter() - t)
        row.append(1e3 * min(ts))
    print('%5d nodes: 1 wave %.2f ms, 2 waves %.2f ms, 4 waves %.2f ms' % (nb, *row), flush=True)
