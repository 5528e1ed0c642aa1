import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier
T = int(os.environ.get('DBG_T', 20))
ch = make_controller(T=T, backend='hip')
x0 = np.array([0., 0., 1., 0.])
for nb in [int(v) for v in os.environ.get('DBG_BS', '1,64,512,1024,4096').split(',')]:
    for p_one, tag in [(0.5, 'p.5'), (0.1, 'p.1')]:
        fix = random_prefix_frontier(T, 4, nb, p_one=p_one)
        ch.qp.solve_batch(x0, fix, want_primal=False, want_dual=False)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = ch.qp.solve_batch(x0, fix); ts.append(time.perf_counter() - t)
        print(nb, tag, 'best %.3f ms' % (1e3 * min(ts)), 'QP/s %.0f' % (nb / min(ts)), 'iters mean %.1f' % r['iters'].mean(), 'feasible', int((r['status'] == 0).sum()), 'bad', int((r['status'] > 1).sum()), ch.qp.launch_info(), flush=True)
