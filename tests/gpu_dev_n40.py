"""Diagnostic (not a test): cart-pole N = 40, random-prefix frontier p = 0.5 of the bench -- iterations and statuses of the
kernel compiled with the problem's sizes against the shipped kernel (HMPC_JIT_SIZED=0), and the slowest nodes.

    python tests/gpu_dev_n40.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier

T = 40
X0 = np.array([0., 0., 1., 0.])
fix = random_prefix_frontier(T, 4, 2048, p_one=float(os.environ.get('DBG_P', '0.5')))
res = {}
for label, env in (('sized', {}), ('shipped', {'HMPC_JIT_SIZED': '0'})):
    os.environ.update(env)
    hip = make_controller('cart_pole_with_walls', T=T, backend='hip')
    for k in env:
        del os.environ[k]
    hip.qp.solve_batch(X0, fix)
    t0 = time.perf_counter()
    r = hip.qp.solve_batch(X0, fix)
    dt = time.perf_counter() - t0
    res[label] = r
    it = r['iters']
    print('%-8s kinds %s %.2f ms; status %s; iters mean %.2f max %d; second solves %d; weak %d; top nodes %s' % (
        label, hip.qp.kernel_info(), 1e3 * dt, dict(zip(*np.unique(r['status'], return_counts=True))), it.mean(), it.max(),
        int(r['second'].sum()) if 'second' in r else -1, int(r['weak'].sum()), [(int(i), int(it[i]), int(r['status'][i])) for i in np.argsort(-it)[:6]]), flush=True)
a, b = res['sized'], res['shipped']
d = np.flatnonzero(a['iters'] != b['iters'])
print('nodes whose iteration counts differ: %d; largest differences %s' % (len(d), [(int(i), int(a['iters'][i]), int(b['iters'][i])) for i in d[np.argsort(-np.abs(a['iters'][d] - b['iters'][d]))[:8]]]))
print('statuses equal', np.array_equal(a['status'], b['status']))
