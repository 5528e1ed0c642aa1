"""Phase cycle stamps of the generic streaming kernel on BASELINE configs[4] (diagnostic; needs `make stamps`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import time
import numpy as np
import warm_start_hmpc_amd.qp_backend as qb
if os.environ.get('DBG_STAMPS', '1') == '1':
    qb.LIBRARY_PATH = qb.LIBRARY_PATH.replace('libhmpc.so', 'libhmpc_stamps.so')
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
mld, objective, x0 = random_mld()
T = int(os.environ.get('DBG_T', 30))
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
qp = qb.HipBatchedQP(ctrl.problem_data(), max_iter=int(os.environ.get('DBG_ITERS', 100)))
B = int(os.environ.get('DBG_B', 256))
fix = np.full((B, T * 8), -1, np.int8)
rng = np.random.default_rng(0)
for k in range(1, B):
    fix[k, :int(rng.integers(0, 40))] = 0
r = qp.solve_batch(x0, fix)
t = time.perf_counter(); r = qp.solve_batch(x0, fix); dt = time.perf_counter() - t
print('B', B, 'time %.1f ms' % (1e3 * dt), 'QP/s %.0f' % (B / dt), 'iters', r['iters'][:4], 'status', np.unique(r['status'], return_counts=True), 'launch', qp.launch_info())
