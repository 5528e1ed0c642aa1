"""Development loop of the streaming kernel on BASELINE configs[4] (diagnostic, not a test): parity of a dive frontier
against the oracle, rate of a full launch, phase stamps of node 0 when the library is a stamps build.

    HMPC_LIBRARY_NAME=libhmpc_dev.so HMPC_TRACE=1 python tests/gpu_dev_cfg4.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
from bench import dive_frontier

mld, objective, x0 = random_mld()
T, nub, nx, nu = 30, 8, 20, 14
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
hip, orc = HipBatchedQP(ctrl.problem_data()), OracleBatchedQP(ctrl.problem_data(), threads=16)
Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
leaf = np.full((1, T * nub), -1, np.int8)
for t in range(T):
    r = orc.solve_batch(x0, leaf)
    leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
NP = int(os.environ.get('DBG_PARITY', 96))
B = int(os.environ.get('DBG_B', 1024))
f = dive_frontier(leaf[0], max(B, NP), 0)
a, b = hip.solve_batch(x0, f[:NP]), orc.solve_batch(x0, f[:NP])
same = np.array_equal(a['status'], b['status'])
opt = b['status'] == 0
dobj = np.max(np.abs(a['obj'][opt] - b['obj'][opt]) / (1 + np.abs(b['obj'][opt]))) if opt.any() else 0.
xs = (T + 1) * nx
pa, pb = a['primal'][opt], b['primal'][opt]
# states and continuous inputs (the relaxed binaries are not unique: they are not in the cost)
ua, ub = pa[:, xs:].reshape(-1, T, nu)[:, :, :nu - nub], pb[:, xs:].reshape(-1, T, nu)[:, :, :nu - nub]
dev = max(np.max(np.abs(pa[:, :xs] - pb[:, :xs])), np.max(np.abs(ua - ub))) if opt.any() else 0.
pol_a, pol_b = a['polished'][opt] != 0, b['polished'][opt] != 0
both = pol_a & pol_b
devp = max(np.max(np.abs(pa[both][:, :xs] - pb[both][:, :xs])), np.max(np.abs(ua[both] - ub[both]))) if both.any() else 0.
print('parity on %d nodes: status equal %s (%d optimal), objective %.2e, trajectories %.2e abs (%.2e where both polished); polished kernel %d oracle %s; '
      'iterations kernel %.1f' % (NP, same, int(opt.sum()), dobj, dev, devp, int(pol_a.sum()), int(pol_b.sum()),
                                  float(np.mean(a['iters']))))
if not same:
    print('  status kernel', a['status'], '\n  status oracle', b['status'])
hip.solve_batch(x0, f[:B])
t0 = time.perf_counter()
r = hip.solve_batch(x0, f[:B])
dt = time.perf_counter() - t0
print('B %d: %.1f ms, %.0f QP/s; launch %s; statuses %s' % (B, 1e3 * dt, B / dt, hip.launch_info(), np.unique(r['status'], return_counts=True)))
if os.environ.get('DBG_SWAP'):   # phase stamps of another node than the root: put it first
    k = int(os.environ['DBG_SWAP'])
    g = f[:B].copy(); g[[0, k]] = g[[k, 0]]
    r2 = hip.solve_batch(x0, g)
    print('node %d first: iterations %d polished %d status %d' % (k, r2['iters'][0], r2['polished'][0], r2['status'][0]))
