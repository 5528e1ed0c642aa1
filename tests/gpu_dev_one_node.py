import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import conftest
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
from bench import dive_frontier
mld, objective, x0 = random_mld()
T, nub, nx = 30, 8, 20
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
orc = OracleBatchedQP(ctrl.problem_data(), threads=16)
Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
leaf = np.full((1, T * nub), -1, np.int8)
for t in range(T):
    r = orc.solve_batch(x0, leaf)
    leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
f = dive_frontier(leaf[0], 4096, 0)[int(sys.argv[1]):int(sys.argv[1]) + 1]
os.environ['HMPC_TRACE'] = '1'
r = HipBatchedQP(ctrl.problem_data()).solve_batch(x0, f)
print(r['status'], r['iters'] & 0xffff, r['polished'])
