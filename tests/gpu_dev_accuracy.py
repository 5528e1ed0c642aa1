"""Diagnostic (GPU box): accuracy of the interior-point iteration of the run-time-sized kernels in late iterations --
dual residual per iteration of one node, kernel (HMPC_TRACE) and oracle (ORACLE_QP_TRACE), polish off, tol 1e-12.

    python tests/gpu_dev_accuracy.py c4 NODE [refine]      BASELINE configs[4], node NODE of the dive frontier (streaming form)
    [HMPC_FORCE_BIG=1] python tests/gpu_dev_accuracy.py small SEED [refine]   random MLD nx=13 nuc=3 nub=5 N=12, root node
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
from bench import dive_frontier

which, arg = sys.argv[1], int(sys.argv[2])
refine = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
if which == 'c4':
    mld, objective, x0 = random_mld()
    T, nub, nx = 30, 8, 20
else:
    nx, nuc, nub, T = 13, 3, 5, 12
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=arg)
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
if which == 'c4':
    orc = OracleBatchedQP(ctrl.problem_data(), threads=os.cpu_count() or 8)
    Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
    leaf = np.full((1, T * nub), -1, np.int8)
    for t in range(T):
        r = orc.solve_batch(x0, leaf)
        leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
    f = dive_frontier(leaf[0], 4096, 0)[arg:arg + 1]
else:
    f = np.full((1, T * nub), -1, np.int8)
os.environ['HMPC_TRACE'] = '1'
hip = HipBatchedQP(ctrl.problem_data(), tol=1e-12, polish=False, refine=refine)
os.environ['ORACLE_QP_TRACE'] = '1'
orc1 = OracleBatchedQP(ctrl.problem_data(), threads=1, tol=1e-12, polish=False, refine=refine)
a = hip.solve_batch(x0, f)
sys.stderr.flush()
b = orc1.solve_batch(x0, f)
sys.stderr.flush()
print('RESULT %s %d refine %d launch %s: status %d/%d iters %d/%d obj %.12f / %.12f state dev %.2e'
      % (which, arg, refine, hip.launch_info(), a['status'][0], b['status'][0], a['iters'][0] & 0xFFFF, b['iters'][0], a['obj'][0], b['obj'][0],
         np.abs(a['primal'][0][:(T + 1) * nx] - b['primal'][0][:(T + 1) * nx]).max()))
