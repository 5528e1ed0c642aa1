"""Diagnostic (GPU box): the tolerance escalation on BASELINE configs[4] -- kernel vs oracle on the 4096-node dive frontier:
which optimal nodes end unpolished on either side, at which attempt the others verified, and, for the first few
that differ, the interior-point traces of both (HMPC_TRACE / ORACLE_QP_TRACE).

    python tests/gpu_dev_escalation.py [nodes]
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

mld, objective, x0 = random_mld()
T, nub, nx, nu = 30, 8, 20, 14
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
hip, orc = HipBatchedQP(ctrl.problem_data()), OracleBatchedQP(ctrl.problem_data(), threads=os.cpu_count() or 8)
Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
leaf = np.full((1, T * nub), -1, np.int8)
for t in range(T):
    r = orc.solve_batch(x0, leaf)
    leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
f = dive_frontier(leaf[0], B, 0)
a, b = hip.solve_batch(x0, f), orc.solve_batch(x0, f)
opt = (a['status'] == 0) & (b['status'] == 0)
xs = (T + 1) * nx
dev = np.abs(a['primal'][:, :xs] - b['primal'][:, :xs]).max(axis=1)
print('%d nodes, status mismatches %d, optimal %d; polished kernel %d oracle %d; oracle attempt numbers %s'
      % (B, int((a['status'] != b['status']).sum()), int(opt.sum()), int((opt & (a['polished'] > 0)).sum()), int((opt & (b['polished'] > 0)).sum()),
         np.bincount(b['polished'][opt]).tolist()))
print('iterations (optimal nodes): kernel %.2f oracle %.2f; worst state deviation %.2e (both polished: %.2e)'
      % ((a['iters'][opt] & 0xFFFF).mean(), (b['iters'][opt] & 0xFFFF).mean(), dev[opt].max(), dev[opt & (a['polished'] > 0) & (b['polished'] > 0)].max()))
odd = np.flatnonzero(opt & ((a['polished'] == 0) | (b['polished'] == 0) | (dev > 1e-6)))
for i in odd[:40]:
    print('  node %4d: polished %d/%d iters %d/%d obj %.10f / %.10f dev %.2e' % (i, a['polished'][i], b['polished'][i], a['iters'][i] & 0xFFFF, b['iters'][i] & 0xFFFF,
                                                                              a['obj'][i], b['obj'][i], dev[i]))
os.environ['HMPC_TRACE'] = '1'
hip2 = HipBatchedQP(ctrl.problem_data())
os.environ['ORACLE_QP_TRACE'] = '1'
orc1 = OracleBatchedQP(ctrl.problem_data(), threads=1)
for i in odd[:2]:
    print('---- node %d, kernel trace then oracle trace' % i, flush=True)
    sys.stderr.flush()
    hip2.solve_batch(x0, f[i:i + 1])
    sys.stderr.flush()
    orc1.solve_batch(x0, f[i:i + 1])
    sys.stderr.flush()
