"""Diagnostic (GPU box): random MLD nx = 6, nu = 2 + 3, N = 40 (seed 41): the two-wave register kernel against the oracle, several builds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import random_mld, random_prefix_frontier, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
os.environ['HMPC_JIT_SELFCHECK'] = '0'
mld, objective, x0 = random_mld(nx=6, nuc=2, nub=3, seed=41)
T = 40
data = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend()).problem_data()
fix = random_prefix_frontier(T, 3, 192, p_one=0.2, seed0=4000)
fix[0, :] = -1
b = OracleBatchedQP(data, threads=16).solve_batch(x0, fix)
print('oracle statuses', np.bincount(b['status'], minlength=4).tolist(), 'iters max', (b['iters'] & 0xffff).max(), 'unpolished optimal', int(((b['status'] == 0) & (b['polished'] == 0)).sum()))
for label, env in (('default schedule', {'HMPC_JIT_SCHED': 'default'}), ('default schedule, readlane beyond 8 slots', {'HMPC_JIT_SCHED': 'default', 'HMPC_JIT_FLAGS': '-DHMPC_DPP_FEW'}),
                   ('default schedule, -O1', {'HMPC_JIT_SCHED': 'default', 'HMPC_JIT_FLAGS': '-O1'}),
                   ('default schedule, no pair', {'HMPC_JIT_SCHED': 'default', 'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0'}), ('shipped run-time-sized kernel', {'HMPC_JIT': '0'})):
    os.environ.update(env)
    q = HipBatchedQP(data)
    for w in ('2', '4'):
        os.environ['HMPC_WAVES'] = w
        a = q.solve_batch(x0, fix)
        diff = np.flatnonzero(a['status'] != b['status'])
        print('%-34s w%s kinds %s statuses %s differ on %s: kernel %s oracle %s iters kernel %s oracle %s' % (label, w, q.kernel_info(), np.bincount(a['status'], minlength=4).tolist(), diff[:8].tolist(),
              a['status'][diff[:8]].tolist(), b['status'][diff[:8]].tolist(), (a['iters'][diff[:8]] & 0xffff).tolist(), (b['iters'][diff[:8]] & 0xffff).tolist()), flush=True)
    del os.environ['HMPC_WAVES']
    for k in env:
        del os.environ[k]
