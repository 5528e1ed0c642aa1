"""Diagnostic: a problem whose compiled kernels the first-use check dropped (random MLD nx=9, nu=3+4, T=6, seed 23): the compiled
kernels WITHOUT the check against the shipped kernel and the oracle, node by node.  (Reproduces on the tree BEFORE the static row
map was cut back to nx + nu <= 15 -- commit "Register kernels end at nx + nu = 15"; since then the problem runs the run-time-sized
kernel with sizes and everything here agrees.  Output of the time: profiles/r04_nz16_register_kernels.txt.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
nx, nuc, nub, T, seed = (int(v) for v in os.environ.get('DBG_SPEC', '9,3,4,6,23').split(','))
mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
os.environ['HMPC_JIT_SELFCHECK'] = '0'
hip = HipBatchedQP(ctrl.problem_data())
os.environ['HMPC_JIT'] = '0'
plain = HipBatchedQP(ctrl.problem_data())
del os.environ['HMPC_JIT']
orc = OracleBatchedQP(ctrl.problem_data(), threads=16)
count = 96
fix = np.full((count, T * nub), -1, np.int8)
Cj = np.array([mld.F[2 * nx + 2 * nuc + 4 * j] for j in range(nub)])
leaf = np.full((1, T * nub), -1, np.int8)
for t in range(T):
    r = orc.solve_batch(x0, leaf)
    if r['status'][0] != 0:
        break
    leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
rng = np.random.default_rng(seed)
for k in range(1, count):
    d = int(rng.integers(1, T * nub + 1))
    fix[k, :d] = leaf[0, :d]
    if k % 2 == 0:
        j = int(rng.integers(0, d))
        if fix[k, j] >= 0:
            fix[k, j] = 1 - fix[k, j]
b = orc.solve_batch(x0, fix)
print('kinds compiled', hip.kernel_info(), 'shipped', plain.kernel_info())
for waves in ('1', '2', '4'):
    os.environ['HMPC_WAVES'] = waves
    a, c = hip.solve_batch(x0, fix), plain.solve_batch(x0, fix)
    a6, c6 = hip.solve_batch(x0, fix[:6]), plain.solve_batch(x0, fix[:6])
    del os.environ['HMPC_WAVES']
    print('waves', waves, 'status compiled == oracle', np.array_equal(a['status'], b['status']), 'shipped == oracle', np.array_equal(c['status'], b['status']))
    print('   first six nodes alone: status compiled', a6['status'], 'shipped', c6['status'], 'oracle', b['status'][:6])
    print('   obj compiled', a6['obj'], '\n   obj shipped ', c6['obj'], '\n   obj oracle  ', b['obj'][:6])
    print('   iters compiled', a6['iters'], 'shipped', c6['iters'], 'oracle', b['iters'][:6], 'polished', a6['polished'], c6['polished'], b['polished'][:6])
