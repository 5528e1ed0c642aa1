"""Diagnostic (GPU box): the paired solve of the register kernels (kkt_solve_reg_pair) against the build without it (-DHMPC_PAIR=0)
and the oracle on the random MLD of the generic_vs_specialised workload: statuses, iteration counts, rays of infeasible nodes."""
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
mld, obj, x0 = random_mld(nx=6, nuc=2, nub=3, seed=3)
T = 12
c = HybridModelPredictiveController(mld, T, obj, None, backend=_NoBackend())
f = random_prefix_frontier(T, 3, 2048, p_one=0.3)
f[0, :] = -1
data = c.problem_data()
b = OracleBatchedQP(data, threads=16).solve_batch(x0, f)
pair = HipBatchedQP(data)
os.environ['HMPC_JIT_FLAGS'] = '-DHMPC_PAIR=0'
single = HipBatchedQP(data)
del os.environ['HMPC_JIT_FLAGS']
os.environ['HMPC_WAVES'] = '1'
a, s = pair.solve_batch(x0, f), single.solve_batch(x0, f)
inf = b['status'] == 1
for tag, r in (('pair', a), ('single', s)):
    print(tag, 'statuses equal', np.array_equal(r['status'], b['status']), 'iters equal', int((r['iters'] == b['iters']).sum()), 'of', len(f),
          'mean iters', (r['iters'] & 0xffff).mean(), 'oracle', (b['iters'] & 0xffff).mean())
    dev = np.max(np.abs(r['dual'][inf] - b['dual'][inf]), axis=1)
    bad = np.flatnonzero(inf)[dev > 1e-6]
    print(tag, 'infeasible nodes whose ray differs from the oracle by > 1e-6:', len(bad), bad[:12].tolist(), 'their iters', (r['iters'][bad[:12]] & 0xffff).tolist(), 'oracle', (b['iters'][bad[:12]] & 0xffff).tolist(),
          'weak', r['weak'][bad[:12]].tolist(), b['weak'][bad[:12]].tolist())
    dobj = np.abs(r['dual_obj'][inf] - b['dual_obj'][inf])
    print(tag, 'dual objective of the rays: worst difference', dobj.max())
dev = np.max(np.abs(a['dual'][inf] - b['dual'][inf]), axis=1)
bad = np.flatnonzero(inf)[dev > 1e-6]
if len(bad):
    k = int(bad[0])
    print('node', k, 'fixed', int((f[k] >= 0).sum()), 'ray pair / single / oracle (largest entries):')
    for tag, r in (('pair', a), ('single', s), ('oracle', b)):
        row = r['dual'][k]
        top = np.argsort(-np.abs(row))[:8]
        print(' ', tag, [(int(i), round(float(row[i]), 6)) for i in top])
    os.environ['HMPC_TRACE'] = '1'
    for tag, fl in (('pair', ''), ('single', '-DHMPC_PAIR=0')):
        os.environ['HMPC_JIT_FLAGS'] = fl
        q = HipBatchedQP(data)
        print('--- trace', tag, flush=True)
        sys.stderr.flush()
        r = q.solve_batch(x0, f[k:k + 1])
        sys.stderr.flush()
        print('status', r['status'], 'iters', r['iters'] & 0xffff, flush=True)
