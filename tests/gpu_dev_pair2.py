"""Diagnostic (GPU box): variants of the build without the paired solve, and the rays of the paired build, on the random MLD of
the generic_vs_specialised workload (see gpu_dev_pair.py)."""
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
f = random_prefix_frontier(T, 3, 256, p_one=0.3)
f[0, :] = -1
data = c.problem_data()
b = OracleBatchedQP(data, threads=16).solve_batch(x0, f)
inf = b['status'] == 1
os.environ['HMPC_WAVES'] = '1'
VARIANTS = (('pair', {}), ('single', {'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0'}), ('single, passes on', {'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0', 'HMPC_JIT_SAFE': '0'}),
            ('single, default schedule', {'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0', 'HMPC_JIT_SCHED': 'default'}),
            ('single, -O1', {'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0 -O1'}), ('pair, default schedule', {'HMPC_JIT_SCHED': 'default'}), ('pair, -O1', {'HMPC_JIT_FLAGS': '-O1'}),
            ('pair, NaN-poisoned', {'HMPC_JIT_FLAGS': '-DHMPC_CHECK'}), ('single, NaN-poisoned', {'HMPC_JIT_FLAGS': '-DHMPC_CHECK -DHMPC_PAIR=0'}),
            ('pair, poisoned with 1.0', {'HMPC_JIT_FLAGS': '-DHMPC_CHECK -DHMPC_POISON_FINITE'}), ('single, poisoned with 1.0', {'HMPC_JIT_FLAGS': '-DHMPC_CHECK -DHMPC_POISON_FINITE -DHMPC_PAIR=0'}))
VARIANTS += (('pair, no occ2', {'HMPC_JIT_NO_OCC2': '1'}), ('single, no occ2', {'HMPC_JIT_NO_OCC2': '1', 'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0'}),
             ('pair, 2 waves', {'HMPC_WAVES': '2'}), ('pair, 4 waves', {'HMPC_WAVES': '4'}), ('single, 2 waves', {'HMPC_WAVES': '2', 'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0'}),
             ('single, 4 waves', {'HMPC_WAVES': '4', 'HMPC_JIT_FLAGS': '-DHMPC_PAIR=0'}))
if os.environ.get('DBG_ONLY'):
    VARIANTS = tuple(v for v in VARIANTS if v[0] in os.environ['DBG_ONLY'].split(';'))
for label, env in VARIANTS:
    keep = os.environ.get('HMPC_WAVES')
    os.environ.update(env)
    q = HipBatchedQP(data)
    r = q.solve_batch(x0, f)
    for k in env:
        del os.environ[k]
    if keep:
        os.environ['HMPC_WAVES'] = keep
    dev = np.max(np.abs(r['dual'][inf] - b['dual'][inf]), axis=1)
    big = np.max(np.abs(r['dual'][inf]), axis=1)
    print('%-28s statuses equal %s, iterations equal on %d of %d, rays off by > 1e-6 on %d of %d nodes (largest entry of the worst ray %.3e), dual objective off by %.2e'
          % (label, np.array_equal(r['status'], b['status']), int(((r['iters'] & 0xffff) == (b['iters'] & 0xffff)).sum()), len(f), int((dev > 1e-6).sum()), int(inf.sum()),
             big[np.argmax(dev)] if inf.any() else 0., np.abs(r['dual_obj'][inf] - b['dual_obj'][inf]).max()),
          'NaNs in the rays', int(np.isnan(r['dual'][inf]).sum()), 'statuses', np.bincount(r['status'], minlength=4).tolist(), flush=True)
