"""Diagnostic: configs[4] on the bounds-checked / NaN-poisoned build (make check), sized streaming kernel with the row state in
registers and in the slab: statuses against the oracle.
    python tests/gpu_dev_check_c4.py
"""
import os
import sys
os.environ['HMPC_LIBRARY_NAME'] = 'libhmpc_check.so'
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
ctrl = HybridModelPredictiveController(mld, 30, objective, None, backend=_NoBackend())
orc4 = OracleBatchedQP(ctrl.problem_data(), threads=16)
Cj = np.array([mld.F[52 + 4 * j] for j in range(8)])
leaf = np.full((1, 240), -1, np.int8)
for t in range(30):
    r = orc4.solve_batch(x0, leaf)
    leaf[0, t * 8:(t + 1) * 8] = (r['primal'][0][:31 * 20].reshape(31, 20)[t] @ Cj.T >= 0)
f4 = dive_frontier(leaf[0], 320, 0)
b = orc4.solve_batch(x0, f4)
cases = [('rows in registers', {}), ('rows in the slab', {'HMPC_JIT_SIZED_ROWS': '0'}), ('shipped', {'HMPC_JIT_SIZED': '0'})]
if os.environ.get('DBG_BISECT'):   # which of the four row arrays is read before it is written
    cases = [('poison %s only' % n, {'HMPC_JIT_FLAGS': '-DHMPC_POISON_MASK=%d' % m}) for n, m in (('s + z', 3),)]
    cases.append(('s + z with the finite value 1.0', {'HMPC_JIT_FLAGS': '-DHMPC_POISON_MASK=3 -DHMPC_POISON_FINITE'}))
for label, env in cases:
    os.environ.update(env)
    try:
        hip4 = HipBatchedQP(ctrl.problem_data())
        a = hip4.solve_batch(x0, f4)
        d = np.flatnonzero(a['status'] != b['status'])
        print(label, hip4.kernel_info(), 'status mismatches', len(d), [(int(i), int(a['status'][i]), int(b['status'][i]), int(a['iters'][i]), int(b['iters'][i])) for i in d[:10]], flush=True)
    except Exception as e:
        print(label, 'raised', repr(e)[:400], flush=True)
    for k in env:
        del os.environ[k]
