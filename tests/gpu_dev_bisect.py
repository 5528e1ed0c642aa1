"""Diagnostic (GPU box): which compiler pass makes a compiled kernel come out wrong (profiles/r04_miscompiled_variants.txt).
Bisects -mllvm -opt-bisect-limit=N over the compilation of ONE kernel of one random MLD and runs six nodes against the oracle.

    DBG_SHAPE=4,4,7,10,61 DBG_WAVES=1 [DBG_FLAGS='...'] python tests/gpu_dev_bisect.py
"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = tuple(int(v) for v in os.environ.get('DBG_SHAPE', '4,4,7,10,61').split(','))
waves = os.environ.get('DBG_WAVES', '1')
base_flags = os.environ.get('DBG_FLAGS', '')

ONE = r'''
import os, sys
sys.path.insert(0, %r)
import conftest
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
nx, nuc, nub, T, seed = %r
mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
fix = np.full((6, T * nub), -1, np.int8)
fix[1, :nub] = 0; fix[2, :2 * nub] = 0; fix[3, 0] = 1; fix[4, :3] = 0; fix[5, :T * nub // 2] = 0
if os.environ.get('DBG_PREFIXES'):      # random prefixes instead (statuses AND the rays of the infeasible nodes are compared)
    from helpers import random_prefix_frontier
    fix = random_prefix_frontier(T, nub, int(os.environ['DBG_PREFIXES']), p_one=0.3)
    fix[0, :] = -1
b = OracleBatchedQP(ctrl.problem_data(), threads=8).solve_batch(x0, fix)
hip = HipBatchedQP(ctrl.problem_data())
a = hip.solve_batch(x0, fix)
inf = b['status'] == 1
same = np.array_equal(a['status'], b['status'])
rays = same and (not inf.any() or np.nanmax(np.abs(a['dual'][inf] - b['dual'][inf])) < 1e-6) and not np.isnan(a['dual'][inf]).any()
print('RESULT', 'ok' if same and rays else 'WRONG', np.bincount(a['status'], minlength=4).tolist(), 'statuses equal', same, 'rays equal', bool(rays), hip.kernel_info())
''' % (HERE, spec)


def run(limit):
    env = dict(os.environ)
    env['HMPC_JIT_SELFCHECK'] = '0'
    env['HMPC_WAVES'] = waves
    env['HMPC_JIT_ONLY_WAVES'] = waves
    flags = base_flags
    if limit is not None:
        flags += ' -mllvm -opt-bisect-limit=%d' % limit
    env['HMPC_JIT_FLAGS'] = flags.strip()
    try:
        p = subprocess.run([sys.executable, '-c', ONE], env=env, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return 'TIMEOUT'
    out = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
    return out[0] if out else 'FAILED ' + p.stderr[-300:].replace('\n', ' | ')


full = run(None)
print('no limit:', full, flush=True)
if 'WRONG' not in full:
    print('the kernel is right without a limit: nothing to bisect')
    sys.exit(0)
lo, hi = 0, int(os.environ.get('DBG_HI', 13000))   # lo: right (or not wrong), hi: wrong
r0 = run(lo)
print('limit 0:', r0, flush=True)
if 'WRONG' in r0:
    print('wrong even with every optional pass skipped')
    sys.exit(0)
while hi - lo > 1:
    mid = (lo + hi) // 2
    r = run(mid)
    print('limit %d: %s' % (mid, r), flush=True)
    if 'WRONG' in r:
        hi = mid
    else:
        lo = mid
print('first wrong limit:', hi)
