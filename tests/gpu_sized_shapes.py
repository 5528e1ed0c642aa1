"""The kernels compiled with the problem's sizes (csrc/hmpc_jit.h) on a spread of random MLD shapes (diagnostic,
hand-run on the GPU box): shapes the static row map holds (register kernels with exact row slots), shapes beyond it that fit
LDS (the run-time-sized kernel with sizes, 1 / 2 / 4 waves per node) and shapes beyond one CU's LDS (the streaming form) --
each against the oracle and against the shipped kernel of the same problem (HMPC_JIT_SIZED=0, HMPC_JIT=0).  Every shape costs
one compilation on the box (~10 - 40 s).

    python tests/gpu_sized_shapes.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import random_mld, random_prefix_frontier, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP

bad = 0
SHAPES = ((5, 2, 2, 9, 21), (7, 3, 3, 14, 22), (9, 3, 4, 6, 23), (4, 1, 5, 16, 24),          # register kernels
          (12, 4, 4, 10, 25), (13, 3, 5, 7, 26), (16, 4, 0, 8, 8), (10, 6, 6, 9, 27),         # run-time-sized kernel with sizes
          (22, 2, 3, 24, 5), (17, 9, 7, 20, 2), (30, 4, 6, 16, 4))                            # streaming form
# DBG_EDGE=1: small and lopsided shapes instead -- one state, no continuous input, one binary, many binaries, the shortest
# horizons, the last admissible nx + nu of the register kernels (15), horizons that outgrow the row slots
if os.environ.get('DBG_EDGE'):
    SHAPES = ((1, 1, 1, 4, 31), (2, 0, 2, 5, 32), (3, 1, 1, 2, 33), (6, 0, 3, 8, 34), (11, 2, 2, 5, 35), (5, 5, 5, 3, 36), (14, 0, 1, 6, 37),
              (3, 3, 6, 12, 38), (2, 1, 8, 4, 39), (8, 2, 2, 30, 40), (6, 2, 3, 40, 41), (4, 2, 1, 25, 42))
if os.environ.get('DBG_SHAPES'):   # "nx,nuc,nub,T,seed;..."
    SHAPES = tuple(tuple(int(v) for v in q.split(',')) for q in os.environ['DBG_SHAPES'].split(';'))
for nx, nuc, nub, T, seed in SHAPES:
    try:
        mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
        ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    except Exception as e:
        print('skip nx=%d nu=%d+%d T=%d (generator): %s' % (nx, nuc, nub, T, repr(e)[:80]))
        continue
    tic = time.perf_counter()
    try:
        hip = HipBatchedQP(ctrl.problem_data())
    except (RuntimeError, ValueError) as e:
        print('skip nx=%d nu=%d+%d T=%d: %s' % (nx, nuc, nub, T, str(e)[:80]))
        continue
    tcreate = time.perf_counter() - tic
    os.environ['HMPC_JIT'] = '0'
    try:
        plain = HipBatchedQP(ctrl.problem_data())
    finally:
        del os.environ['HMPC_JIT']
    orc = OracleBatchedQP(ctrl.problem_data(), threads=16)
    count = 96
    fix = np.full((count, T * max(nub, 1)), -1, np.int8)[:, :T * nub]
    if nub:
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
    line = []
    good = True
    for waves in ('1', '2', '4'):
        os.environ['HMPC_WAVES'] = waves
        try:
            a, c = hip.solve_batch(x0, fix), plain.solve_batch(x0, fix)
        finally:
            del os.environ['HMPC_WAVES']
        same = np.array_equal(a['status'], b['status']) and np.array_equal(a['status'], c['status'])
        opt = (a['status'] == 0) & (b['status'] == 0)
        dobj = np.max(np.abs(a['obj'][opt] - b['obj'][opt]) / (1 + np.abs(b['obj'][opt]))) if opt.any() else 0.
        dship = np.max(np.abs(a['obj'][opt] - c['obj'][opt]) / (1 + np.abs(c['obj'][opt]))) if opt.any() else 0.
        both = opt & (a['polished'] > 0) & (b['polished'] > 0)
        xs = (T + 1) * nx
        dev = np.abs(a['primal'][both][:, :xs] - b['primal'][both][:, :xs]).max() if both.any() else 0.
        nan = int(np.isnan(a['primal'][a['status'] == 0]).sum() + np.isnan(a['dual']).sum())
        ok = same and dobj < 2e-6 and dship < 2e-6 and dev < 1e-5 and nan == 0 and np.all(a['status'] <= 1)
        # the hand-down instantiation of the same kernels (hmpc_warm; the first-use check runs the cold kernel only): every
        # optimal, polished node is handed ITS OWN record -- its active set verifies without an interior-point iteration --,
        # every other node the record of its neighbour in the batch (mostly the wrong set: dropped, cold solve)
        idx = np.where((a['status'] == 0) & (a['polished'] > 0), np.arange(count), np.where(a['status'][(np.arange(count) + 1) % count] == 0, (np.arange(count) + 1) % count, -1)).astype(np.int32)
        os.environ['HMPC_WAVES'] = waves
        try:
            aw = hip.solve_batch(x0, fix, warm=(a['primal'], a['dual'], idx))
            cw = plain.solve_batch(x0, fix, warm=(c['primal'], c['dual'], idx))
        finally:
            del os.environ['HMPC_WAVES']
        own = (a['status'] == 0) & (a['polished'] > 0)
        okw = np.array_equal(aw['status'], a['status']) and np.array_equal(cw['status'], a['status']) and (aw['iters'][own] == 0).mean() > 0.9 and \
            (not opt.any() or np.max(np.abs(aw['obj'][opt] - a['obj'][opt]) / (1 + np.abs(a['obj'][opt]))) < 2e-6)
        ok = ok and okw
        good = good and ok
        line.append('w%s: obj %.0e / shipped %.0e, x %.0e, handed down %d of %d%s' % (waves, dobj, dship, dev, int((aw['iters'][own] == 0).sum()), int(own.sum()), '' if ok else ' FAIL'))
    bad += not good
    print('ok  ' if good else 'FAIL', 'nx=%d nu=%d+%d T=%d: kinds %s (create %.0f s), optimal %d (polished on both sides %d), infeasible %d; %s'
          % (nx, nuc, nub, T, hip.kernel_info(), tcreate, int(opt.sum()), int(both.sum()), int((a['status'] == 1).sum()), '; '.join(line)), flush=True)
print('SIZED SHAPES:', 'all clean' if bad == 0 else '%d FAILURES' % bad)
