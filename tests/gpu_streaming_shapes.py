"""The streaming form of the generic kernel on shapes other than BASELINE configs[4] (diagnostic, hand-run on the GPU box;
`HMPC_LIBRARY_NAME=libhmpc_check.so` runs it on the bounds-checked / NaN-poisoned build): state and input counts that are
not multiples of four (padded blocks, partial batches of the matrix-core tiles), nu = 16 (the panel's limit), more than 32
dense rows per stage (the batched operand loads), nz > 48 and nu > 16 (the LDS form of the factorisation), against the
oracle.  Forced into the streaming form (HMPC_FORCE_BIG) where the problem would fit LDS."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import random_mld, random_prefix_frontier, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP

bad = 0
for nx, nuc, nub, T, seed in ((20, 6, 8, 30, 0), (13, 3, 5, 12, 1), (17, 9, 7, 10, 2), (8, 8, 8, 10, 3), (30, 4, 6, 8, 4), (22, 2, 3, 9, 5),
                              (32, 6, 6, 6, 6), (32, 10, 8, 6, 10), (14, 10, 8, 8, 11), (12, 12, 8, 8, 7), (16, 4, 0, 8, 8), (9, 3, 4, 14, 9)):   # (the oracle takes nx <= 32, nz <= 64)
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
    ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    os.environ['HMPC_FORCE_BIG'] = '1'
    try:
        hip = HipBatchedQP(ctrl.problem_data())
    except RuntimeError as e:
        print('skip nx=%d nu=%d+%d T=%d: %s' % (nx, nuc, nub, T, str(e)[:80]))
        continue
    finally:
        del os.environ['HMPC_FORCE_BIG']
    orc = OracleBatchedQP(ctrl.problem_data(), threads=16)
    # a dive towards a feasible leaf (binaries from the sign test of the generator), its prefixes with and without a flip
    count = 40
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
    a, b = hip.solve_batch(x0, fix), orc.solve_batch(x0, fix)
    same = np.array_equal(a['status'], b['status'])
    opt = (a['status'] == 0) & (b['status'] == 0)
    dobj = np.max(np.abs(a['obj'][opt] - b['obj'][opt]) / (1 + np.abs(b['obj'][opt]))) if opt.any() else 0.
    both = opt & (a['polished'] > 0) & (b['polished'] > 0)
    xs = (T + 1) * nx
    dev = np.abs(a['primal'][both][:, :xs] - b['primal'][both][:, :xs]).max() if both.any() else 0.
    nan = int(np.isnan(a['primal'][a['status'] == 0]).sum() + np.isnan(a['dual']).sum())
    good = same and dobj < 2e-6 and dev < 1e-5 and nan == 0 and np.all(a['status'] <= 1)
    bad += not good
    print('ok  ' if good else 'FAIL', 'nx=%d nu=%d+%d T=%d (nz=%d): LDS %d B, optimal %d (polished on both sides %d), infeasible %d, status equal %s, objective %.1e, x %.1e, NaNs %d'
          % (nx, nuc, nub, T, nx + nuc + nub, hip.launch_info()[1], int(opt.sum()), int(both.sum()), int((a['status'] == 1).sum()), same, dobj, dev, nan), flush=True)
print('STREAMING SHAPES:', 'all clean' if bad == 0 else '%d FAILURES' % bad)
