"""Hand-down (hmpc_warm) on the streaming kernel, BASELINE configs[4] (diagnostic): the chain of prefixes of a dive and
their one-flip siblings, every node handed the record of its parent; kernel vs oracle, and the rate with / without."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
from bench import dive_tree

mld, objective, x0 = random_mld()
T, nub, nx = 30, 8, 20
ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
hip, orc = HipBatchedQP(ctrl.problem_data()), OracleBatchedQP(ctrl.problem_data(), threads=16)
Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
leaf = np.full((1, T * nub), -1, np.int8)
for t in range(T):
    r = orc.solve_batch(x0, leaf)
    leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
f, parent = dive_tree(leaf[0])
cold_k, cold_o = hip.solve_batch(x0, f), orc.solve_batch(x0, f)
good = (parent >= 0) & (cold_o['status'][np.maximum(parent, 0)] == 0) & (cold_o['polished'][np.maximum(parent, 0)] > 0)
idx = np.where(good, parent, -1).astype(np.int32)
a = hip.solve_batch(x0, f, warm=(cold_k['primal'], cold_k['dual'], idx))
b = orc.solve_batch(x0, f, warm=(cold_o['primal'], cold_o['dual'], idx))
opt = b['status'] == 0
print('%d nodes, %d handed a parent; status equal %s (vs cold %s); optimal %d; handed-down verified kernel %d oracle %d; iterations kernel %.2f (cold %.2f)'
      % (len(f), int(good.sum()), np.array_equal(a['status'], b['status']), np.array_equal(a['status'], cold_k['status']), int(opt.sum()),
         int(a['handed'].sum()), int((b['polished'] == 64).sum()) if 'polished' in b else -1, a['iters'].mean(), cold_k['iters'].mean()))
both = opt & (a['polished'] > 0) & (b['polished'] > 0)
xs = (T + 1) * nx
print('objective %.1e, x of vertex records %.1e (%d), vs the cold records %.1e'
      % (np.max(np.abs(a['obj'][opt] - b['obj'][opt]) / (1 + np.abs(b['obj'][opt]))), np.abs(a['primal'][both][:, :xs] - b['primal'][both][:, :xs]).max(), int(both.sum()),
         np.abs(a['primal'][both][:, :xs] - cold_k['primal'][both][:, :xs]).max()))
for name, w in (('cold', None), ('handed', (cold_k['primal'], cold_k['dual'], idx))):
    hip.solve_batch(x0, f, warm=w)
    t0 = time.perf_counter(); hip.solve_batch(x0, f, warm=w); dt = time.perf_counter() - t0
    print('%-7s %d nodes: %.1f ms, %.0f QP/s' % (name, len(f), 1e3 * dt, len(f) / dt))
