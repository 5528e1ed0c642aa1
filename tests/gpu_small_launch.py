"""Diagnostic: where the time of a small launch goes -- kernel alone (device API, HIP events) against the host-pointer
call, with the iteration counts of the nodes in it (the launch lasts as long as its slowest node)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import time
import numpy as np
import torch
from helpers import make_controller, random_prefix_frontier, load_fixture
X0 = np.array([0., 0., 1., 0.])
ctrl = make_controller('cart_pole_with_walls', backend='hip')
dev = torch.device('cuda', 0)
g = load_fixture('qp_golden')
tree = g['n20tree_fix']
for label, fix in (('root only', np.full((1, 80), -1, np.int8)), ('8 random p=0.1', random_prefix_frontier(20, 4, 8, p_one=0.1)),
                   ('8 nodes of a real tree', tree[40:48]), ('77 nodes of a real tree', tree[60:137]), ('8 infeasible', None)):
    if fix is None:
        pool = random_prefix_frontier(20, 4, 400, p_one=0.5)
        st = ctrl.qp.solve_batch(X0, pool)['status']
        fix = pool[st == 1][:8]
    B = len(fix)
    r = ctrl.qp.solve_batch(X0, fix)
    ts = []
    for _ in range(8):
        t = time.perf_counter(); ctrl.qp.solve_batch(X0, fix, want_primal=False, want_dual=False); ts.append(time.perf_counter() - t)
    ts_full = []
    for _ in range(8):
        t = time.perf_counter(); ctrl.qp.solve_batch(X0, fix); ts_full.append(time.perf_counter() - t)
    fx, x0 = torch.from_numpy(np.ascontiguousarray(fix)).to(dev), torch.from_numpy(X0).to(dev)
    out = dict(obj=torch.empty(B, dtype=torch.float64, device=dev), dual_obj=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev),
               primal=torch.empty(B, ctrl.qp.n_primal, dtype=torch.float64, device=dev), dual=torch.empty(B, ctrl.qp.n_dual, dtype=torch.float64, device=dev))
    ks = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctrl.qp.solve_batch_device(x0, fx, out); b.record(); torch.cuda.synchronize()
        ks.append(a.elapsed_time(b))
    it = r['iters'] & 0xFFFF
    print('%-24s: kernel %.3f ms | host call without records %.3f ms, with records %.3f ms | iterations max %d mean %.1f, optimal %d, per iteration of the slowest node %.1f us'
          % (label, min(ks), 1e3 * min(ts), 1e3 * min(ts_full), it.max(), it.mean(), int((r['status'] == 0).sum()), 1e3 * min(ks) / it.max()), flush=True)
