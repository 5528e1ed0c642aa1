"""Diagnostic (hand-run on the GPU box): what a node costs by how its solve goes -- cold; handed its parent's record and verified
(HMPC_ITERS_HANDED); handed the record but not verified (the attempt, then the cold solve).  Nodes of the headline tree, each class
tiled to 4096 nodes and timed by itself.  python tests/gpu_dev_handdown_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import torch
from helpers import make_controller, load_fixture
import bench

dev = torch.device('cuda', 0)
ctrl = make_controller('cart_pole_with_walls', backend='hip')
qp = ctrl.qp
x_max = load_fixture('cart_pole_with_walls')['x_max']
x0n, fix_h, par = bench.real_tree_frontier(ctrl, 4096, 0, x_max, spread=0.)
B = 4096


def alloc(n):
    return dict(obj=torch.empty(n, dtype=torch.float64, device=dev), dual_obj=torch.empty(n, dtype=torch.float64, device=dev),
                status=torch.empty(n, dtype=torch.int32, device=dev), iters=torch.empty(n, dtype=torch.int32, device=dev),
                primal=torch.empty(n, qp.n_primal, dtype=torch.float64, device=dev), dual=torch.empty(n, qp.n_dual, dtype=torch.float64, device=dev))


def timed(fix, warm, reps=5):
    out = alloc(fix.shape[0])
    for _ in range(2):
        qp.solve_batch_device(x0, fix, out, warm=warm)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); qp.solve_batch_device(x0, fix, out, warm=warm); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])), out


x0 = torch.from_numpy(np.ascontiguousarray(x0n[0])).to(dev)
fix = torch.from_numpy(fix_h).to(dev)
ms_cold, rec = timed(fix, None)
st, itf = rec['status'].cpu().numpy(), rec['iters'].cpu().numpy()
good = (par >= 0) & (st[np.maximum(par, 0)] == 0) & (((itf[np.maximum(par, 0)] >> 16) & 1) > 0)
widx = torch.from_numpy(np.where(good, par, -1).astype(np.int32)).to(dev)
ms_warm, out = timed(fix, (rec['primal'], rec['dual'], widx))
raw = out['iters'].cpu().numpy()
handed = ((raw >> 18) & 1) > 0
print('the frontier: cold %.3f ms, with the hand-down %.3f ms; handed a record %d, verified %d; iterations cold %.2f, with the hand-down %.2f'
      % (ms_cold, ms_warm, good.sum(), handed.sum(), (itf & 0xFFFF).mean(), (raw & 0xFFFF).mean()))
classes = {'verified hand-downs': np.flatnonzero(handed), 'hand-downs that did not verify': np.flatnonzero(good & ~handed),
           'nodes without a record to hand down': np.flatnonzero(~good)}
for name, sel in classes.items():
    idx = np.resize(sel, B)
    f = torch.from_numpy(np.ascontiguousarray(fix_h[idx])).to(dev)
    w = torch.from_numpy(np.where(good[idx], par[idx], -1).astype(np.int32)).to(dev)
    a, oa = timed(f, (rec['primal'], rec['dual'], w))
    c, oc = timed(f, None)
    ia, ic = oa['iters'].cpu().numpy() & 0xFFFF, oc['iters'].cpu().numpy() & 0xFFFF
    sa = oc['status'].cpu().numpy()
    print('%-40s %4d distinct, tiled to %d: handed %.3f ms (%.0f k QP/s, %.2f interior-point iterations), the same nodes cold %.3f ms (%.0f k QP/s, %.2f iterations; optimal %d, infeasible %d)'
          % (name, len(sel), B, a, B / a, ia.mean(), c, B / c, ic.mean(), (sa == 0).sum(), (sa == 1).sum()), flush=True)

# a launch of the fleet's mix: 85 % verified hand-downs, 15 % that do not verify
rng = np.random.RandomState(0)
ver, bad = np.flatnonzero(handed), np.flatnonzero(good & ~handed)
for n in (1900, 3800, 7600, 15200):
    for frac in (0.85, 1.0, 0.0):
        nv = int(round(frac * n))
        idx = np.concatenate((np.resize(ver, nv), np.resize(bad, n - nv)))
        rng.shuffle(idx)
        f = torch.from_numpy(np.ascontiguousarray(fix_h[idx])).to(dev)
        w = torch.from_numpy(par[idx].astype(np.int32)).to(dev)
        a, oa = timed(f, (rec['primal'], rec['dual'], w))
        raw2 = oa['iters'].cpu().numpy()
        its = raw2 & 0xFFFF
        print('%6d nodes, %3.0f %% of them verified hand-downs: %.3f ms (%.0f k QP/s); verified %d; iterations mean %.2f max %d'
              % (n, 100 * frac, a, n / a, (((raw2 >> 18) & 1) > 0).sum(), its.mean(), its.max()), flush=True)

# does the hand-out order matter?  HMPC_NO_ORDER=1: nodes are handed out in array order
if os.environ.get('HMPC_NO_ORDER'):
    for n in (1900, 3800):
        nv = int(round(0.85 * n))
        for name, idx in (('unverified first', np.concatenate((np.resize(bad, n - nv), np.resize(ver, nv)))),
                          ('unverified last', np.concatenate((np.resize(ver, nv), np.resize(bad, n - nv)))),
                          ('unverified first, longest first', None)):
            if idx is None:
                b2 = np.resize(bad, n - nv)
                b2 = b2[np.argsort(-(itf[b2] & 0xFFFF), kind='stable')]
                idx = np.concatenate((b2, np.resize(ver, nv)))
            f = torch.from_numpy(np.ascontiguousarray(fix_h[idx])).to(dev)
            w = torch.from_numpy(par[idx].astype(np.int32)).to(dev)
            a, oa = timed(f, (rec['primal'], rec['dual'], w))
            print('%6d nodes in array order, %s: %.3f ms' % (n, name, a), flush=True)
