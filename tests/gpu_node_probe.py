"""Diagnostic: one frontier through the HIP path and the oracle, per-node comparison with the dense active-set solve.
usage: gpu_node_probe.py [fixture T file.npz]   (npz with x0 [B, nx] or [nx], fix [B, T nub]); default: the frontier of
test_every_kernel_instantiation[cart_pole_one_wall-40]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier
import dense_qp
if len(sys.argv) > 3:
    name, T = sys.argv[1], int(sys.argv[2])
    d = np.load(sys.argv[3])
    X0, fix = d['x0'], d['fix']
else:
    name, T = 'cart_pole_one_wall', 40
    X0 = np.array([0., 0., 1., 0.])
    fix = random_prefix_frontier(T, 2, 40, p_one=0.1, seed0=9000)
    fix[0, :] = -1
hip = make_controller(name, T=T, backend='hip')
orc = make_controller(name, T=T, backend='oracle', threads=8)
dq = dense_qp.dense_qp(hip)
n = (T + 1) * 4
for waves in (('1', '2', '4') if len(sys.argv) <= 3 else ('',)):
    if waves:
        os.environ['HMPC_WAVES'] = waves
    a = hip.qp.solve_batch(X0, fix)
    b = orc.qp.solve_batch(X0, fix)
    opt = np.flatnonzero((a['status'] == 0) & (b['status'] == 0))
    dev = np.abs(a['primal'][opt][:, :n] - b['primal'][opt][:, :n]).max(axis=1)
    show = opt if opt.size <= 8 else opt[(a['polished'][opt] != b['polished'][opt]) | (dev > 5e-8)]
    print('waves %s: %d nodes, status mismatches %d, optimal %d, listed %d' % (waves, len(fix), int((a['status'] != b['status']).sum()), opt.size, show.size))
    for i in show:
        x0 = X0 if X0.ndim == 1 else X0[i]
        wa, _ = dense_qp.active_set_primal(hip, dq, x0, fix[i], a['dual'][i])
        wb, _ = dense_qp.active_set_primal(hip, dq, x0, fix[i], b['dual'][i])
        print('  node %4d: obj hip %.10f oracle %.10f | polished %d/%d iters %d/%d | hip-oracle %.2e hip-dense(hip set) %.2e oracle-dense(oracle set) %.2e | active rows %d/%d'
              % (i, a['obj'][i], b['obj'][i], a['polished'][i], b['polished'][i], a['iters'][i] & 0xFFFF, b['iters'][i] & 0xFFFF,
                 np.abs(a['primal'][i][:n] - b['primal'][i][:n]).max(), np.abs(a['primal'][i][:n] - wa[:n]).max(), np.abs(b['primal'][i][:n] - wb[:n]).max(),
                 int((a['dual'][i][n:] > 0).sum()), int((b['dual'][i][n:] > 0).sum())), flush=True)
