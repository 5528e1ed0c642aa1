"""Runs every kernel instantiation through the HMPC_CHECK build (`make -C warm-start-hybrid-mpc_amd/csrc check`): bounds-checked
row handles and gathered indices, NaN-poisoned per-node LDS vectors and row slots.  A read of something never written, or an
index out of range, fails here even where the shipped build happens to get away with it (diagnostic, hand-run on the GPU box)."""
import sys, os
os.environ['HMPC_LIBRARY_NAME'] = 'libhmpc_check.so'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier, random_mld, _NoBackend
X0 = np.array([0., 0., 1., 0.])
bad = 0


def run(tag, hip, orc, x0, fix, T, nx):
    global bad
    try:
        a = hip.solve_batch(x0, fix)
    except RuntimeError as e:
        print('FAIL', tag, e)
        bad += 1
        return
    b = orc.solve_batch(x0, fix)
    ok = np.array_equal(a['status'], b['status'])
    fin = (a['status'] == 0) & (b['status'] == 0)
    dev = np.abs(a['primal'][fin][:, :(T + 1) * nx] - b['primal'][fin][:, :(T + 1) * nx]).max() if fin.any() else 0.
    nan = int(np.isnan(a['primal'][a['status'] == 0]).sum() + np.isnan(a['dual']).sum())
    good = ok and dev < 1e-5 and nan == 0
    bad += not good
    print('ok  ' if good else 'FAIL', tag, 'nodes', len(fix), 'optimal', int(fin.sum()), 'status equal', ok, 'x dev %.1e' % dev, 'NaNs', nan, flush=True)


for fixture, T, nub in (('cart_pole_with_walls', 20, 4), ('cart_pole_with_walls', 40, 4), ('cart_pole_one_wall', 40, 2), ('cart_pole_with_walls', 10, 4)):
    for terminal in (True, False):
        hip = make_controller(fixture, T=T, terminal=terminal, backend='hip')
        orc = make_controller(fixture, T=T, terminal=terminal, backend='oracle', threads=16)
        fix = random_prefix_frontier(T, nub, 96, p_one=0.1, seed0=9000 + T)
        fix[0, :] = -1
        for w in ('1', '2', '4'):
            os.environ['HMPC_WAVES'] = w
            run('%s T=%d terminal=%s waves=%s' % (fixture, T, terminal, w), hip.qp, orc.qp, X0 if T != 10 or not terminal else X0 * 0.5, fix, T, 4)
        del os.environ['HMPC_WAVES']
        big = random_prefix_frontier(T, nub, 2400 if T == 20 else 600, p_one=0.1, seed0=500 + T)     # dynamic hand-out + ordering
        run('%s T=%d terminal=%s large batch' % (fixture, T, terminal), hip.qp, orc.qp, X0 if T != 10 or not terminal else X0 * 0.5, big, T, 4)
for env in ('HMPC_FORCE_GENERIC', 'HMPC_FORCE_BIG'):
    os.environ[env] = '1'
    hip = make_controller('cart_pole_with_walls', backend='hip')
    del os.environ[env]
    orc = make_controller('cart_pole_with_walls', backend='oracle', threads=16)
    fix = random_prefix_frontier(20, 4, 96, p_one=0.2, seed0=11000)
    for w in ('1', '4'):
        os.environ['HMPC_WAVES'] = w
        run('%s waves=%s' % (env, w), hip.qp, orc.qp, X0, fix, 20, 4)
    del os.environ['HMPC_WAVES']
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
mld, objective, x0 = random_mld(nx=6, nuc=2, nub=3, seed=3)
ctrl = HybridModelPredictiveController(mld, 8, objective, None, backend=_NoBackend())
fix = random_prefix_frontier(8, 3, 128, p_one=0.3)
run('random MLD nx=6', HipBatchedQP(ctrl.problem_data()), OracleBatchedQP(ctrl.problem_data(), threads=16), x0, fix, 8, 6)
# BASELINE configs[4]: the streaming form with the split stage rows, the panel factorisation and the padded sweeps
# (a dive frontier: mostly optimal nodes; the x deviation is of POLISHED records only -- see test_gpu_parity.py on the others)
mld, objective, x0 = random_mld()
ctrl = HybridModelPredictiveController(mld, 30, objective, None, backend=_NoBackend())
hip4, orc4 = HipBatchedQP(ctrl.problem_data()), OracleBatchedQP(ctrl.problem_data(), threads=16)
Cj = np.array([mld.F[52 + 4 * j] for j in range(8)])
leaf = np.full((1, 240), -1, np.int8)
for t in range(30):
    r = orc4.solve_batch(x0, leaf)
    leaf[0, t * 8:(t + 1) * 8] = (r['primal'][0][:31 * 20].reshape(31, 20)[t] @ Cj.T >= 0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import dive_frontier
f4 = dive_frontier(leaf[0], 320, 0)
a, b = hip4.solve_batch(x0, f4), orc4.solve_batch(x0, f4)
both = (a['status'] == 0) & (a['polished'] > 0) & (b['polished'] > 0)
dev = np.abs(a['primal'][both][:, :31 * 20] - b['primal'][both][:, :31 * 20]).max()
nan = int(np.isnan(a['primal'][a['status'] == 0]).sum() + np.isnan(a['dual']).sum())
good = np.array_equal(a['status'], b['status']) and dev < 1e-5 and nan == 0
bad += not good
print('ok  ' if good else 'FAIL', 'random MLD nx=20 nu=6+8 N=30 (streaming form) nodes', len(f4), 'optimal', int((a['status'] == 0).sum()), 'polished on both sides', int(both.sum()),
      'x dev %.1e' % dev, 'NaNs', nan, flush=True)
print('CHECK BUILD:', 'all instantiations clean' if bad == 0 else '%d FAILURES' % bad)
