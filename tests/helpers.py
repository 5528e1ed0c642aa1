"""Shared test helpers: controllers built from the committed fixtures, synthetic frontiers."""
import os

import numpy as np

from warm_start_hmpc_amd.mld_system import MLDSystem
from warm_start_hmpc_amd.controller import HybridModelPredictiveController

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class _NoBackend(object):
    """Placeholder so that a controller can be built before its backend exists."""


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def make_controller(name='cart_pole_with_walls', T=None, terminal=True, backend='oracle', **opts):
    """backend: 'oracle' (CPU, test infrastructure), 'hip' (the product path) or an object."""
    d = load_fixture(name)
    mld = MLDSystem([d['A'], d['B']], [d['F'], d['G'], d['h']], int(d['nub']))
    T = int(d['T']) if T is None else T
    term = [d['F_T'], d['h_T']] if terminal else None
    ctrl = HybridModelPredictiveController(mld, T, [d['Q'], d['R'], d['Q_T']], term, backend=_NoBackend())
    if backend == 'oracle':
        from oracle.oracle_qp import OracleBatchedQP
        ctrl.qp = OracleBatchedQP(ctrl.problem_data(), **opts)
    elif backend == 'hip':
        from warm_start_hmpc_amd.qp_backend import HipBatchedQP
        ctrl.qp = HipBatchedQP(ctrl.problem_data(), **opts)
    else:
        ctrl.qp = backend
    return ctrl


def random_prefix_frontier(T, nub, count, p_one=0.5, seed0=1000):
    """Synthetic frontier of SURVEY.md 8(d) C2: node k (seed seed0+k) fixes the first d binaries
    in (t, i) order, d ~ U{0..T*nub}, to Bernoulli(p_one) values; the rest are free (-1)."""
    fix = np.full((count, T * nub), -1, dtype=np.int8)
    for k in range(count):
        rng = np.random.default_rng(seed0 + k)
        d = int(rng.integers(0, T * nub + 1))
        fix[k, :d] = (rng.random(d) < p_one).astype(np.int8)
    return fix


def random_mld(nx=20, nuc=6, nub=8, seed=0):
    """Random MLD of SURVEY.md 8(d) C4 (a build decision, BASELINE.json leaves the rows open):
    stable A, box rows on x and uc, four big-M rows per binary."""
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((nx, nx))
    A = 0.95 * W / np.max(np.abs(np.linalg.eigvals(W)))
    nu = nuc + nub
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    B[:, nuc:] *= 0.5
    rows_F, rows_G, h = [], [], []
    for i in range(nx):
        for s in (1., -1.):
            f = np.zeros(nx); f[i] = s
            rows_F.append(f); rows_G.append(np.zeros(nu)); h.append(5.)
    for i in range(nuc):
        for s in (1., -1.):
            g = np.zeros(nu); g[i] = s
            rows_F.append(np.zeros(nx)); rows_G.append(g); h.append(1.)
    for j in range(nub):
        c = rng.standard_normal(nx); c /= np.linalg.norm(c)
        big = 5. * np.sum(np.abs(c))
        # b_j = 1 <=> c'x >= 0  (big-M), and the binary gates continuous input j % nuc
        g = np.zeros(nu); g[nuc + j] = -big
        rows_F.append(c.copy()); rows_G.append(g.copy()); h.append(0.)            #  c'x <= M b
        g = np.zeros(nu); g[nuc + j] = big
        rows_F.append(-c); rows_G.append(g.copy()); h.append(big)                 # -c'x <= M (1 - b)
        g = np.zeros(nu); g[j % nuc] = 1.; g[nuc + j] = -1.
        rows_F.append(np.zeros(nx)); rows_G.append(g.copy()); h.append(0.5)       #  uc <= .5 + b
        g = np.zeros(nu); g[j % nuc] = -1.; g[nuc + j] = -1.
        rows_F.append(np.zeros(nx)); rows_G.append(g.copy()); h.append(0.5)       # -uc <= .5 + b
    F, G, h = np.array(rows_F), np.array(rows_G), np.array(h)
    mld = MLDSystem([A, B], [F, G, h], nub)
    Q = np.eye(nx)
    R = np.hstack((np.eye(nuc), np.zeros((nuc, nub))))
    x0 = rng.uniform(-1, 1, nx) * 0.5
    return mld, [Q, R, Q.copy()], x0


def record_close(a, b, tol=1e-5):
    """Relative comparison of two result dicts from solve_batch (status exact, floats to tol)."""
    assert np.array_equal(a['status'], b['status']), (a['status'], b['status'])
    fin = np.isfinite(a['obj'])
    assert np.array_equal(fin, np.isfinite(b['obj']))
    np.testing.assert_allclose(a['obj'][fin], b['obj'][fin], rtol=tol, atol=tol * 1e-2)
