"""The Gurobi leg (tests/gurobi_reference.py).  Three kinds of test:
  * always: the assembly of the reference's QP and of the records from solver vectors, on a stand-in for the solver --
    the only part of the leg this image can execute (no gurobipy);
  * where ``import gurobipy`` succeeds: this repository's CPU oracle against live Gurobi;
  * where tests/golden/gurobi_golden.npz exists (made by a licence holder with tests/golden/make_gurobi_golden.py): the
    oracle (CPU) and the HIP kernel (``-m gpu``) against Gurobi's recorded vectors -- BASELINE.json's tolerance: statuses
    and binary assignments exact, continuous trajectories within 1e-5 relative."""
import os

import numpy as np
import pytest

from helpers import make_controller, load_fixture, random_prefix_frontier
from dense_qp import determined_inputs
import gurobi_reference

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gurobi_golden.npz')
RTOL = 1e-5


def test_record_assembly_on_a_stand_in_solver():
    # the oracle's records play the solver: turned into what Gurobi would hand back (X, Pi = -multipliers, FarkasDual =
    # +multipliers, ObjVal), NodeQP must assemble the same records again -- row order, signs, rho / sigma, the Farkas
    # objective over ALL constraints -- and its dense matrices must certify them (stationarity, primal rows)
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=4)
    nq = gurobi_reference.NodeQP(ctrl.problem_data())
    x0 = np.array([0., 0., .5, 0.])
    fix = random_prefix_frontier(10, 4, 48, p_one=0.15, seed0=4000)
    fix[0, :] = -1
    res = ctrl.qp.solve_batch(x0, fix)
    assert (res['status'] == 0).sum() >= 5 and (res['status'] == 1).sum() >= 5
    assert nq.n_dual == res['dual'].shape[1] and nq.n == res['primal'].shape[1]
    o = 11 * 4
    m = nq.n_mu + 2 * 10 * 4
    for b in range(len(fix)):
        beq, bin_ = nq.rhs(x0, fix[b])
        y, z = res['dual'][b][:o], res['dual'][b][o:o + m]
        if res['status'][b] == 0:
            w = res['primal'][b]
            obj, dobj, status, prim, dual = nq.record(True, w, -y, -z, res['obj'][b], beq, bin_)
            np.testing.assert_allclose(dual, res['dual'][b], rtol=1e-12, atol=1e-12)        # (rho, sigma recomputed from w)
            assert status == 0 and obj == res['obj'][b]
            # the dense statement certifies the record: stationarity 2 Hq w + Eq'y + In'z = 0, rows met, z >= 0, z (b - In w) = 0
            assert np.max(np.abs(2 * nq.Hq @ w + nq.Eq.T @ y + nq.In.T @ z)) < 1e-7 * (1 + np.max(np.abs(z)))
            assert np.max(np.abs(nq.Eq @ w - beq)) < 1e-9 and np.max(nq.In @ w - bin_) < 1e-8 and z.min() >= 0
            assert abs(w @ nq.Hq @ w - res['obj'][b]) < 1e-9 * (1 + res['obj'][b])
        else:
            obj, dobj, status, prim, dual = nq.record(False, None, y, z, np.inf, beq, bin_)
            assert status == 1 and np.isinf(obj) and np.all(np.isnan(prim))
            np.testing.assert_allclose(dual, res['dual'][b], rtol=0, atol=0)
            assert dobj > 0 and abs(dobj - res['dual_obj'][b]) <= 1e-9 * (1 + abs(dobj))     # - sum RHS * FarkasDual
            assert np.max(np.abs(nq.Eq.T @ y + nq.In.T @ z)) < 1e-5 * dobj and z.min() >= 0       # a proof: E'y + C'z = 0, z >= 0


def _close(ctrl, T, fix, a, b):
    """a against b (Gurobi): statuses exact; objectives; states and determined inputs within RTOL relative; all inputs of
    fully fixed nodes; proofs by their sign."""
    assert np.array_equal(a['status'], b['status'])
    fin = b['status'] == 0
    np.testing.assert_allclose(a['obj'][fin], b['obj'][fin], rtol=1e-6, atol=1e-9)
    nx, nu = ctrl.mld.nx, ctrl.mld.nu
    pa, pb = a['primal'][fin], b['primal'][fin]

    def rel(u, v):
        return (np.max(np.abs(u - v), axis=1) / np.maximum(1e-2, np.max(np.abs(v), axis=1))).max(initial=0)
    assert rel(pa[:, :(T + 1) * nx], pb[:, :(T + 1) * nx]) < RTOL
    ua, ub = pa[:, (T + 1) * nx:].reshape(-1, T, nu), pb[:, (T + 1) * nx:].reshape(-1, T, nu)
    for j in determined_inputs(ctrl):
        assert rel(ua[:, :, j], ub[:, :, j]) < RTOL
    full = (np.asarray(fix)[fin] >= 0).all(axis=1)
    if full.any():
        assert rel(ua[full].reshape(full.sum(), -1), ub[full].reshape(full.sum(), -1)) < RTOL
    inf = b['status'] == 1
    assert np.all(a['dual_obj'][inf] > 0) and np.all(b['dual_obj'][inf] > 0)


@pytest.mark.skipif(not gurobi_reference.available(), reason='gurobipy is not importable here (no licence in this image): the live Gurobi leg is optional')
def test_oracle_against_live_gurobi():
    g = load_fixture('qp_golden')
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=4)
    grb = gurobi_reference.GurobiBatchedQP(ctrl.problem_data(), gurobi_params={'FeasibilityTol': 1e-9, 'OptimalityTol': 1e-9})
    fix, x0 = g['n10_fix'], g['n10_x0']
    _close(ctrl, 10, fix, ctrl.qp.solve_batch(x0, fix), grb.solve_batch(x0, fix))


def _sets():
    if not os.path.exists(GOLDEN):
        return []
    g = np.load(GOLDEN)
    return sorted({k[:-4] for k in g.files if k.endswith('_fix')})


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason='tests/golden/gurobi_golden.npz has not been generated (needs a Gurobi licence: tests/golden/make_gurobi_golden.py)')
@pytest.mark.parametrize('backend', ['oracle', pytest.param('hip', marks=pytest.mark.gpu)])
def test_against_gurobi_golden_vectors(backend):
    g = np.load(GOLDEN)
    for name in _sets():
        T = int(g[name + '_T'])
        ctrl = make_controller(str(g[name + '_fixture']), T=T, terminal=bool(g[name + '_terminal']), backend=backend)
        rec = {k: g['%s_%s' % (name, k)] for k in ('status', 'obj', 'dual_obj', 'primal', 'dual')}
        _close(ctrl, T, g[name + '_fix'], ctrl.qp.solve_batch(g[name + '_x0'], g[name + '_fix']), rec)
