"""the facet LPs of the offline terminal ingredients (mcais.py:44-184, controller.py:186-227).

Three solvers of the same batched call ``lp(A, c, b, relax) -> obj, x, z, status``:
  * HiGHS (tests/highs_lp.py) -- independent code, produced the committed terminal sets;
  * the LP oracle (oracle/dense_lp.c) -- CPU restatement of the kernel's algorithm, pinned here against HiGHS, the
    committed fixtures and the LP forms of the reference's known answers (test_bounded_qp.py:145-205,
    test_controller.py:40-59);
  * the HIP kernel through the C ABI (``hmpc_lp_solve_batch``) -- ``-m gpu``, same assertions plus parity with the oracle.
"""
import numpy as np
import pytest

import highs_lp
from helpers import load_fixture, lp_for
from warm_start_hmpc_amd import terminal_set as ts

VALUE_TOL = 1e-10   # optimal values: both sides end on the vertex (measured <= 3e-14)


def _ingredients(name):
    d = load_fixture(name)
    E = np.hstack((d['F'], d['G']))
    R = np.hstack((np.vstack((d['F'], d['F_T'].dot(d['A']))), np.vstack((d['G'], d['F_T'].dot(d['B'])))))
    return d, E, R


def _check_known_answers(lp):
    n = 15
    # feasible (test_bounded_qp.py:104-143 with a linear cost): max -1'x s.t. -x <= -lb  ->  x = lb, multipliers 1
    lb = np.arange(1., n + 1.)
    r = lp(-np.eye(n), -np.ones(n), -lb)
    assert r['status'][0] == 0
    np.testing.assert_allclose(r['x'][0], lb, atol=1e-12)
    np.testing.assert_allclose(r['z'][0], np.ones(n), atol=1e-12)
    np.testing.assert_allclose(r['obj'][0], -lb.sum(), rtol=1e-14)
    # empty set (test_bounded_qp.py:145-189): x <= a < 0 < b <= x; Farkas proof p = -q > 0 in the reference's signs,
    # i.e. equal nonnegative multipliers on both families with a'p - b'p < 0
    rng = np.random.RandomState(1)
    a, b = -rng.rand(n), rng.rand(n)
    A = np.vstack((np.eye(n), -np.eye(n)))
    r = lp(A, np.ones(n), np.concatenate((a, -b)))
    assert r['status'][0] == 1 and np.isnan(r['obj'][0])
    p, q = r['z'][0][:n], r['z'][0][n:]
    assert p.min() >= 0. and q.min() >= 0. and p.max() > 0.
    np.testing.assert_allclose(p, q, atol=1e-7 * p.max())
    assert a.dot(p) - b.dot(q) < 0.
    # unbounded (test_bounded_qp.py:191-205 without the quadratic part): max -x_n s.t. x <= 1
    c = np.zeros(n); c[-1] = -1.
    r = lp(np.eye(n), c, np.ones(n))
    assert r['status'][0] == 4
    ray = r['x'][0]
    assert c.dot(ray) > 0. and np.max(ray) <= 1e-7 * c.dot(ray)
    # a batch with per-LP right-hand sides, shared cost, and a cost of zero
    A = np.vstack((np.eye(2), -np.eye(2)))
    b = np.array([[1., 2., 0., 0.], [3., 1., 1., 1.]])
    r = lp(A, np.array([1., 1.]), b)
    np.testing.assert_allclose(r['obj'], [3., 4.], rtol=1e-13)
    r = lp(A, np.zeros(2), b[0])
    assert r['status'][0] == 0 and r['obj'][0] == 0.


def _check_against_highs(lp, name):
    d, E, R = _ingredients(name)
    # the LPs of the multiplier map
    got, ref = lp(E, R, d['h']), highs_lp.lp_solve_batch(E, R, d['h'])
    assert np.all(got['status'] == 0)
    np.testing.assert_allclose(got['obj'], ref['obj'], rtol=VALUE_TOL, atol=VALUE_TOL)
    assert got['z'].min() >= 0.
    np.testing.assert_allclose(got['z'].dot(E), R, atol=1e-12 * (1 + np.abs(R).max()))    # E'z = r
    np.testing.assert_allclose(got['z'].dot(d['h']), ref['obj'], rtol=VALUE_TOL, atol=VALUE_TOL)  # strong duality
    assert np.max(got['x'].dot(E.T) - d['h'][None, :]) <= 1e-10
    # the redundancy LPs on the committed terminal set: every facet of a minimal description moves by a full unit
    F_T, h_T = d['F_T'], d['h_T']
    m = F_T.shape[0]
    got, ref = lp(F_T, F_T, h_T, relax=np.arange(m)), highs_lp.lp_solve_batch(F_T, F_T, h_T, relax=np.arange(m))
    assert np.all(got['status'] == 0)
    np.testing.assert_allclose(got['obj'], ref['obj'], rtol=VALUE_TOL, atol=VALUE_TOL)
    assert np.all(got['obj'] - h_T >= 1e-7)


def _check_terminal_ingredients(lp):
    # mcais.py:44-144 from the closed-loop data of the fixture: same facets, same order, as the committed set
    d = load_fixture('cart_pole_with_walls')
    A_cl = d['A'] + d['B'][:, :1].dot(d['K'])
    D = d['F'] + d['G'][:, :1].dot(d['K'])
    F_T, h_T = ts.mcais(A_cl, D, d['h'], lp=lp)
    assert F_T.shape == d['F_T'].shape
    np.testing.assert_allclose(F_T, d['F_T'], atol=1e-13)
    np.testing.assert_allclose(h_T, d['h_T'], atol=1e-13)
    # a description with known redundant rows
    box = np.vstack((np.eye(3), -np.eye(3)))
    E = np.vstack((box, [[1., 1., 0.], [1., 0., 0.], [.5, .5, .5]]))
    f = np.concatenate((np.ones(6), [3., 2., 1.]))
    E_min, f_min = ts.remove_redundant_inequalities(E, f, lp=lp)
    np.testing.assert_array_equal(E_min, np.vstack((box, [[.5, .5, .5]])))
    np.testing.assert_array_equal(f_min, np.concatenate((np.ones(6), [1.])))
    with pytest.raises(ValueError):
        ts.mcais(1.1 * np.eye(2), np.vstack((np.eye(2), -np.eye(2))), np.ones(4), lp=lp)
    with pytest.raises(ValueError):
        ts.mcais(.5 * np.eye(2), np.vstack((np.eye(2), -np.eye(2))), np.array([1., 1., -1., 1.]), lp=lp)


def _check_multiplier_map(lp):
    for name in ('cart_pole_with_walls', 'cart_pole_one_wall'):
        d, E, R = _ingredients(name)
        F, G, h = d['F'], d['G'], d['h']
        M = ts.update_mu(F, G, h, R[:, :F.shape[1]], R[:, F.shape[1]:], lp=lp)
        assert M.shape == (h.size, R.shape[0]) and M.min() >= 0.
        np.testing.assert_allclose(E.T.dot(M), R.T, atol=1e-12 * (1 + np.abs(R).max()))
        # every column is optimal for the reference's LP (controller.py:205-226): same value as HiGHS on that LP
        best = highs_lp.lp_solve_batch(E, R, h)['obj']
        np.testing.assert_allclose(h.dot(M), best, rtol=VALUE_TOL, atol=VALUE_TOL)
        # the reference's known answer (test_controller.py:47-51): without a terminal set the map is the identity
        np.testing.assert_allclose(ts.update_mu(F, G, h, F, G, lp=lp), np.eye(h.size), atol=1e-12)
    # one wall: the optimum is unique there, the committed map (HiGHS, dual simplex) is reproduced
    d, E, R = _ingredients('cart_pole_one_wall')
    np.testing.assert_allclose(M, d['M'], atol=1e-12)
    # multipliers that are not unique, next to a row that is feasible but NOT optimal: r = (1, 1) is e_1 + e_2 (cost 2),
    # row 3 (cost 2: the optimal face has two vertices) or row 4 (the same direction at cost 3).  The least-weight pick
    # must stay on the optimal face -- row 4 gets nothing, however small the value the first launch left on it
    F = np.array([[1., 0.], [0., 1.], [1., 1.], [1., 1.]]); G = np.zeros((4, 1)); h = np.array([1., 1., 2., 3.])
    M = ts.update_mu(F, G, h, np.array([[1., 1.]]), np.zeros((1, 1)), lp=lp)
    np.testing.assert_allclose(F.T.dot(M[:, 0]), [1., 1.], atol=1e-12)
    assert abs(h.dot(M[:, 0]) - 2.) < 1e-12 and M[3, 0] == 0. and M.min() >= 0.
    # a row outside the conic hull: the reference's ValueError (controller.py:223-224)
    F = np.array([[1., 0.], [0., 1.]]); G = np.zeros((2, 1)); h = np.ones(2)
    with pytest.raises(ValueError):
        ts.update_mu(F, G, h, np.array([[-1., 0.]]), np.zeros((1, 1)), lp=lp)


def test_lp_oracle_known_answers():
    _check_known_answers(lp_for('oracle'))


@pytest.mark.parametrize('name', ['cart_pole_with_walls', 'cart_pole_one_wall'])
def test_lp_oracle_against_highs(name):
    _check_against_highs(lp_for('oracle'), name)


def test_terminal_ingredients_on_the_lp_oracle():
    _check_terminal_ingredients(lp_for('oracle'))
    _check_multiplier_map(lp_for('oracle'))


def test_lp_oracle_random_batches_against_highs():
    rng = np.random.RandomState(7)
    for n, m, B in ((2, 9, 16), (5, 40, 24), (12, 70, 12), (34, 150, 4)):
        A = rng.randn(m, n)
        x_in = rng.randn(n)
        b = A.dot(x_in) + rng.rand(B, m) + .1          # x_in strictly inside every set
        A = np.vstack((A, np.eye(n), -np.eye(n)))      # boxed: every LP bounded
        b = np.hstack((b, np.full((B, 2 * n), 10.)))
        c = rng.randn(B, n)
        got, ref = lp_for('oracle')(A, c, b), highs_lp.lp_solve_batch(A, c, b)
        assert np.all(got['status'] == 0)
        np.testing.assert_allclose(got['obj'], ref['obj'], rtol=1e-9, atol=1e-9)
        assert np.max(np.einsum('rj,kj->kr', A, got['x']) - b) <= 1e-9
        np.testing.assert_allclose(got['z'].dot(A), c, atol=1e-9)


# ---- the HIP kernel through the C ABI ----

@pytest.mark.gpu
def test_lp_kernel_known_answers():
    _check_known_answers(lp_for('hip'))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cart_pole_with_walls', 'cart_pole_one_wall'])
def test_lp_kernel_against_highs_and_oracle(name):
    _check_against_highs(lp_for('hip'), name)
    d, E, R = _ingredients(name)
    hip, orc = lp_for('hip')(E, R, d['h']), lp_for('oracle')(E, R, d['h'])
    np.testing.assert_array_equal(hip['status'], orc['status'])
    np.testing.assert_allclose(hip['obj'], orc['obj'], rtol=1e-12, atol=1e-12)
    # where the optimum is a face rather than a vertex both return a point of its relative interior, which is not
    # pinned beyond the accuracy of the iterate it is read from: compared loosely; values above, exact
    np.testing.assert_allclose(hip['x'], orc['x'], atol=1e-3)
    np.testing.assert_allclose(hip['z'], orc['z'], atol=1e-3 * (1 + np.abs(orc['z']).max()))
    assert np.max(np.abs(hip['iters'] - orc['iters'])) <= 1


@pytest.mark.gpu
def test_terminal_ingredients_on_the_lp_kernel():
    _check_terminal_ingredients(lp_for('hip'))
    _check_multiplier_map(lp_for('hip'))


@pytest.mark.gpu
def test_lp_kernel_random_batches_and_limits():
    from warm_start_hmpc_amd.qp_backend import lp_solve_batch
    rng = np.random.RandomState(7)
    for n, m, B in ((2, 9, 16), (5, 40, 700), (12, 70, 12), (34, 150, 4), (64, 300, 3), (3, 1700, 2)):
        A = rng.randn(m, n)
        x_in = rng.randn(n)
        b = A.dot(x_in) + rng.rand(B, m) + .1
        A = np.vstack((A, np.eye(n), -np.eye(n)))
        b = np.hstack((b, np.full((B, 2 * n), 10.)))
        c = rng.randn(B, n)
        got, orc = lp_solve_batch(A, c, b), lp_for('oracle')(A, c, b, threads=8)
        assert np.all(got['status'] == 0)
        np.testing.assert_allclose(got['obj'], orc['obj'], rtol=1e-9, atol=1e-9)
        assert np.max(np.einsum('rj,kj->kr', A, got['x']) - b) <= 1e-9
        np.testing.assert_allclose(got['z'].dot(A), c, atol=1e-9)
    with pytest.raises(RuntimeError):      # n beyond one wavefront
        lp_solve_batch(np.ones((70, 65)), np.ones(65), np.ones(70))
    with pytest.raises(RuntimeError):      # row vectors beyond one CU's LDS
        lp_solve_batch(np.ones((4000, 2)), np.ones(2), np.ones(4000))
    with pytest.raises(ValueError):
        lp_solve_batch(np.ones((4, 2)), np.ones(3), np.ones(4))


@pytest.mark.gpu
def test_controller_builds_its_multiplier_map_on_the_gpu():
    # the product path: no lp= argument, the HIP backend's LP kernel computes the map at construction
    from warm_start_hmpc_amd.mld_system import MLDSystem
    from warm_start_hmpc_amd.controller import HybridModelPredictiveController
    d = load_fixture('cart_pole_with_walls')
    mld = MLDSystem([d['A'], d['B']], [d['F'], d['G'], d['h']], int(d['nub']))
    ctrl = HybridModelPredictiveController(mld, 10, [d['Q'], d['R'], d['Q_T']], [d['F_T'], d['h_T']])
    assert 'mu' in dict.keys(ctrl._update)
    ref = ts.update_mu(d['F'], d['G'], d['h'], ctrl.F_Tm1, ctrl.G_Tm1, lp=lp_for('oracle'))
    np.testing.assert_allclose(ctrl._update['mu'], ref, atol=1e-9)
