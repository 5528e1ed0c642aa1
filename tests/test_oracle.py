"""Pins the CPU oracle (oracle/hsde_qp.c): the reference's known-answer tests for the QP
boundary, its KKT / cover / lower-bound properties, its published solve counts, and the
committed golden vectors.  No GPU needed."""
import numpy as np
import pytest

from helpers import make_controller, load_fixture, _NoBackend
from kkt_checks import check_solution, is_disjoint_cover, dual_residuals, dual_objective
from warm_start_hmpc_amd.mld_system import MLDSystem
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from oracle.oracle_qp import OracleBatchedQP

X0 = np.array([0., 0., 1., 0.])


def _generic_qp(n, G, h, R):
    """min |R u|^2 s.t. G u <= h through the boundary: two identical stages with a dummy state
    (nx = 1, A = 0, B = 0).  Talks to the backend directly (no controller: the warm-start LPs of
    its constructor have no meaning for an arbitrary, possibly infeasible, constraint set)."""
    from warm_start_hmpc_amd.subproblem_solution import RecordLayout, SubproblemSolution
    m = G.shape[0]
    prob = dict(nx=1, nu=n, nub=0, T=2, A=np.zeros((1, 1)), B=np.zeros((1, n)), F=np.zeros((m, 1)), G=G, h=h,
                F_Tm1=np.zeros((m, 1)), G_Tm1=G, h_Tm1=h, Q=np.zeros((1, 1)), R=R, Q_T=np.zeros((1, 1)))
    res = OracleBatchedQP(prob).solve_batch(np.zeros(1), np.zeros((1, 0), dtype=np.int8))
    layout = RecordLayout(1, n, 0, 2, m, m, 1, R.shape[0], 1)
    return SubproblemSolution.from_rows(layout, np.zeros(0, dtype=np.int8), res['obj'][0], res['dual_obj'][0],
                                        res['status'][0], res['primal'][0], res['dual'][0])


def test_known_answer_feasible():
    # reference: test_bounded_qp.py:104-143 -- min 1/2 |x|^2 s.t. x >= 1, n = 15:
    # x* = 1, multiplier magnitude 1, objective n/2 (primal = dual)
    n = 15
    sol = _generic_qp(n, -np.eye(n), -np.ones(n), np.eye(n) / np.sqrt(2.))
    # two stages carry the same constraint and cost: each stage reproduces the known answer
    for t in range(2):
        np.testing.assert_allclose(sol.primal.variables['uc'][t], np.ones(n), atol=1e-7)
        np.testing.assert_allclose(sol.dual.variables['mu'][t], np.ones(n), atol=1e-6)
    assert abs(sol.primal.objective - n) < 1e-6          # 2 stages x n/2
    assert abs(sol.dual.objective - n) < 1e-6
    assert np.all(sol.dual.variables['mu'][0] > 0)        # "<=" rows: nonnegative multipliers


def test_known_answer_infeasible():
    # reference: test_bounded_qp.py:145-189 -- x <= a < 0, x >= b > 0: primal None, objective inf,
    # Farkas proof p >= 0 on "x <= a", q on "-x <= -b" with p == q (their q is -ours), objective -a'p + b'q > 0
    np.random.seed(1)
    n = 15
    a, b = -np.random.rand(n), np.random.rand(n)
    G = np.vstack((np.eye(n), -np.eye(n)))
    h = np.concatenate((a, -b))
    sol = _generic_qp(n, G, h, np.eye(n))
    assert sol.primal.variables['uc'][0] is None and np.isinf(sol.primal.objective)
    proof = 0.
    for t in range(2):
        p, q = sol.dual.variables['mu'][t][:n], sol.dual.variables['mu'][t][n:]
        assert np.min(p) >= 0 and np.min(q) >= 0
        np.testing.assert_allclose(p, q, atol=1e-9 * (1 + np.max(p)))
        proof += -a.dot(p) + b.dot(q)
    assert proof > 0
    assert abs(sol.dual.objective - proof) <= 1e-9 * (1 + proof)
    assert all(np.all(v == 0) for v in sol.dual.variables['rho'] + sol.dual.variables['sigma'])


@pytest.fixture(scope='module')
def solved():
    ctrl = make_controller('cart_pole_with_walls', backend='oracle')
    sol, leaves, solves, _ = ctrl.feedforward(X0, printing_period=None)
    return ctrl, sol, leaves, solves


def test_cold_start_matches_published_counts(solved):
    ctrl, sol, leaves, solves = solved
    # reference data: notebooks/cart_pole_with_walls/data/solve_log_sd_0.000.log:8-57 -- 158..161 solves
    assert 157 <= solves <= 162
    assert len(leaves) == 81
    ub = np.array(sol.variables['ub'])
    assert np.all(ub[:, :2] == 0)                               # left wall never touched
    assert list(np.flatnonzero(ub[:, 2])) == list(range(10, 17))  # el_r
    assert list(np.flatnonzero(ub[:, 3])) == list(range(8, 15))   # dam_r
    assert abs(sol.objective - 0.069257) < 2e-6


def test_leaves_certify_themselves(solved):
    # reference: test_controller.py:84-120 (KKT residuals; leaf bound <= leaf optimum; cover)
    ctrl, sol, leaves, _ = solved
    for leaf in leaves:
        s = ctrl._solve_subproblem(leaf.identifier, X0)[0]
        check_solution(ctrl, s, leaf.identifier, X0, tol=1e-6)
        assert s.primal.objective >= leaf.lb - 1e-7
        assert s.primal.objective >= sol.objective - 1e-7
    assert is_disjoint_cover(ctrl, leaves)


def test_warm_start_properties(solved):
    # reference: test_controller.py:122-170
    ctrl, sol, leaves, _ = solved
    np.random.seed(1)
    uc0, ub0 = sol.variables['uc'][0], sol.variables['ub'][0]
    e0 = np.random.randn(ctrl.mld.nx) * .001
    x1 = ctrl.mld.A.dot(X0) + ctrl.mld.B.dot(np.concatenate((uc0, ub0))) + e0
    ws = ctrl.construct_warm_start(leaves, X0, uc0, ub0, e0)[0]
    # published cover size (solve_log_sd_0.000.log:8): 77
    assert len(ws) == 77
    assert is_disjoint_cover(ctrl, ws)
    kept = 0
    for node in ws:
        s = ctrl._solve_subproblem(node.identifier, x1)[0]
        assert s.primal.objective >= node.lb - 1e-7          # implied bounds are valid
        if node.extra.dual is not None:
            zero, nonneg = dual_residuals(ctrl, node.extra.dual.variables)
            assert np.max(np.abs(zero)) < 1e-5 * (1 + np.max(np.abs(np.concatenate(node.extra.dual.variables['mu']))))
            assert np.min(nonneg) >= -1e-9
            obj = max(0., dual_objective(ctrl, node.extra.dual.variables, node.identifier, x1))
            if np.isinf(node.lb):
                assert obj > 0.
                kept += 1
            else:
                assert abs(obj - node.lb) < 1e-6
    assert kept >= 70   # the Farkas proofs survive the shift (73-75 of 77)
    cold = ctrl.feedforward(x1, printing_period=None)
    warm = ctrl.feedforward(x1, printing_period=None, warm_start=ws)
    assert warm[0].objective == cold[0].objective             # test_controller.py:165-170 (assertEqual)
    assert warm[2] <= 25 and cold[2] >= 150                   # published: 10-17 warm, 158-161 cold


def test_other_horizons_known_answers():
    # SURVEY.md Appendix E
    c10 = make_controller('cart_pole_with_walls', T=10, backend='oracle')
    sol, leaves, solves, _ = c10.feedforward(X0, printing_period=None)
    assert sol is None and all(np.isinf(l.lb) for l in leaves)          # N=10 from x0=[0,0,1,0] is infeasible
    sol, leaves, solves, _ = c10.feedforward(np.array([0., 0., .5, 0.]), printing_period=None)
    assert abs(sol.objective - 0.0995300) < 1e-6 and np.all(np.array(sol.variables['ub']) == 0)
    assert abs(solves - 80) <= 2 and abs(len(leaves) - 41) <= 1     # (counts wobble with the last digits of the multipliers)


def test_golden_vectors():
    # the golden file comes from this oracle run tighter (tol 1e-10, polish from a converged iterate) and accepted
    # against an independent dense solve; the default configuration -- the product's -- must reproduce it
    g = load_fixture('qp_golden')
    for name, fixture in [('n20', 'cart_pole_with_walls'), ('n20dive', 'cart_pole_with_walls'), ('n20tree', 'cart_pole_with_walls'),
                          ('n20x0', 'cart_pole_with_walls'), ('n10', 'cart_pole_with_walls'), ('onewall', 'cart_pole_one_wall')]:
        T = int(g[name + '_T'])
        ctrl = make_controller(fixture, T=T, terminal=bool(g[name + '_terminal']), backend='oracle', threads=8)
        res = ctrl.qp.solve_batch(g[name + '_x0'], g[name + '_fix'])
        assert np.array_equal(res['status'], g[name + '_status'])
        fin = res['status'] == 0
        assert np.all(res['polished'][fin] > 0)
        np.testing.assert_allclose(res['obj'][fin], g[name + '_obj'][fin], rtol=1e-8, atol=1e-12)
        nx = ctrl.mld.nx
        np.testing.assert_allclose(res['primal'][fin][:, :(T + 1) * nx], g[name + '_x'][fin], rtol=1e-6, atol=1e-7)


def test_trajectories_against_dense_active_set_solve():
    # independent of the Riccati / interior-point / polish code: the node QP as dense matrices, the active set read
    # from the record's multipliers, numpy SVD (tests/dense_qp.py).  Nodes of a real tree (the ill-conditioned ones:
    # without the polish the interior-point iterate is off by up to 4e-5 there) and random prefixes.
    from dense_qp import dense_qp, active_set_primal
    g = load_fixture('qp_golden')
    ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
    raw = make_controller('cart_pole_with_walls', backend='oracle', threads=8, polish=False)
    dq = dense_qp(ctrl)
    fix = g['n20tree_fix'][::3]
    res, unp = ctrl.qp.solve_batch(X0, fix), raw.qp.solve_batch(X0, fix)
    worst = worst_raw = 0.
    for b in np.flatnonzero(res['status'] == 0):
        w, resid = active_set_primal(ctrl, dq, X0, fix[b], res['dual'][b])
        scale = max(1e-2, np.max(np.abs(w[:84])))
        worst = max(worst, np.max(np.abs(w[:84] - res['primal'][b][:84])) / scale)
        worst_raw = max(worst_raw, np.max(np.abs(w[:84] - unp['primal'][b][:84])) / scale)
        assert resid < 1e-10
    assert worst < 1e-7, worst
    assert worst_raw < 1e-3          # (what the polish is for: measured 4e-5 on this set)


def test_root_relaxations_against_an_independent_qp_solver():
    # scipy's trust-constr (Byrd-Hribar-Nocedal interior point; no code, factorisation or active-set logic in common with
    # the oracle or the kernel) on the dense statement of the node QP (tests/dense_qp.py): the root relaxation of the
    # N=10 case from two initial states.  (~7 s each; on nodes with fixed binaries its equality-constrained SQP meets a
    # singular Jacobian and takes minutes to reach 1e-6 -- those are pinned by the dense active-set solve instead.)
    from scipy.optimize import minimize, LinearConstraint, Bounds
    from dense_qp import dense_qp
    T, nx, nu, nub = 10, 4, 7, 4
    ctrl = make_controller('cart_pole_with_walls', T=T, backend='oracle')
    H, E, C, h = dense_qp(ctrl)
    n = H.shape[0]
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    for t in range(T):
        for b in range(nub):
            lo[(T + 1) * nx + t * nu + (nu - nub) + b], hi[(T + 1) * nx + t * nu + (nu - nub) + b] = 0., 1.
    fix = np.full((1, T * nub), -1, np.int8)
    for x0 in (np.array([0., 0., .5, 0.]), np.array([.1, -.05, .4, .2])):
        res = ctrl.qp.solve_batch(x0, fix)
        assert res['status'][0] == 0 and res['polished'][0] > 0
        beq = np.zeros(E.shape[0])
        beq[:nx] = x0
        ref = minimize(lambda w: .5 * w @ H @ w, np.zeros(n), jac=lambda w: H @ w, hess=lambda w: H, method='trust-constr',
                       constraints=[LinearConstraint(E, beq, beq), LinearConstraint(C, -np.inf, h)], bounds=Bounds(lo, hi),
                       options=dict(gtol=1e-12, xtol=1e-14, barrier_tol=1e-12, maxiter=3000))
        assert np.max(C @ ref.x - h) < 1e-9 and np.max(np.abs(E @ ref.x - beq)) < 1e-9
        np.testing.assert_allclose(res['obj'][0], ref.fun, rtol=1e-9)                     # (measured 2e-15)
        X, Xr = res['primal'][0][:(T + 1) * nx], ref.x[:(T + 1) * nx]
        assert np.max(np.abs(X - Xr)) / max(1e-2, np.max(np.abs(Xr))) < 1e-6              # (measured 1.3e-8)


def test_polish_settles_nearly_dependent_active_sets():
    # one-wall system, N=40, initial states over 60 % of the state box: ~2 % of the feasible nodes have active sets
    # (terminal-set facets next to the state bounds they were pushed through the dynamics from) on which the multiplier
    # steps do not settle at the first penalty level; with the second level every optimal node polishes, and the result
    # is the solution of the dense active-set solve
    from helpers import shallow_wide_family
    from dense_qp import dense_qp, active_set_primal
    x0, fix = shallow_wide_family('cart_pole_one_wall', 40, 4000, .6)
    ctrl = make_controller('cart_pole_one_wall', T=40, backend='oracle', threads=8)
    res = ctrl.qp.solve_batch(x0, fix)
    assert np.all(res['status'] <= 1)
    opt = np.flatnonzero(res['status'] == 0)
    assert opt.size > 300 and np.all(res['polished'][opt] > 0), (opt.size, int((res['polished'][opt] == 0).sum()))
    dq = dense_qp(ctrl)
    late = opt[np.argsort(-(res['iters'][opt] & 0xFFFF))][:40]          # the nodes that needed most iterations
    for b in np.concatenate((late, opt[:40])):
        w, resid = active_set_primal(ctrl, dq, x0[b], fix[b], res['dual'][b])
        n = 41 * 4
        assert resid < 1e-10
        assert np.max(np.abs(w[:n] - res['primal'][b][:n])) / max(1e-2, np.max(np.abs(w[:n]))) < 1e-7


def test_degenerate_relaxations_polish_after_tolerance_escalation():
    # BASELINE configs[4] (random MLD nx=20, nu=6+8, N=30): relaxations without strict complementarity -- ~100 rows whose
    # slack AND multiplier vanish.  Read from an iterate of gap 1e-8 the active-set exchange of the polish cycles on 7 % of
    # the nodes of a dive tree (round 3: they returned the interior-point iterate, 1e-4 off); with the tolerance escalation
    # of round 4 (solve_one: a failed last attempt tightens the stopping tolerance and tries again) every one polishes, and
    # the result is the solution of the dense active-set solve, which shares no code with the oracle.
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import dive_tree
    from helpers import random_mld
    from dense_qp import dense_qp, active_set_primal
    mld, objective, x0 = random_mld()
    T, nub, nx = 30, 8, 20
    ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    orc = OracleBatchedQP(ctrl.problem_data(), threads=8)
    Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
    leaf = np.full((1, T * nub), -1, np.int8)
    for t in range(T):
        r = orc.solve_batch(x0, leaf)
        leaf[0, t * nub:(t + 1) * nub] = (r['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
    fix = dive_tree(leaf[0])[0][:120]
    res = orc.solve_batch(x0, fix)
    opt = np.flatnonzero(res['status'] == 0)
    assert opt.size > 100 and np.all(res['polished'][opt] > 0), (opt.size, int((res['polished'][opt] == 0).sum()))
    escalated = opt[res['polished'][opt] > 4]               # (attempt numbers beyond the regular three + the last one)
    assert escalated.size >= 10, escalated.size              # (round 3 left 32 of these 120 unpolished)
    dq = dense_qp(ctrl)
    n = (T + 1) * nx
    for b in escalated[:6]:
        w, resid = active_set_primal(ctrl, dq, x0, fix[b], res['dual'][b])
        assert resid < 1e-10
        assert np.max(np.abs(w[:n] - res['primal'][b][:n])) / max(1e-2, np.max(np.abs(w[:n]))) < 1e-7


def test_product_configuration_against_the_tight_one():
    # The product polishes from an iterate that is only good to 1e-4; the result must still be THE vertex solution.
    # Random prefixes with one initial state per node; this set holds nodes on which a clipped multiplier of -5e-8
    # (sign tolerance of the first polish: 1e-9 relative) was worth 1.5e-4 in the trajectory.
    from helpers import random_prefix_frontier
    rng = np.random.default_rng(5)
    fix = random_prefix_frontier(20, 4, 1024, p_one=0.05)
    x0 = rng.uniform(-1, 1, (1024, 4)) * np.array([.3, .1, .6, .4])
    a = make_controller('cart_pole_with_walls', backend='oracle', threads=8).qp.solve_batch(x0, fix)
    b = make_controller('cart_pole_with_walls', backend='oracle', threads=8, tol=1e-10, polish_tol=1e-8).qp.solve_batch(x0, fix)
    assert np.array_equal(a['status'], b['status'])
    fin = a['status'] == 0
    assert fin.sum() > 150 and np.all(a['polished'][fin] > 0) and np.all(b['polished'][fin] > 0)
    xa, xb = a['primal'][fin][:, :84], b['primal'][fin][:, :84]
    dev = np.max(np.abs(xa - xb), axis=1) / np.maximum(1e-2, np.max(np.abs(xb), axis=1))
    assert dev.max() < 1e-6, dev.max()
    np.testing.assert_allclose(a['obj'][fin], b['obj'][fin], rtol=1e-9, atol=1e-13)


def test_results_do_not_depend_on_batch_or_threads():
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle')
    g = load_fixture('qp_golden')
    fix, x0 = g['n10_fix'], g['n10_x0']
    a = ctrl.qp.solve_batch(x0, fix)
    b = OracleBatchedQP(ctrl.problem_data(), threads=4).solve_batch(x0, fix[::-1].copy())
    assert np.array_equal(a['obj'], b['obj'][::-1]) and np.array_equal(a['dual'], b['dual'][::-1])


def test_speculative_expansion_changes_launches_not_results():
    # descendants solved ahead of time in the same launch; the search must consume
    # the same results in the same order (same incumbent, leaves, solve count), in fewer launches
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=8)
    x0 = np.array([0., 0., .5, 0.])
    base_stats, spec_stats = {}, {}
    sol0, leaves0, solves0, _ = ctrl.feedforward(x0, printing_period=None, stats=base_stats)
    sol1, leaves1, solves1, _ = ctrl.feedforward(x0, printing_period=None, speculation_depth=4, stats=spec_stats)
    assert solves0 == solves1 and abs(solves0 - 80) <= 2 and len(leaves0) == len(leaves1) == 41
    assert [sorted(l.identifier.items()) for l in leaves0] == [sorted(l.identifier.items()) for l in leaves1]
    assert [l.lb for l in leaves0] == [l.lb for l in leaves1]
    assert sol0.objective == sol1.objective
    assert np.array_equal(np.array(sol0.variables['ub']), np.array(sol1.variables['ub']))
    assert base_stats['rounds'] == solves0 and base_stats['speculative'] == 0
    assert spec_stats['rounds'] < 30
    assert spec_stats['launched'] >= solves1 and spec_stats['wasted'] == spec_stats['launched'] - solves1
    # warm-started step: one stage of new binaries, the dive collapses into a couple of launches
    e0 = np.zeros(4)
    ws = ctrl.construct_warm_start(leaves0, x0, sol0.variables['uc'][0], sol0.variables['ub'][0], e0)[0]
    x1 = sol0.variables['x'][1]
    a_stats, b_stats = {}, {}
    import copy
    sa = ctrl.feedforward(x1, warm_start=copy.deepcopy(ws), printing_period=None, stats=a_stats)
    sb = ctrl.feedforward(x1, warm_start=copy.deepcopy(ws), printing_period=None, speculation_depth=4, stats=b_stats)
    assert sa[2] == sb[2] and sa[0].objective == sb[0].objective
    assert b_stats['rounds'] < a_stats['rounds']
