"""Experiment (CPU, oracle backend; hand-run, not a test): why warm-started searches solve FEWER nodes here than in the
reference's published runs (9.1 against 12.6 per step, profiles/mc_r04/README.md).

A warm-started step solves the dive through the entering stage's binaries (1 + 2 nub = 9 nodes) plus every leaf of the cover
whose infeasibility proof did not survive the shift (controller.py:555-558: shifted dual objective <= 0 -> lb = 0, solved again).
Which Farkas ray a solver returns for an infeasible node is not unique.  This repository's rays come from solves with the
terminal-set rows masked (lazy terminal set): they carry no terminal multipliers.  The reference's come from Gurobi's simplex:
BASIC rays (vertices of the ray polytope) of the node's full row set.  Gurobi is not available; the nearest thing is: for every
infeasible leaf, the vertex certificate HiGHS (scipy, dual simplex) returns for the elastic phase-1 LP of the node's rows --
min sum(s) s.t. E w = b, A w - s <= rhs, s >= 0, whose optimal multipliers (0 <= z <= 1) are a basic Farkas ray --, pushed
through the SAME shift (construct_warm_start of this repository, which agrees with the reference's to 1e-13, test_bb_traces.py).
Counted per step along the published disturbances of sd = .001: leaves reopened with this repository's rays, with the basic
rays, and with the rays of solves that keep the terminal rows (lazy_terminal=False).

    python tests/cpu_basic_rays.py [simulations] [steps]      ->  profiles/r05_warm_solve_gap.txt
"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from scipy.optimize import linprog
from scipy.sparse import csr_matrix, hstack, identity, vstack

from helpers import make_controller, load_fixture
from dense_qp import dense_qp
from kkt_checks import dual_residuals, dual_objective
from warm_start_hmpc_amd.subproblem_solution import DualSolution, SubproblemSolution

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
full = make_controller('cart_pole_with_walls', backend='oracle', threads=8, lazy_terminal=False)
ref = load_fixture('reference_closed_loop')
m, T = ctrl.mld, ctrl.T
nx, nu, nub, nuc = m.nx, m.nu, m.nub, m.nu - m.nub
H, E, C, h = dense_qp(ctrl)
n = H.shape[0]
Vb = np.zeros((T * nub, n))
for t in range(T):
    for b in range(nub):
        Vb[t * nub + b, (T + 1) * nx + t * nu + nuc + b] = 1.
A_in = csr_matrix(np.vstack((C, -Vb, Vb)))
m_in = A_in.shape[0]
A_ub = hstack((A_in, -identity(m_in, format='csr')), format='csr')
A_eq = hstack((csr_matrix(E), csr_matrix((E.shape[0], m_in))), format='csr')
cost = np.concatenate((np.zeros(n), np.ones(m_in)))
bounds = [(None, None)] * n + [(0, None)] * m_in


A_ub1 = hstack((A_in, csr_matrix(-np.ones((m_in, 1)))), format='csr')          # one scalar t: A w - t 1 <= rhs, min t
A_eq1 = hstack((csr_matrix(E), csr_matrix((E.shape[0], 1))), format='csr')
rng = np.random.RandomState(0)


def basic_ray(identifier, x0, kind='sum'):
    """kind: 'sum' min sum(s) (0 <= z <= 1); 'max' min t with ONE elastic variable (sum z = 1); 'random': min c's with random
    weights in [0.2, 5] (z <= c) -- three of the many vertex certificates a simplex code may return."""
    lo, hi = ctrl._get_bound_binaries(identifier)
    rhs = np.concatenate((h, -np.concatenate(lo), np.concatenate(hi)))
    beq = np.concatenate((x0, np.zeros(T * nx)))
    if kind == 'max':
        res = linprog(np.concatenate((np.zeros(n), [1.])), A_ub=A_ub1, b_ub=rhs, A_eq=A_eq1, b_eq=beq, bounds=[(None, None)] * n + [(0, None)], method='highs-ds')
    else:
        cc = cost if kind == 'sum' else np.concatenate((np.zeros(n), np.exp(rng.uniform(np.log(.2), np.log(5.), m_in))))
        res = linprog(cc, A_ub=A_ub, b_ub=rhs, A_eq=A_eq, b_eq=beq, bounds=bounds, method='highs-ds')
    assert res.status == 0 and res.fun > 1e-9, (res.status, res.fun)
    z, y = -res.ineqlin.marginals, -res.eqlin.marginals      # (HiGHS returns -1e-8 on some nonbasic rows: its dual feasibility tolerance)
    nmu = C.shape[0]
    nc = m.F.shape[0]
    var = {'lam': [y[t * nx:(t + 1) * nx] for t in range(T + 1)],
           'mu': [z[t * nc:(t + 1) * nc] for t in range(T - 1)] + [z[(T - 1) * nc:nmu]],
           'nu_lb': [z[nmu + t * nub:nmu + (t + 1) * nub] for t in range(T)],
           'nu_ub': [z[nmu + T * nub + t * nub:nmu + T * nub + (t + 1) * nub] for t in range(T)],
           'rho': [np.zeros(ctrl.Q.shape[0])] * T + [np.zeros(ctrl.Q_T.shape[0])], 'sigma': [np.zeros(ctrl.R.shape[0])] * T}
    zero, nonneg = dual_residuals(ctrl, var)
    scale = max(1., np.max(np.abs(z)))
    assert np.max(np.abs(zero)) < 1e-7 * scale and nonneg.min() > -1e-6, (np.max(np.abs(zero)), nonneg.min())
    dobj = dual_objective(ctrl, var, identifier, x0)
    assert dobj > 0 and abs(dobj - res.fun) < 1e-5 * (1 + res.fun), (dobj, res.fun)
    return DualSolution(var, dobj), int((z[(T - 1) * nc + nc:nmu] > 1e-12).sum())


def reopened(cover):
    return sum(1 for node in cover if node.extra.dual is None)


rows = []
tic = time.time()
for sim in range(S):
    x, ws = np.array([0., 0., 1., 0.]), None
    for t in range(STEPS):
        sol, leaves, solves, _ = ctrl.feedforward(x, warm_start=ws, printing_period=None)
        if sol is None:
            break
        e = ref['errors_0001'][sim, t]
        uc0, ub0 = sol.variables['uc'][0], sol.variables['ub'][0]
        ws, _, _ = ctrl.construct_warm_start(leaves, x, uc0, ub0, e)
        inf = [l for l in leaves if np.isinf(l.lb) and ctrl._retain_leaf(l.identifier, ub0)]
        # the same leaves with basic rays, and with the rays of solves that keep the terminal rows
        swapped, swapped1, swapped2, termful, with_term = [], [], [], [], 0
        fr = full.solve_frontier([l.identifier for l in inf], x)[0] if inf else []
        k = 0
        for l in leaves:
            a, a1, a2, b = copy.copy(l), copy.copy(l), copy.copy(l), copy.copy(l)
            if np.isinf(l.lb) and ctrl._retain_leaf(l.identifier, ub0):
                d, nt = basic_ray(l.identifier, x)
                with_term += nt > 0
                a.extra = SubproblemSolution(None, d)
                a1.extra = SubproblemSolution(None, basic_ray(l.identifier, x, 'max')[0])
                a2.extra = SubproblemSolution(None, basic_ray(l.identifier, x, 'random')[0])
                b.extra = SubproblemSolution(None, fr[k].dual)
                k += 1
            swapped.append(a)
            swapped1.append(a1)
            swapped2.append(a2)
            termful.append(b)
        wb, _, _ = ctrl.construct_warm_start(swapped, x, uc0, ub0, e)
        wb1, _, _ = ctrl.construct_warm_start(swapped1, x, uc0, ub0, e)
        wb2, _, _ = ctrl.construct_warm_start(swapped2, x, uc0, ub0, e)
        wf, _, _ = ctrl.construct_warm_start(termful, x, uc0, ub0, e)
        assert len(wb) == len(ws) == len(wf)
        rows.append((sim, t, solves, len(ws), len(inf), reopened(ws), reopened(wb), reopened(wf), with_term,
                     int(ref['nodes_ws_0001'][sim, t]), int(ref['nodes_len_ws_0001'][sim, t]), reopened(wb1), reopened(wb2)))
        x = sol.variables['x'][1] + e
    print('simulation %d done (%.0f s)' % (sim, time.time() - tic), flush=True)
r = np.array(rows)
warm = r[r[:, 1] > 0]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r05_warm_solve_gap.txt')
with open(out, 'w') as f:
    def say(s=''):
        print(s)
        f.write(s + '\n')
    say('# tests/cpu_basic_rays.py %d %d: first %d published simulations of sd = .001, %d steps each (CPU oracle; the shift is this' % (S, STEPS, S, STEPS))
    say('# repository\'s construct_warm_start, equal to the reference\'s to 1e-13).  Per step: the cover the shift builds from the leaves of the')
    say('# step, how many of its leaves were infeasible, and how many of those LOSE their proof in the shift (are reopened: one more solve')
    say('# in the next step) with three kinds of Farkas rays for the same leaves.')
    say('cover size == published on %d of %d steps' % (int((r[:, 3] == r[:, 10]).sum()), len(r)))
    say('infeasible leaves in a cover (mean): %.1f of %.1f' % (r[:, 4].mean(), r[:, 3].mean()))
    say('reopened per shift, rays of THIS repository (terminal rows masked: no terminal multipliers):   %.2f' % r[:, 5].mean())
    say('reopened per shift, BASIC rays (HiGHS dual simplex on the phase-1 LP of the full row set):      %.2f   (%.1f of the rays carry terminal multipliers)' % (r[:, 6].mean(), r[:, 8].mean()))
    say('reopened per shift, BASIC rays, phase 1 with ONE elastic variable (min of the largest violation):  %.2f' % r[:, 11].mean())
    say('reopened per shift, BASIC rays, phase 1 with random weights on the elastic variables:             %.2f' % r[:, 12].mean())
    say('reopened per shift, interior-point rays of solves WITH the terminal rows (lazy_terminal=False): %.2f' % r[:, 7].mean())
    say('warm solves per step here (steps >= 1): %.2f   published: %.2f   difference %.2f' % (warm[:, 2].mean(), warm[:, 9].mean(), warm[:, 9].mean() - warm[:, 2].mean()))
    say('reopened with basic rays minus reopened with this repository\'s rays: %.2f per step' % (r[:, 6].mean() - r[:, 5].mean()))
    say()
    say('sim step solves cover infeasible reopened(own) reopened(basic) reopened(with terminal rows) basic_rays_with_terminal_multipliers published_solves published_cover reopened(basic, one elastic variable) reopened(basic, random weights)')
    for row in rows:
        say(' '.join(str(v) for v in row))
