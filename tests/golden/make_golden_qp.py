"""Regenerates tests/golden/qp_golden.npz: input / expected-output vectors of the hot path.

The reference stores no numeric solutions and its solver (Gurobi) is absent, so these vectors
come from this repo's float64 CPU oracle (oracle/hsde_qp.c) and are accepted only after every
one of them passed the reference's own certificate checks restated in tests/kkt_checks.py
(KKT residuals / Farkas conditions below 1e-6).  They pin (a) the oracle against regressions and
(b) the GPU path on the GPU box, where neither /root/reference nor this script's inputs beyond
the fixtures exist.

    python tests/golden/make_golden_qp.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

from helpers import make_controller, random_prefix_frontier  # noqa: E402
from kkt_checks import check_solution  # noqa: E402
from warm_start_hmpc_amd.subproblem_solution import SubproblemSolution  # noqa: E402


def identifier_of(fix_row, nub):
    return {(k // nub, k % nub): float(v) for k, v in enumerate(fix_row) if v >= 0}


def case(name, T, x0, fix, terminal=True, fixture='cart_pole_with_walls'):
    ctrl = make_controller(fixture, T=T, terminal=terminal, backend='oracle')
    res = ctrl.qp.solve_batch(x0, fix)
    assert np.all(res['status'] <= 1)
    for b in range(fix.shape[0]):
        sol = SubproblemSolution.from_rows(ctrl.layout, fix[b], res['obj'][b], res['dual_obj'][b], res['status'][b],
                                           res['primal'][b], res['dual'][b])
        check_solution(ctrl, sol, identifier_of(fix[b], ctrl.mld.nub), x0, tol=1e-6)
    nx = ctrl.mld.nx
    out = {name + '_T': T, name + '_x0': x0, name + '_fix': fix, name + '_status': res['status'],
           name + '_obj': res['obj'], name + '_dual_obj': res['dual_obj'],
           name + '_x': res['primal'][:, :(T + 1) * nx], name + '_terminal': terminal}
    # branch and bound summary from the same state
    sol, leaves, solves, _ = ctrl.feedforward(x0, printing_period=None)
    out[name + '_bb_solves'] = solves
    out[name + '_bb_leaves'] = len(leaves)
    out[name + '_bb_cost'] = np.inf if sol is None else sol.objective
    out[name + '_bb_ub'] = np.zeros((T, ctrl.mld.nub)) if sol is None else np.array(sol.variables['ub'])
    print(name, 'nodes', fix.shape[0], 'optimal', int((res['status'] == 0).sum()), 'B&B', solves, len(leaves), out[name + '_bb_cost'])
    return out


if __name__ == '__main__':
    data = {}
    f = np.vstack((random_prefix_frontier(20, 4, 40, p_one=0.1), random_prefix_frontier(20, 4, 24, p_one=0.5, seed0=5000)))
    f[0, :] = -1
    data.update(case('n20', 20, np.array([0., 0., 1., 0.]), f))
    # prefixes of the optimal binary assignment of the N=20 problem: feasible relaxations of every depth
    best = data['n20_bb_ub'].astype(np.int8).reshape(-1)
    f = np.full((20, 80), -1, dtype=np.int8)
    for k in range(20):
        f[k, :4 * (k + 1)] = best[:4 * (k + 1)]
    data.update(case('n20dive', 20, np.array([0., 0., 1., 0.]), f))
    f = random_prefix_frontier(10, 4, 24, p_one=0.1, seed0=2000)
    f[0, :] = -1
    data.update(case('n10', 10, np.array([0., 0., .5, 0.]), f))
    data.update(case('n10free', 10, np.array([0., 0., 1., 0.]), f, terminal=False))
    f = random_prefix_frontier(40, 4, 16, p_one=0.1, seed0=3000)
    f[0, :] = -1
    data.update(case('n40', 40, np.array([0., 0., 1., 0.]), f))
    f = random_prefix_frontier(40, 2, 24, p_one=0.1, seed0=4000)
    f[0, :] = -1
    data.update(case('onewall', 40, np.array([0., 0., 1., 0.]), f, fixture='cart_pole_one_wall'))
    np.savez_compressed(os.path.join(HERE, 'qp_golden.npz'), **data)
