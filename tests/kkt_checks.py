"""Solver-agnostic certificate checks for one QP relaxation.

These restate, for any controller of this package, the three checkers that
the reference's test fixture defines for its one-wall cart-pole
(``warm_start_hmpc/test/cart_pole_with_wall.py:171-268``): primal
feasibility, dual feasibility (stationarity + sign) and the dual objective.
They are the executable definition of "this point is a KKT point / a Farkas
proof" and therefore the way both the CPU oracle and the GPU path are pinned
in the absence of Gurobi outputs.
"""
import numpy as np


def primal_residuals(ctrl, variables, identifier, x0):
    """Returns (equality residuals, slacks that must be >= 0)."""
    mld, T = ctrl.mld, ctrl.T
    x, uc, ub = variables['x'], variables['uc'], variables['ub']
    u = [np.concatenate((uc[t], ub[t])) for t in range(T)]
    lo, hi = ctrl._get_bound_binaries(identifier)
    zero = [x0 - x[0]]
    for t in range(T):
        zero.append(mld.A.dot(x[t]) + mld.B.dot(u[t]) - x[t + 1])
    nonneg = []
    for t in range(T - 1):
        nonneg.append(mld.h - mld.F.dot(x[t]) - mld.G.dot(u[t]))
    nonneg.append(ctrl.h_Tm1 - ctrl.F_Tm1.dot(x[T - 1]) - ctrl.G_Tm1.dot(u[T - 1]))
    for t in range(T):
        nonneg.append(ub[t] - lo[t])
        nonneg.append(hi[t] - ub[t])
    return np.concatenate(zero), np.concatenate(nonneg)


def dual_residuals(ctrl, variables):
    """Stationarity residuals (must vanish) and multipliers that must be >= 0."""
    mld, T = ctrl.mld, ctrl.T
    rho, lam, sigma = variables['rho'], variables['lam'], variables['sigma']
    mu, nu_lb, nu_ub = variables['mu'], variables['nu_lb'], variables['nu_ub']
    zero = [ctrl.Q_T.T.dot(rho[T]) + lam[T]]
    for t in range(T):
        F, G = (mld.F, mld.G) if t < T - 1 else (ctrl.F_Tm1, ctrl.G_Tm1)
        zero.append(ctrl.Q.T.dot(rho[t]) + lam[t] - mld.A.T.dot(lam[t + 1]) + F.T.dot(mu[t]))
        zero.append(ctrl.R.T.dot(sigma[t]) - mld.B.T.dot(lam[t + 1]) + G.T.dot(mu[t])
                    + mld.V.T.dot(nu_ub[t] - nu_lb[t]))
    nonneg = np.concatenate(list(mu) + list(nu_lb) + list(nu_ub))
    return np.concatenate(zero), nonneg


def dual_objective(ctrl, variables, identifier, x0):
    """Value of the Lagrangian dual at the given multipliers."""
    mld = ctrl.mld
    lo, hi = ctrl._get_bound_binaries(identifier)
    val = 0.
    for k in ('rho', 'sigma'):
        val -= sum(v.dot(v) for v in variables[k]) / 4.
    val -= variables['lam'][0].dot(x0)
    val += sum(lo[t].dot(v) for t, v in enumerate(variables['nu_lb']))
    val -= sum(hi[t].dot(v) for t, v in enumerate(variables['nu_ub']))
    val -= sum(mld.h.dot(v) for v in variables['mu'][:-1])
    val -= ctrl.h_Tm1.dot(variables['mu'][-1])
    return val


def primal_objective(ctrl, variables):
    T = ctrl.T
    val = 0.
    for t in range(T):
        u = np.concatenate((variables['uc'][t], variables['ub'][t]))
        Qx, Ru = ctrl.Q.dot(variables['x'][t]), ctrl.R.dot(u)
        val += Qx.dot(Qx) + Ru.dot(Ru)
    QxT = ctrl.Q_T.dot(variables['x'][T])
    return val + QxT.dot(QxT)


def check_solution(ctrl, solution, identifier, x0, tol=1e-6):
    """Asserts that ``solution`` certifies itself: a KKT point if its
    objective is finite, a Farkas proof of infeasibility otherwise."""
    zero, nonneg = dual_residuals(ctrl, solution.dual.variables)
    scale = 1. + max(np.max(np.abs(np.concatenate(solution.dual.variables[k]))) for k in ('lam', 'mu'))
    assert np.max(np.abs(zero)) <= tol * scale, ('dual stationarity', np.max(np.abs(zero)), scale)
    assert np.min(nonneg) >= -tol * scale, ('dual sign', np.min(nonneg))
    dobj = dual_objective(ctrl, solution.dual.variables, identifier, x0)
    if np.isinf(solution.primal.objective):
        assert all(np.all(v == 0) for v in solution.dual.variables['rho'])
        assert all(np.all(v == 0) for v in solution.dual.variables['sigma'])
        assert dobj > 0., ('farkas objective', dobj)
        assert abs(dobj - solution.dual.objective) <= tol * (1 + abs(dobj)), ('farkas objective', dobj, solution.dual.objective)
        return 'infeasible'
    zero, nonneg = primal_residuals(ctrl, solution.primal.variables, identifier, x0)
    assert np.max(np.abs(zero)) <= tol, ('primal equality', np.max(np.abs(zero)))
    assert np.min(nonneg) >= -tol, ('primal inequality', np.min(nonneg))
    pobj = primal_objective(ctrl, solution.primal.variables)
    assert abs(pobj - solution.primal.objective) <= tol * (1 + abs(pobj))
    assert abs(pobj - dobj) <= tol * (1 + abs(pobj)), ('duality gap', pobj, dobj)
    assert abs(dobj - solution.dual.objective) <= tol * (1 + abs(dobj))
    return 'optimal'


def is_disjoint_cover(ctrl, nodes, n_samples=100, seed=1):
    """Random vertices of the binary cube must lie in exactly one node
    (cart_pole_with_wall.py:147-169)."""
    rng = np.random.RandomState(seed)
    boxes = [tuple(np.concatenate(b) for b in ctrl._get_bound_binaries(n.identifier)) for n in nodes]
    for _ in range(n_samples):
        v = rng.randint(0, 2, ctrl.mld.nub * ctrl.T)
        hits = sum(1 for lo, hi in boxes if np.all(v >= lo) and np.all(v <= hi))
        if hits != 1:
            return False
    return True
