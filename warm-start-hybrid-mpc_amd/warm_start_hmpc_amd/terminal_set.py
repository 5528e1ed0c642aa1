"""Offline terminal ingredients (run once per controller, on the host CPU).

The reference computes these with Gurobi used as an LP solver
(``warm_start_hmpc/mcais.py:10-184`` and ``controller.py:186-227``).  They are
not on the hot path; they are restated here on top of
HiGHS (``scipy.optimize.linprog``) because the cart-pole-with-walls controller
needs their output as *input data*: the LQR terminal cost, the maximal
constraint-admissible invariant set, and the matrix ``M`` that maps the last
stage multiplier to the previous stage in the warm-start shift.
"""
import numpy as np
from scipy.linalg import solve_discrete_are
from scipy.optimize import linprog


def solve_dare(A, B, Q, R):
    """Infinite-horizon LQR: cost-to-go Hessian ``P`` and gain ``K`` (u = K x).

    Same contract as ``mcais.py:10-42``.
    """
    P = solve_discrete_are(A, B, Q, R)
    K = -np.linalg.solve(B.T.dot(P).dot(B) + R, B.T.dot(P).dot(A))
    return P, K


def _maximize(c, D, e):
    """max c'x s.t. D x <= e, x free.  Returns the optimal value."""
    res = linprog(-c, A_ub=D, b_ub=e, bounds=(None, None), method='highs')
    if res.status != 0:
        raise RuntimeError('LP failed in terminal-set computation: ' + res.message)
    return -res.fun


def remove_redundant_inequalities(E, f, tol=1.e-7):
    """Minimal representation of {x | E x <= f}: one LP per facet.

    Facet i is redundant when relaxing it by one unit does not let ``E_i x``
    exceed ``f_i`` by ``tol`` (``mcais.py:146-184``).  Facets already found
    redundant stay in the LP, as in the reference.
    """
    keep = []
    for i in range(E.shape[0]):
        f_relaxed = f.copy()
        f_relaxed[i] += 1.
        if _maximize(E[i], E, f_relaxed) - f[i] >= tol:
            keep.append(i)
    return E[keep], f[keep]


def mcais(A, D, e, verbose=False):
    """Maximal constraint-admissible invariant set of x+ = A x in {D x <= e}.

    Gilbert & Tan, Algorithm 3.2, as organised in ``mcais.py:44-144``: at
    horizon t every original facet is pushed t steps through the dynamics and
    added if some point of the current set violates it.
    """
    if np.max(np.abs(np.linalg.eigvals(A))) > 1.:
        raise ValueError('Unstable system, cannot derive maximal constraint-admissible set.')
    if np.min(e) < 0.:
        raise ValueError('The origin is not in the constraint set, cannot derive maximal constraint-admissible set.')

    D_inf, e_inf = D.copy(), e.copy()
    t = 1
    while True:
        J = D.dot(np.linalg.matrix_power(A, t))
        residuals = [_maximize(J[i], D_inf, e_inf) - e[i] for i in range(D.shape[0])]
        if verbose:
            print(f'Time horizon: {t}. Convergence index: {max(residuals)}. '
                  f'Number of facets: {D_inf.shape[0]}.')
        new_facets = [i for i, r in enumerate(residuals) if r > 0.]
        if not new_facets:
            break
        D_inf = np.vstack((D_inf, J[new_facets]))
        e_inf = np.concatenate((e_inf, e[new_facets]))
        t += 1

    D_inf, e_inf = remove_redundant_inequalities(D_inf, e_inf)
    if verbose:
        print(f'Maximal constraint-admissible invariant set found: {D_inf.shape[0]} minimal facets.')
    return D_inf, e_inf


def update_mu(F, G, h, F_Tm1, G_Tm1):
    """Matrix ``M >= 0`` with ``[F G]' M = [F_Tm1 G_Tm1]'``, column by column the
    cheapest (``min h'm``) nonnegative combination (``controller.py:186-227``).

    Raises ``ValueError`` when the conic hull of ``[F G]`` does not contain a
    row of ``[F_Tm1 G_Tm1]``, as the reference does (``controller.py:223-224``).
    """
    lhs = np.vstack((F.T, G.T))
    columns = []
    for i in range(F_Tm1.shape[0]):
        rhs = np.concatenate((F_Tm1[i], G_Tm1[i]))
        res = linprog(h, A_eq=lhs, b_eq=rhs, bounds=(0., None), method='highs')
        if res.status != 0:
            raise ValueError('The conic hull of [F G] does not contain the one of [F_Tm1 G_Tm1].')
        columns.append(res.x)
    return np.vstack(columns).T
