"""Offline terminal ingredients: LQR terminal cost, maximal constraint-admissible invariant set, multiplier map.

The reference computes these with Gurobi used as an LP solver (``warm_start_hmpc/mcais.py:10-184`` and
``controller.py:186-227``): one LP per facet and horizon, one per facet of the redundant description, one per row of
``[F_Tm1 G_Tm1]`` -- 130 + ~1100 small LPs for the cart-pole controller.  All of them share one constraint matrix per
sweep and differ in cost / right-hand side only, so every sweep is **one batched launch** of the dense LP kernel
(``csrc/hmpc_lp.hip`` behind ``hmpc_lp_solve_batch``).  ``lp`` is the batched solver
``lp(A, c, b, relax=None) -> dict(obj, x, z, status)`` for ``max c_k'x s.t. A x <= b_k (+1 on row relax[k])``;
the default is the HIP one and fails loudly without the library or a GPU (no CPU path in the product; the tests
pass the oracle's or HiGHS explicitly).
"""
import numpy as np
from scipy.linalg import solve_discrete_are

LP_OPTIMAL, LP_INFEASIBLE, LP_UNBOUNDED = 0, 1, 4


def _hip_lp():
    from .qp_backend import lp_solve_batch  # raises without the HIP library / a GPU
    return lp_solve_batch


def solve_dare(A, B, Q, R):
    """Infinite-horizon LQR: cost-to-go Hessian ``P`` and gain ``K`` (u = K x).

    Same contract as ``mcais.py:10-42``.
    """
    P = solve_discrete_are(A, B, Q, R)
    K = -np.linalg.solve(B.T.dot(P).dot(B) + R, B.T.dot(P).dot(A))
    return P, K


def _values(res, what):
    if np.any(res['status'] != LP_OPTIMAL):
        bad = int(np.flatnonzero(res['status'] != LP_OPTIMAL)[0])
        raise RuntimeError('LP %d of the %s sweep ended with status %d.' % (bad, what, int(res['status'][bad])))
    return res['obj']


def remove_redundant_inequalities(E, f, tol=1.e-7, lp=None):
    """Minimal representation of {x | E x <= f}: one LP per facet, one launch for all of them.

    Facet i is redundant when relaxing it by one unit does not let ``E_i x`` exceed ``f_i`` by ``tol``
    (``mcais.py:146-184``).  Facets already found redundant stay in the LP, as in the reference.
    """
    lp = lp or _hip_lp()
    values = _values(lp(E, E, f, relax=np.arange(E.shape[0])), 'redundancy')
    keep = [i for i in range(E.shape[0]) if values[i] - f[i] >= tol]
    return E[keep], f[keep]


def mcais(A, D, e, verbose=False, lp=None):
    """Maximal constraint-admissible invariant set of x+ = A x in {D x <= e}.

    Gilbert & Tan, Algorithm 3.2, as organised in ``mcais.py:44-144``: at horizon t every original facet is pushed
    t steps through the dynamics and added if some point of the current set violates it.
    """
    lp = lp or _hip_lp()
    if np.max(np.abs(np.linalg.eigvals(A))) > 1.:
        raise ValueError('Unstable system, cannot derive maximal constraint-admissible set.')
    if np.min(e) < 0.:
        raise ValueError('The origin is not in the constraint set, cannot derive maximal constraint-admissible set.')

    D_inf, e_inf = D.copy(), e.copy()
    t = 1
    while True:
        J = D.dot(np.linalg.matrix_power(A, t))
        residuals = _values(lp(D_inf, J, e_inf), 'horizon-%d' % t) - e
        if verbose:
            print(f'Time horizon: {t}. Convergence index: {max(residuals)}. '
                  f'Number of facets: {D_inf.shape[0]}.')
        new_facets = [i for i, r in enumerate(residuals) if r > 0.]
        if not new_facets:
            break
        D_inf = np.vstack((D_inf, J[new_facets]))
        e_inf = np.concatenate((e_inf, e[new_facets]))
        t += 1

    D_inf, e_inf = remove_redundant_inequalities(D_inf, e_inf, lp=lp)
    if verbose:
        print(f'Maximal constraint-admissible invariant set found: {D_inf.shape[0]} minimal facets.')
    return D_inf, e_inf


def update_mu(F, G, h, F_Tm1, G_Tm1, lp=None):
    """Matrix ``M >= 0`` with ``[F G]' M = [F_Tm1 G_Tm1]'``, column by column the cheapest (``min h'm``) nonnegative
    combination (``controller.py:186-227``).  Each column is the multiplier vector of the dual statement
    ``max r_i'y s.t. [F G] y <= h`` -- the shape the batched kernel solves; an unbounded dual is the reference's
    infeasible primal.  Two launches of 130 LPs for the cart-pole controller.

    Raises ``ValueError`` when the conic hull of ``[F G]`` does not contain a row of ``[F_Tm1 G_Tm1]``, as the
    reference does (``controller.py:223-224``).
    """
    lp = lp or _hip_lp()
    E, R = np.hstack((F, G)), np.hstack((F_Tm1, G_Tm1))
    res = lp(E, R, h)
    if np.any(res['status'] == LP_UNBOUNDED):
        raise ValueError('The conic hull of [F G] does not contain the one of [F_Tm1 G_Tm1].')
    _values(res, 'multiplier-map')
    # The optimal multipliers are rarely unique (the rows of [F G] meet in degenerate vertices).  The first launch
    # returns, per column, a point in the relative interior of the optimal face: its support S is the face.  The
    # second picks the vertex a simplex code returns in the reference's own known answer (test_controller.py:47-51:
    # M = I when [F_Tm1 G_Tm1] = [F G]): among the optimal multipliers the one of least total weight
    # sum_r |E_r| m_r -- for a row of [F G] itself that is the unit vector, by the triangle inequality.  Its dual is
    # an LP of the same shape, max r_i'y s.t. E_r y <= |E_r| on S (rows outside S pushed out of the way).
    norms = np.linalg.norm(E, axis=1)
    norms[norms == 0.] = 1.
    # (support read with a relative threshold: the kernel's least-norm correction of the multipliers leaves denormal-size
    # positives on rows outside the face -- 2957 of 3640 entries > 0 but only 886 above 1e-12 on the cart-pole map)
    support = res['z'] > 1.e-9 * np.max(res['z'], axis=1, keepdims=True)
    res = lp(E, R, np.where(support, norms[None, :], 1.e4 * norms[None, :]))
    _values(res, 'multiplier-map (least weight)')
    z = res['z'].copy()
    z[z < 1.e-14 * np.max(z, axis=1, keepdims=True)] = 0.      # (the same residue on the rows off the chosen vertex)
    return np.ascontiguousarray(z.T)
