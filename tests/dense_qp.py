"""Independent check of a node's trajectory: dense null-space solve on the active set.

TEST INFRASTRUCTURE.  Shares no code with the Riccati / interior-point / polish path of the oracle or
of the HIP kernel: the node QP of the reference (``warm_start_hmpc/controller.py:119-184``) is stated
as dense matrices, the active set is read from the multipliers of a record (``mu > 0``, ``nu > 0``,
fixed binaries), and the equality-constrained QP on it is solved with numpy's SVD (particular solution
+ null-space minimisation).  Where the cost is strictly convex (the states, and the inputs the cost
sees) the result is THE solution of the node if the record's active set is right -- which the KKT
checkers of ``kkt_checks.py`` establish separately.
"""
import numpy as np


def dense_qp(ctrl):
    """H, E, C, h with variables w = [x_0..x_T, u_0..u_{T-1}] (the order of the primal record):
    cost 1/2 w'Hw, dynamics E w = [x0; 0], stage rows C w <= h (bounds of the binaries not included)."""
    m, T = ctrl.mld, ctrl.T
    nx, nu = m.nx, m.nu
    n = (T + 1) * nx + T * nu
    xo = lambda t: t * nx
    uo = lambda t: (T + 1) * nx + t * nu
    H = np.zeros((n, n))
    for t in range(T):
        H[xo(t):xo(t) + nx, xo(t):xo(t) + nx] = 2 * ctrl.Q.T @ ctrl.Q
        H[uo(t):uo(t) + nu, uo(t):uo(t) + nu] = 2 * ctrl.R.T @ ctrl.R
    H[xo(T):xo(T) + nx, xo(T):xo(T) + nx] = 2 * ctrl.Q_T.T @ ctrl.Q_T
    E = np.zeros(((T + 1) * nx, n))
    E[:nx, :nx] = np.eye(nx)
    for t in range(T):
        r = (t + 1) * nx
        E[r:r + nx, xo(t + 1):xo(t + 1) + nx] = np.eye(nx)
        E[r:r + nx, xo(t):xo(t) + nx] = -m.A
        E[r:r + nx, uo(t):uo(t) + nu] = -m.B
    rows, rhs = [], []
    for t in range(T):
        F, G, h = (m.F, m.G, m.h) if t < T - 1 else (ctrl.F_Tm1, ctrl.G_Tm1, ctrl.h_Tm1)
        C = np.zeros((h.size, n))
        C[:, xo(t):xo(t) + nx] = F
        C[:, uo(t):uo(t) + nu] = G
        rows.append(C)
        rhs.append(h)
    return H, E, np.vstack(rows), np.concatenate(rhs)


def active_set_primal(ctrl, dq, x0, fix, dual_row):
    """Primal point of the equality-constrained QP on the active set of ``dual_row`` (flat dual record,
    layout of include/hmpc.h).  Returns (w, residual of the equalities)."""
    H, E, C, h = dq
    m, T = ctrl.mld, ctrl.T
    nx, nu, nub = m.nx, m.nu, m.nub
    n = H.shape[0]
    o = (T + 1) * nx
    mu = dual_row[o:o + C.shape[0]]
    nlb = dual_row[o + C.shape[0]:o + C.shape[0] + T * nub]
    nubb = dual_row[o + C.shape[0] + T * nub:o + C.shape[0] + 2 * T * nub]
    A_rows, A_rhs = [C[mu > 0]], [h[mu > 0]]
    for t in range(T):
        for b in range(nub):
            e = np.zeros((1, n))
            e[0, (T + 1) * nx + t * nu + (nu - nub) + b] = 1.
            f = fix[t * nub + b]
            if f >= 0:
                A_rows.append(e); A_rhs.append([float(f)])
            elif nlb[t * nub + b] > 0:
                A_rows.append(e); A_rhs.append([0.])
            elif nubb[t * nub + b] > 0:
                A_rows.append(e); A_rhs.append([1.])
    Aeq = np.vstack([E] + A_rows)
    beq = np.concatenate([x0, np.zeros(T * nx)] + [np.asarray(a, dtype=float) for a in A_rhs])
    U, s, Vt = np.linalg.svd(Aeq, full_matrices=True)
    r = int((s > 1e-10 * s[0]).sum())
    wp = Vt[:r].T @ ((U[:, :r].T @ beq) / s[:r])
    Z = Vt[r:].T
    y = np.linalg.lstsq(Z.T @ H @ Z, -Z.T @ (H @ wp), rcond=1e-13)[0]
    w = wp + Z @ y
    return w, np.max(np.abs(Aeq @ w - beq))


def determined_inputs(ctrl):
    """Indices of the inputs the cost is strictly convex in (a column of R of their own): these, and the
    states, are unique at every node; the other inputs of a relaxation may not be."""
    RtR = ctrl.R.T @ ctrl.R
    return [j for j in range(ctrl.mld.nu) if RtR[j, j] > 0 and np.count_nonzero(RtR[j]) == 1]
