"""The reference's numerical path -- the node QP solved by Gurobi -- restated against this repository's record layout.

TEST INFRASTRUCTURE, and the one leg of it that needs a Gurobi licence.  ``gurobipy`` is not part of the build image nor of
the GPU boxes: nothing here runs in the regular suites.  Where ``import gurobipy`` succeeds it gives

  * ``GurobiBatchedQP(problem).solve_batch(x0, fix)`` -- the backend interface of the controller (as the CPU oracle and
    the HIP library offer it), one Gurobi solve per node, records in the layout of include/hmpc.h;
  * ``tests/golden/make_gurobi_golden.py`` -- writes ``tests/golden/gurobi_golden.npz`` for the node sets of
    ``qp_golden.npz``; with that file in the tree ``test_gurobi_golden.py`` holds oracle (CPU) and kernel (GPU) to the
    reference's own solver at the 1e-5 of BASELINE.json: the step from "property-pinned" to "vector-pinned";
  * ``bench.py``'s ``cpu_baseline`` with ``kind: "gurobi"``.

What is restated (no reference source is copied; statements and conventions only):

  the QP              /root/reference/warm_start_hmpc/controller.py:119-184   free variables x_0..x_T, u_t = (uc_t, ub_t);
                      rows lam_0: x_0 == x0;  nu_lb_t: -ub_t <= -lb;  nu_ub_t: ub_t <= ub;  lam_{t+1}: x_{t+1} == A x_t + B u_t;
                      mu_t: F x_t + G u_t <= h (last stage: F_Tm1, G_Tm1, h_Tm1);  cost sum |Q x_t|^2 + |R u_t|^2 + |Q_T x_T|^2
  node -> bounds      controller.py:273-327   free binary: (lb, ub) = (0, 1), fixed to v: (v, v), i.e. right-hand sides (-v, v)
  optimize / Farkas   bounded_qp.py:200-228   not OPTIMAL (2) => objective 0, InfUnbdInfo = 1, optimize again, must be
                      INFEASIBLE (3) -- anything else is the reference's AssertionError
  signs               bounded_qp.py:260-332   multipliers = -Pi at an optimum, +FarkasDual for a proof; dual objective =
                      objVal, or - sum RHS * FarkasDual over ALL constraints
  record              subproblem_solution.py:68-168   rho_t = 2 Q x_t, rho_T = 2 Q_T x_T, sigma_t = 2 R u_t; zeros if infeasible

Only ``_GurobiModel`` touches gurobipy (matrix interface: one MVar, two MConstr blocks).  Assembly of the matrices, of a
node's right-hand sides and of the records is plain numpy (``NodeQP``) and IS exercised by the CPU suite, on a stand-in
for the solver (tests/test_gurobi_golden.py::test_record_assembly_on_a_stand_in_solver)."""
import time

import numpy as np


def available():
    try:
        import gurobipy  # noqa: F401
        return True
    except Exception:
        return False


class NodeQP(object):
    """Dense statement of the node QP in the reference's row order, and the maps node -> right-hand sides and
    (solver vectors) -> record.  Variables w = [x_0 .. x_T, u_0 .. u_{T-1}] (the order of the primal record).

    equalities   Eq w == beq      rows: lam_0 (nx), then lam_1 .. lam_T (nx each)                    [(T+1) nx]
    inequalities In w <= bin      rows: mu_0 .. mu_{T-1} (nc each, the last nc + n_T), then nu_lb_t (t major), then nu_ub_t
    cost         w' Hq w          (no 1/2: the reference's objective)
    """

    def __init__(self, problem):
        p = {k: (np.atleast_2d(np.asarray(v, dtype=np.float64)) if k not in ('nx', 'nu', 'nub', 'T', 'h', 'h_Tm1') else v)
             for k, v in problem.items()}
        self.nx, self.nu, self.nub, self.T = int(p['nx']), int(p['nu']), int(p['nub']), int(p['T'])
        nx, nu, nub, T = self.nx, self.nu, self.nub, self.T
        h, hT = np.asarray(problem['h'], dtype=np.float64).ravel(), np.asarray(problem['h_Tm1'], dtype=np.float64).ravel()
        self.nc, self.ncL = h.size, hT.size
        self.Q, self.R, self.QT = p['Q'], p['R'], p['Q_T']
        n = (T + 1) * nx + T * nu
        self.n = n
        xo = lambda t: t * nx                           # noqa: E731
        uo = lambda t: (T + 1) * nx + t * nu            # noqa: E731
        self.xo, self.uo = xo, uo
        Hq = np.zeros((n, n))
        for t in range(T):
            Hq[xo(t):xo(t) + nx, xo(t):xo(t) + nx] = self.Q.T @ self.Q
            Hq[uo(t):uo(t) + nu, uo(t):uo(t) + nu] = self.R.T @ self.R
        Hq[xo(T):xo(T) + nx, xo(T):xo(T) + nx] = self.QT.T @ self.QT
        self.Hq = Hq
        Eq = np.zeros(((T + 1) * nx, n))
        Eq[:nx, :nx] = np.eye(nx)
        for t in range(T):
            r = (t + 1) * nx
            Eq[r:r + nx, xo(t + 1):xo(t + 1) + nx] = np.eye(nx)
            Eq[r:r + nx, xo(t):xo(t) + nx] = -p['A']
            Eq[r:r + nx, uo(t):uo(t) + nu] = -p['B']
        self.Eq = Eq
        rows, rhs = [], []
        for t in range(T):
            F, G, hh = (p['F'], p['G'], h) if t < T - 1 else (p['F_Tm1'], p['G_Tm1'], hT)
            C = np.zeros((hh.size, n))
            C[:, xo(t):xo(t) + nx] = F
            C[:, uo(t):uo(t) + nu] = G
            rows.append(C)
            rhs.append(hh)
        self.n_mu = sum(r.shape[0] for r in rows)
        lo, hi = np.zeros((T * nub, n)), np.zeros((T * nub, n))
        for t in range(T):
            for b in range(nub):
                lo[t * nub + b, uo(t) + (nu - nub) + b] = -1.        # -ub_t <= -lb
                hi[t * nub + b, uo(t) + (nu - nub) + b] = 1.         #  ub_t <=  ub
        self.In = np.vstack(rows + [lo, hi])
        self.b_mu = np.concatenate(rhs)
        self.n_dual = (T + 1) * nx + self.n_mu + 2 * T * nub + T * self.Q.shape[0] + self.QT.shape[0] + T * self.R.shape[0]

    def rhs(self, x0, fix_row):
        """Right-hand sides of a node: (beq, bin).  Free binary: -ub <= 0, ub <= 1; fixed to v: -ub <= -v, ub <= v."""
        beq = np.concatenate((np.asarray(x0, dtype=np.float64), np.zeros(self.T * self.nx)))
        f = np.asarray(fix_row)
        lb = np.where(f >= 0, f, 0).astype(np.float64)
        ub = np.where(f >= 0, f, 1).astype(np.float64)
        return beq, np.concatenate((self.b_mu, -lb, ub))

    def record(self, optimal, w, pi_eq, pi_in, objval, beq, bin_):
        """One record in the layout of include/hmpc.h from what the solver returns: ``pi_*`` are Gurobi's ``Pi`` at an
        optimum and its ``FarkasDual`` for an infeasibility proof.  Returns (obj, dual_obj, status, primal row, dual row)."""
        nx, nu, nub, T = self.nx, self.nu, self.nub, self.T
        dual = np.zeros(self.n_dual)
        o = (T + 1) * nx
        sgn = -1. if optimal else 1.                                  # bounded_qp.py:285 / :290
        dual[:o] = sgn * pi_eq
        dual[o:o + self.n_mu + 2 * T * nub] = sgn * pi_in
        if not optimal:
            proof = -(beq @ pi_eq + bin_ @ pi_in)                     # bounded_qp.py:332 (all constraints)
            return np.inf, proof, 1, np.full(self.n, np.nan), dual
        o += self.n_mu + 2 * T * nub
        x = w[:(T + 1) * nx].reshape(T + 1, nx)
        u = w[(T + 1) * nx:].reshape(T, nu)
        rho = np.concatenate([2. * self.Q @ x[t] for t in range(T)] + [2. * self.QT @ x[T]])
        sig = np.concatenate([2. * self.R @ u[t] for t in range(T)])
        dual[o:o + rho.size] = rho
        dual[o + rho.size:] = sig
        return objval, objval, 0, w.copy(), dual


class _GurobiModel(object):
    """The only code that touches gurobipy: the model of NodeQP, right-hand sides rewritten per node (the reference edits
    one model in place as well, controller.py:254-257), the optimize / Farkas sequence of bounded_qp.py:200-228."""

    def __init__(self, qp, params=None):
        import gurobipy as gp
        from gurobipy import GRB
        self.gp, self.GRB, self.qp = gp, GRB, qp
        self.m = gp.Model()
        self.m.Params.OutputFlag = 0
        for k, v in (params or {}).items():
            self.m.setParam(k, v)
        self.w = self.m.addMVar(qp.n, lb=-GRB.INFINITY, ub=GRB.INFINITY, name='w')
        beq, bin_ = qp.rhs(np.zeros(qp.nx), np.full(qp.T * qp.nub, -1))
        self.ceq = self.m.addMConstr(qp.Eq, self.w, '=', beq)
        self.cin = self.m.addMConstr(qp.In, self.w, '<', bin_)
        self.m.setMObjective(qp.Hq, None, 0.0, sense=GRB.MINIMIZE)
        self.m.update()

    def solve(self, x0, fix_row):
        GRB, m, qp = self.GRB, self.m, self.qp
        beq, bin_ = qp.rhs(x0, fix_row)
        self.ceq.setAttr('RHS', beq)                                  # (explicit attribute calls: documented for MVar / MConstr alike)
        self.cin.setAttr('RHS', bin_)
        m.optimize()                                                  # (between the nodes of a search the reference keeps the solver's state, too)
        runtime = m.Runtime
        if m.Status == GRB.OPTIMAL:
            return qp.record(True, np.array(self.w.getAttr('X')), np.array(self.ceq.getAttr('Pi')), np.array(self.cin.getAttr('Pi')), m.ObjVal, beq, bin_), runtime
        m.setObjective(0.0)
        m.Params.InfUnbdInfo = 1
        m.optimize()
        runtime += m.Runtime
        if m.Status != GRB.INFEASIBLE:
            raise AssertionError('The problem seems to be unbounded.')   # (the reference's words, bounded_qp.py:221)
        out = qp.record(False, None, np.array(self.ceq.getAttr('FarkasDual')), np.array(self.cin.getAttr('FarkasDual')), np.inf, beq, bin_)
        m.setMObjective(qp.Hq, None, 0.0, sense=GRB.MINIMIZE)
        return out, runtime


class GurobiBatchedQP(object):
    """Backend interface of the controller (``solve_batch(x0, fix)``) over Gurobi, one node at a time as the reference."""

    def __init__(self, problem, gurobi_params=None, **_ignored):
        if not available():
            raise RuntimeError('gurobipy is not importable here (no licence / not installed): the Gurobi leg is optional')
        self.node_qp = NodeQP(problem)
        self.model = _GurobiModel(self.node_qp, gurobi_params)
        self.n_primal, self.n_dual = self.node_qp.n, self.node_qp.n_dual

    def solve_batch(self, x0, fix, warm=None):
        fix = np.ascontiguousarray(fix, dtype=np.int8)
        B = fix.shape[0]
        x0 = np.asarray(x0, dtype=np.float64)
        out = dict(obj=np.empty(B), dual_obj=np.empty(B), status=np.empty(B, dtype=np.int32), iters=np.zeros(B, dtype=np.int32),
                   primal=np.empty((B, self.n_primal)), dual=np.empty((B, self.n_dual)), polished=np.ones(B, dtype=np.int32))
        self.model.m.reset()                                           # (controller.py:362: nothing carried over from an earlier search)
        tic, solver_time = time.perf_counter(), 0.
        for b in range(B):
            (obj, dobj, status, prim, dual), rt = self.model.solve(x0 if x0.ndim == 1 else x0[b], fix[b])
            out['obj'][b], out['dual_obj'][b], out['status'][b] = obj, dobj, status
            out['primal'][b], out['dual'][b] = prim, dual
            solver_time += rt
        out['time'] = time.perf_counter() - tic
        out['solver_time'] = solver_time                               # (Gurobi's own Runtime, what the reference logs)
        out['weak'] = np.zeros(B, dtype=np.int32)
        out['second'] = np.zeros(B, dtype=np.int32)
        return out
