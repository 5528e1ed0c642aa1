"""The node QP of a controller behind the accessor interface of the reference's ``BoundedQP``.

The reference keeps one mutable Gurobi model per controller
(``warm_start_hmpc/bounded_qp.py:5``, built by ``controller.py:119-184``) and, per
branch-and-bound node, edits right-hand sides by constraint-family name,
optimizes and reads optimizers back by name (``controller.py:254-271``,
``subproblem_solution.py:68-168``).  This class offers that part of the
interface -- same method names, argument meaning, sign conventions and error
behaviour (``bounded_qp.py:127-341``) -- on top of the batched GPU solver, for
callers and tests written against the reference's one-node-at-a-time flow:

    qp = controller.bounded_qp()
    qp.set_constraint_rhs('lam_0', x0)                 # controller.py:257
    qp.set_constraint_rhs('nu_lb_3', -lb)              # controller.py:273-298 (the rhs is MINUS the lower bound)
    qp.set_constraint_rhs('nu_ub_3', ub)
    qp.optimize()                                      # bounded_qp.py:200-228
    qp.primal_optimizer('x_1'); qp.dual_optimizer('mu_0'); qp.dual_objective()

What it is not: a general model builder.  ``add_variables`` / ``add_constraints``
raise -- the QP is the controller's (Appendix A of SURVEY.md), its matrices are
fixed at construction, and only the right-hand sides the reference itself edits
per node can be set: ``lam_0`` (initial state) and ``nu_lb_t`` / ``nu_ub_t``
(bounds of the binaries; each binary either free, (0, 1), or fixed, lb == ub).
"""
import numpy as np

from .subproblem_solution import SubproblemSolution


class BoundedQP(object):

    def __init__(self, controller):
        self._c = controller
        mld, T = controller.mld, controller.T
        self._x0 = np.zeros(mld.nx)
        self._lb = np.zeros((T, mld.nub))
        self._ub = np.ones((T, mld.nub))
        self._solution = None
        self.Runtime = 0.

    # ------------------------------------------------------------------ model building: not offered
    def add_variables(self, n, **kwargs):
        if 'lb' in kwargs or 'ub' in kwargs:       # the reference's own guard (bounded_qp.py:36-37) comes first
            raise KeyError('Cannot set bounds with add_variables, use add_constraints instead.')
        raise NotImplementedError('The QP of a controller is fixed at construction (controller.py:119-184).')

    def add_constraints(self, x, op, y, **kwargs):
        if len(x) != len(y):
            raise ValueError('Left- and right-hand side must have the same size.')
        raise NotImplementedError('The QP of a controller is fixed at construction (controller.py:119-184).')

    # ------------------------------------------------------------------ right-hand sides
    def _family(self, name):
        """(kind, t) of a constraint family name, or (None, None)."""
        T = self._c.T
        for kind in ('nu_lb', 'nu_ub', 'lam', 'mu'):
            if name.startswith(kind + '_'):
                try:
                    t = int(name[len(kind) + 1:])
                except ValueError:
                    return None, None
                last = T if kind == 'lam' else T - 1
                if 0 <= t <= last:
                    return kind, t
        return None, None

    def get_constraint_rhs(self, name):
        """Right-hand side of a family of constraints (bounded_qp.py:170-198); empty if the family does not exist."""
        kind, t = self._family(name)
        c = self._c
        if kind == 'lam':
            return self._x0.copy() if t == 0 else np.zeros(c.mld.nx)     # x_{t} - A x_{t-1} - B u_{t-1} == 0
        if kind == 'nu_lb':
            return -self._lb[t]
        if kind == 'nu_ub':
            return self._ub[t].copy()
        if kind == 'mu':
            return np.array(c.h_Tm1 if t == c.T - 1 else c.mld.h, dtype=np.float64)
        return np.array([])

    def set_constraint_rhs(self, name, rhs):
        """Sets the right-hand side of a family of constraints (bounded_qp.py:127-168)."""
        rhs = np.asarray(rhs, dtype=np.float64).reshape(-1)
        if rhs.size != self.get_constraint_rhs(name).size:
            raise ValueError('The rhs does not have the right dimension.')
        kind, t = self._family(name)
        if kind == 'lam' and t == 0:
            self._x0 = rhs.copy()
        elif kind == 'nu_lb':
            self._lb[t] = -rhs
        elif kind == 'nu_ub':
            self._ub[t] = rhs
        elif rhs.size:
            raise ValueError('Only lam_0, nu_lb_t and nu_ub_t can be edited: the other right-hand sides are part of '
                             'the model that lives on the GPU.')
        self._solution = None

    # ------------------------------------------------------------------ solve
    def reset(self):
        """Forgets the last solution (the reference resets Gurobi's warm-start state, controller.py:362)."""
        self._solution = None

    def optimize(self):
        """Solves the QP; if infeasible the solution holds a Farkas proof (bounded_qp.py:200-228)."""
        fixed = self._lb == self._ub
        free = (self._lb == 0.) & (self._ub == 1.)
        if not np.all(fixed | free) or not np.all(np.isin(self._lb[fixed], (0., 1.))):
            raise ValueError('Each binary must be free, bounds (0, 1), or fixed to 0 or 1 (lb == ub).')
        fix = np.where(fixed, self._lb, -1).astype(np.int8).reshape(1, -1)
        res = self._c.qp.solve_batch(self._x0, fix)
        if res['status'][0] > 1:
            # the reference's counterpart: a solve that is neither optimal nor certified infeasible is an
            # error ('The problem seems to be unbounded.', bounded_qp.py:224-225)
            raise AssertionError('The solver did not converge (status %d).' % res['status'][0])
        self.Runtime = res['time']
        self._solution = SubproblemSolution.from_rows(self._c.layout, fix[0], res['obj'][0], res['dual_obj'][0],
                                                      res['status'][0], res['primal'][0], res['dual'][0])

    def _raise_if_not_solved(self):
        if self._solution is None:
            raise RuntimeError('Problem not solved yet.')

    # ------------------------------------------------------------------ results
    def primal_optimizer(self, name):
        """Optimal value of the variables 'x_t', 'uc_t' or 'ub_t'; None if the QP is infeasible (bounded_qp.py:230-258)."""
        self._raise_if_not_solved()
        for kind in ('x', 'uc', 'ub'):
            if name.startswith(kind + '_'):
                values = self._solution.primal.variables[kind]
                try:
                    t = int(name[len(kind) + 1:])
                except ValueError:
                    break
                if 0 <= t < len(values):
                    return None if values[t] is None else values[t].copy()
        return np.array([])

    def dual_optimizer(self, name):
        """Multipliers of a family of constraints: optimal ones (nonnegative for inequalities) or, if the QP is
        infeasible, the Farkas proof (bounded_qp.py:260-290)."""
        self._raise_if_not_solved()
        kind, t = self._family(name)
        if kind is None:
            return np.array([])
        return self._solution.dual.variables[kind][t].copy()

    def primal_objective(self):
        """Optimal value, inf if infeasible (bounded_qp.py:292-311)."""
        self._raise_if_not_solved()
        return self._solution.primal.objective

    def dual_objective(self):
        """Optimal value of the dual; if infeasible, the cost of the Farkas proof, minus the sum over all
        constraints of rhs times multiplier (bounded_qp.py:313-332)."""
        self._raise_if_not_solved()
        return self._solution.dual.objective

    def solution(self):
        """The record of the last solve as the controller uses it."""
        self._raise_if_not_solved()
        return self._solution
