"""Output record of one QP relaxation.

Same three containers, attribute names and dictionary keys as the reference
(``warm_start_hmpc/subproblem_solution.py:4-168``): the primal dictionary has
keys ``'x','uc','ub'`` (lists over time), the dual one ``'lam','mu','nu_lb',
'nu_ub','rho','sigma'``.  The reference fills them by querying Gurobi by
variable name; here they are cut out of the flat rows that the batched solver
returns (layout: ``include/hmpc.h``).
"""
import numpy as np


class SubproblemSolution(object):

    def __init__(self, primal, dual, active_set=None):
        self._primal = primal
        self._dual = dual
        self._rows = None
        self.weak = False
        # the reference hands a simplex basis from parent to child when Gurobi
        # runs its dual simplex (subproblem_solution.py:38-43); the batched
        # interior-point kernels have no basis, the slot stays None
        self.active_set = active_set

    # The containers are cut out of the flat result rows on first use: a batched round returns
    # many nodes (speculative ones included) of which branch and bound opens only a few.
    def _build(self):
        layout, fix_row, obj, dual_obj, primal_row, dual_row = self._rows
        self._rows = None
        self._primal = PrimalSolution.from_row(layout, fix_row, obj, primal_row, not np.isfinite(obj))
        self._dual = DualSolution.from_row(layout, dual_obj, dual_row)
        self._dual.weak = self.weak

    @property
    def primal(self):
        if self._rows is not None:
            self._build()
        return self._primal

    @primal.setter
    def primal(self, value):
        if self._rows is not None:
            self._build()
        self._primal = value

    @property
    def dual(self):
        if self._rows is not None:
            self._build()
        return self._dual

    @dual.setter
    def dual(self, value):
        if self._rows is not None:
            self._build()
        self._dual = value

    def objective_and_feasibility(self):
        """(primal objective, binary feasible) without building the containers."""
        if self._rows is not None:
            _, fix_row, obj, _, _, _ = self._rows
            return (float(obj) if np.isfinite(obj) else np.inf), bool(np.all(np.asarray(fix_row) >= 0))
        return self._primal.objective, self._primal.binary_feasible

    @staticmethod
    def from_rows(layout, fix_row, obj, dual_obj, status, primal_row, dual_row, weak=False):
        """Record of one node from one row of a batch result (the rows are kept by reference).
        ``weak``: the node is infeasible but its ray is no proof to tolerance (HMPC_ITERS_WEAK, include/hmpc.h)."""
        sol = SubproblemSolution(None, None)
        sol._rows = (layout, fix_row, obj, dual_obj, primal_row, dual_row)
        sol.status = int(status)
        sol.weak = bool(weak)
        return sol


class PrimalSolution(object):
    '''
    Primal feasible (not necessarily optimal) solution of the quadratic subproblem.
    '''

    def __init__(self, variables, objective, binary_feasible):
        self.variables = variables
        self.objective = objective
        self.binary_feasible = binary_feasible

    @staticmethod
    def from_row(layout, fix_row, obj, row, infeasible):
        T, nx, nuc, nub, nu = layout.T, layout.nx, layout.nuc, layout.nub, layout.nu
        if infeasible:
            # subproblem_solution.py:86-91 with primal_optimizer -> None
            variables = {'x': [None] * (T + 1), 'uc': [None] * T, 'ub': [None] * T}
            objective = np.inf
        else:
            x = row[:(T + 1) * nx].reshape(T + 1, nx)
            u = row[(T + 1) * nx:].reshape(T, nu)
            variables = {
                'x': [x[t].copy() for t in range(T + 1)],
                'uc': [u[t, :nuc].copy() for t in range(T)],
                'ub': [u[t, nuc:].copy() for t in range(T)],
            }
            objective = float(obj)
        # "binary feasible" = every binary is fixed by the node (subproblem_solution.py:94-97)
        binary_feasible = bool(np.all(np.asarray(fix_row) >= 0))
        return PrimalSolution(variables, objective, binary_feasible)


class DualSolution(object):
    '''
    Dual feasible (not necessarily optimal) solution of the quadratic subproblem.
    '''

    def __init__(self, variables, objective):
        self.variables = variables
        self.objective = objective
        # an infeasibility ray that misses the proof tolerance (the node is infeasible by about the accuracy of the
        # arithmetic): it prunes its node, but the warm-start shift does not carry it to the next step
        self.weak = False

    @staticmethod
    def from_row(layout, dual_obj, row):
        T = layout.T
        cut = layout.dual_slices()
        variables = {}
        variables['lam'] = [row[cut['lam'][t]].copy() for t in range(T + 1)]
        for k in ('mu', 'nu_lb', 'nu_ub', 'sigma'):
            variables[k] = [row[cut[k][t]].copy() for t in range(T)]
        variables['rho'] = [row[cut['rho'][t]].copy() for t in range(T + 1)]
        return DualSolution(variables, float(dual_obj))


class RecordLayout(object):
    """Sizes and offsets of the flat primal/dual rows (mirrors include/hmpc.h)."""

    def __init__(self, nx, nu, nub, T, nc, ncL, nq, nr, nqT):
        self.nx, self.nu, self.nub, self.T = nx, nu, nub, T
        self.nuc = nu - nub
        self.nc, self.ncL, self.nq, self.nr, self.nqT = nc, ncL, nq, nr, nqT
        self.n_primal = (T + 1) * nx + T * nu
        self.n_mu = (T - 1) * nc + ncL
        self.n_dual = (T + 1) * nx + self.n_mu + 2 * T * nub + T * nq + nqT + T * nr
        self._cut = None

    def dual_slices(self):
        if self._cut is None:
            T, nx, nub, nc = self.T, self.nx, self.nub, self.nc
            o = 0
            cut = {}
            cut['lam'] = [slice(o + t * nx, o + (t + 1) * nx) for t in range(T + 1)]
            o += (T + 1) * nx
            cut['mu'] = [slice(o + t * nc, o + (t + 1) * nc) for t in range(T - 1)]
            cut['mu'].append(slice(o + (T - 1) * nc, o + self.n_mu))
            o += self.n_mu
            cut['nu_lb'] = [slice(o + t * nub, o + (t + 1) * nub) for t in range(T)]
            o += T * nub
            cut['nu_ub'] = [slice(o + t * nub, o + (t + 1) * nub) for t in range(T)]
            o += T * nub
            cut['rho'] = [slice(o + t * self.nq, o + (t + 1) * self.nq) for t in range(T)]
            cut['rho'].append(slice(o + T * self.nq, o + T * self.nq + self.nqT))
            o += T * self.nq + self.nqT
            cut['sigma'] = [slice(o + t * self.nr, o + (t + 1) * self.nr) for t in range(T)]
            self._cut = cut
        return self._cut

    def bytes_per_qp(self):
        """ALGORITHMIC bytes per QP of SURVEY.md 8(d): fixing vector in, full record out."""
        return self.T * self.nub + 16 + 8 * (self.n_primal + self.n_dual)
