"""Hybrid MPC controller: drop-in for the reference's Gurobi-backed
``HybridModelPredictiveController`` (``warm_start_hmpc/controller.py:46-843``).

Solves, by branch and bound over the binary inputs,

    min  |Q_T x_T|^2 + sum_{t<T} |Q x_t|^2 + |R u_t|^2
    s.t. x_0 given,  x_{t+1} = A x_t + B u_t,  F x_t + G u_t <= h,
         F_T x_T <= h_T,  V u_t binary.

Kept from the reference: constructor signature, ``feedforward`` and
``construct_warm_start`` signatures and return tuples, ``_solve_subproblem``,
``_get_bound_binaries``, ``branch_in_time``, the node-shift semantics of the
warm start.  Replaced: the QP relaxations are not
solved one at a time on a mutable Gurobi model but as batches on the GPU
through the C-ABI library of ``include/hmpc.h``.  Added: ``feedback`` (the
reference has no such method; it is the closed-loop step of
``notebooks/cart_pole_with_walls/plot_trajectory.py:19-45``) and
``solve_frontier`` (a whole synthetic frontier in one call).
"""
import gc
from time import time

import numpy as np

from .subproblem_solution import SubproblemSolution, DualSolution, RecordLayout
from .branch_and_bound import Node, branch_and_bound, best_first, depth_first, breadth_first  # noqa: F401
from .terminal_set import update_mu


def branch_in_time(identifier, nub):
    '''
    Branching rule: fix the next binary in chronological (t, i) order
    (same contract as controller.py:13-44).  Returns the two sub-identifiers,
    zero branch first.
    '''
    if identifier:
        t = max(k[0] for k in identifier)
        i = max(k[1] for k in identifier if k[0] == t) + 1
    else:
        t, i = 0, 0
    if i >= nub:
        t, i = t + 1, 0
    return [{(t, i): 0.}, {(t, i): 1.}]


class _ShiftMaps(dict):
    """``{'mu': M, 'rho': ...}`` of controller.py:94-97; ``'mu'`` is computed by the controller's LP solver (its
    ``lp=`` argument, else its backend's ``lp_solve_batch``, else the HIP LP kernel) the first time it is read."""

    def __init__(self, controller):
        dict.__init__(self)
        self._controller = controller

    def __missing__(self, key):
        if key != 'mu':
            raise KeyError(key)
        c = self._controller
        lp = c._lp or getattr(c.qp, 'lp_solve_batch', None)
        self['mu'] = update_mu(c.mld.F, c.mld.G, c.mld.h, c.F_Tm1, c.G_Tm1, lp=lp)
        return self['mu']


class HybridModelPredictiveController(object):

    def __init__(self, mld, T, objective, terminal_set, backend=None, solver_params=None, lp=None):
        '''
        Parameters
        ----------
        mld : MLDSystem (this package's or the reference's)
        T : int, horizon
        objective : [Q, R, Q_T]
        terminal_set : [F_T, h_T] or None
        backend : object with ``solve_batch(x0, fix)``; default: the HIP
            library on the current GPU (raises if it cannot be loaded --
            there is no CPU fallback in the product path).
        solver_params : dict, options of the batched QP solver
            (``tol``, ``max_iter``); replaces the reference's ``gurobi_params``.
        lp : batched LP solver for the multiplier map of the warm start
            (``terminal_set.update_mu``, controller.py:186-227); default: the
            backend's ``lp_solve_batch`` if it has one, else the HIP LP kernel.
        '''
        self.mld = mld
        self.T = int(T)
        self.Q, self.R, self.Q_T = [np.atleast_2d(np.asarray(M, dtype=np.float64)) for M in objective]

        # the terminal constraint is enforced through the stage constraint of
        # time T-1 (controller.py:81-87)
        if terminal_set is None:
            terminal_set = [np.empty((0, mld.nx)), np.empty(0)]
        F_T = np.atleast_2d(np.asarray(terminal_set[0], dtype=np.float64))
        self.F_Tm1 = np.vstack((mld.F, F_T.dot(mld.A)))
        self.G_Tm1 = np.vstack((mld.G, F_T.dot(mld.B)))
        self.h_Tm1 = np.concatenate((mld.h, np.asarray(terminal_set[1], dtype=np.float64)))
        self._check_input_sizes()

        self.layout = RecordLayout(mld.nx, mld.nu, mld.nub, self.T, mld.h.size, self.h_Tm1.size,
                                   self.Q.shape[0], self.R.shape[0], self.Q_T.shape[0])

        # warm start construction (controller.py:94-97).  The multiplier map is one batched launch of the LP kernel
        # (controller.py:186-227); it is computed here, as in the reference, whenever an LP solver is at hand, and on
        # first use when the controller is built around a placeholder backend that is bound afterwards.
        self._lp = lp
        self._update = _ShiftMaps(self)
        self._update['rho'] = np.linalg.pinv(self.Q.T).dot(self.Q_T.T)

        self.solver_params = dict(solver_params or {})
        # parent -> child hand-down of active sets in feedforward (the reference's optional simplex-basis hand-down,
        # controller.py:260-264): off by default, as in the reference's published runs
        self.handdown = bool(self.solver_params.pop('handdown', False))
        if backend is None:
            from .qp_backend import HipBatchedQP  # fails loudly without the HIP library / a GPU
            backend = HipBatchedQP(self.problem_data(), **self.solver_params)
        self.qp = backend
        if lp is not None or hasattr(backend, 'lp_solve_batch'):
            self._update['mu']

    def problem_data(self):
        """Node-independent data of the QP, as the C ABI takes it (include/hmpc.h)."""
        mld = self.mld
        return dict(nx=mld.nx, nu=mld.nu, nub=mld.nub, T=self.T,
                    A=mld.A, B=mld.B, F=mld.F, G=mld.G, h=mld.h,
                    F_Tm1=self.F_Tm1, G_Tm1=self.G_Tm1, h_Tm1=self.h_Tm1,
                    Q=self.Q, R=self.R, Q_T=self.Q_T)

    def _check_input_sizes(self):
        if self.Q.shape[1] != self.mld.nx:
            raise ValueError('Matrix Q has wrong number of columns.')
        if self.R.shape[1] != self.mld.nu:
            raise ValueError('Matrix R has wrong number of columns.')
        if self.Q_T.shape[1] != self.mld.nx:
            raise ValueError('Matrix Q_T has wrong number of columns.')
        if self.F_Tm1.shape[0] != self.h_Tm1.size or self.F_Tm1.shape[1] != self.mld.nx:
            raise ValueError('Terminal-set matrices have wrong number of rows.')
        if self.G_Tm1.shape[0] != self.h_Tm1.size:
            raise ValueError('Terminal-set matrices have wrong number of rows.')

    # ------------------------------------------------------------------
    # node <-> bounds
    # ------------------------------------------------------------------

    def _fix_vector(self, identifier):
        """identifier -> int8 row of length T*nub: -1 free, 0 / 1 fixed."""
        fix = np.full(self.T * self.mld.nub, -1, dtype=np.int8)
        nub = self.mld.nub
        for (t, i), v in identifier.items():
            fix[t * nub + i] = int(v)
        return fix

    def bounded_qp(self):
        """The node QP behind the accessor interface of the reference's ``BoundedQP`` (see ``bounded_qp.py``)."""
        from .bounded_qp import BoundedQP
        return BoundedQP(self)

    def _set_bound_binaries(self, identifier, qp):
        '''
        Writes the bounds that an identifier imposes on the binaries into the
        right-hand sides of ``qp`` (a ``BoundedQP`` of this controller), as
        controller.py:273-298 does on the reference's Gurobi model: the rhs of
        nu_lb_t is MINUS the lower bound, the rhs of nu_ub_t the upper bound.
        '''
        ub_lb, ub_ub = self._get_bound_binaries(identifier)
        for t in range(self.T):
            qp.set_constraint_rhs('nu_lb_%d' % t, -ub_lb[t])
            qp.set_constraint_rhs('nu_ub_%d' % t, ub_ub[t])

    def _get_bound_binaries(self, identifier):
        '''
        Lower and upper bounds that an identifier imposes on the binaries,
        two arrays of shape (T, nub) (controller.py:300-327).
        '''
        ub_lb = np.zeros((self.T, self.mld.nub))
        ub_ub = np.ones((self.T, self.mld.nub))
        for k, v in identifier.items():
            ub_lb[k] = v
            ub_ub[k] = v
        return ub_lb, ub_ub

    # ------------------------------------------------------------------
    # QP relaxations
    # ------------------------------------------------------------------

    def solve_frontier(self, identifiers, x0, active_sets=None):
        """Solves the QP relaxations of many nodes in one batched call.

        identifiers : list of dict, or an int8 array (B, T*nub) with -1 = free
        x0 : (nx,) shared by all nodes, or (B, nx)
        active_sets : optional list, per node the ``active_set`` of its parent's solution (what ``_brancher`` hands to a
            child, controller.py:426) or None -- the parent -> child hand-down of ``hmpc_warm`` (include/hmpc.h)
        Returns (list of SubproblemSolution, solver wall time in seconds).
        """
        if isinstance(identifiers, np.ndarray):
            fix = np.ascontiguousarray(identifiers, dtype=np.int8)
        else:
            fix = np.stack([self._fix_vector(i) for i in identifiers])
        warm = None
        if active_sets is not None and any(a is not None for a in active_sets):
            rows = [a for a in active_sets if a is not None]
            index = np.full(fix.shape[0], -1, dtype=np.int32)
            index[[b for b, a in enumerate(active_sets) if a is not None]] = np.arange(len(rows), dtype=np.int32)
            warm = (np.stack([a[0] for a in rows]), np.stack([a[1] for a in rows]), index)
        res = (self.qp.solve_batch(np.asarray(x0, dtype=np.float64), fix, warm=warm) if warm is not None
               else self.qp.solve_batch(np.asarray(x0, dtype=np.float64), fix))
        bad = np.flatnonzero(res['status'] > 1)
        if bad.size:
            # MAXITER / NUMERICAL nodes are surfaced, never silently treated as solved
            raise RuntimeError('QP solver did not converge on %d of %d nodes (status %s, first node %d)'
                               % (bad.size, fix.shape[0], sorted(set(res['status'][bad].tolist())), bad[0]))
        weak = res.get('weak')
        unc = res.get('uncertified')
        if unc is not None and np.any(unc):
            # pruned on the collapse of tau alone -- no ray verified, even loosely (HMPC_ITERS_UNCERTIFIED): said aloud, the
            # reference's solver would have stated infeasibility with a certificate (bounded_qp.py:216-228)
            import warnings
            warnings.warn('%d of %d nodes were declared infeasible WITHOUT a certificate (collapse of tau alone): a subtree pruned on '
                          'that is not proved infeasible' % (int(np.sum(unc)), fix.shape[0]), RuntimeWarning, stacklevel=3)
        sols = [SubproblemSolution.from_rows(self.layout, fix[b], res['obj'][b], res['dual_obj'][b],
                                             res['status'][b], res['primal'][b], res['dual'][b],
                                             weak=weak is not None and weak[b])
                for b in range(fix.shape[0])]
        # what a child of this node may be handed: the record of an optimal, polished (exactly complementary) node
        polished = res.get('polished')
        for b, sol in enumerate(sols):
            if res['status'][b] == 0 and polished is not None and polished[b]:
                sol.active_set = (res['primal'][b], res['dual'][b])
        return sols, res['time']

    def _solve_subproblem(self, identifier, x0, active_set=None):
        '''
        Solves the QP relaxation of one node (controller.py:229-271).
        Returns (SubproblemSolution, solve time).
        '''
        sols, t = self.solve_frontier([identifier], x0)
        return sols[0], t

    # ------------------------------------------------------------------
    # MIQP
    # ------------------------------------------------------------------

    def feedforward(self, x0, gurobi_params={}, search_rule=best_first, branch_rule=branch_in_time, **kwargs):
        '''
        Solves the mixed integer program by branch and bound (controller.py:329-393).

        ``gurobi_params`` is accepted for drop-in compatibility and ignored
        (there is no Gurobi); ``handdown=True`` hands every parent's active set to
        its children (the reference's optional simplex-basis hand-down,
        controller.py:260-264 -- see ``hmpc_warm`` in include/hmpc.h; default: the
        controller's ``solver_params['handdown']``, off); extra keyword arguments go to
        ``branch_and_bound`` (``tol``, ``warm_start``, ``printing_period``,
        ``frontier_width``, ``stats`` ...).  ``speculation_depth=k`` solves, together
        with every selected node, its descendants down to k more binaries
        (2 + 4 + ... + 2^k nodes per selected node) in the same kernel launch;
        the search consumes them only if it gets there, so its result is
        unchanged and only the number of launches per MPC step drops
        (the serial dive becomes a GPU-sized batch).

        Returns
        -------
        PrimalSolution or None, list of Node (leaves), int (QP solves), float (solver time)
        '''
        x0 = np.asarray(x0, dtype=np.float64)

        def unpack(solution, share):
            objective, binary_feasible = solution.objective_and_feasibility()
            return objective, binary_feasible, share, solution

        def solver(identifier, cutoff, extra):
            solution, solve_time = self._solve_subproblem(identifier, x0)
            return unpack(solution, solve_time)

        handdown = bool(kwargs.pop('handdown', self.handdown))

        def batch_solver(nodes, cutoff):
            if handdown:
                sets = [n.extra.active_set if n.extra is not None else None for n in nodes]
                sols, t = self.solve_frontier([n.identifier for n in nodes], x0, active_sets=sets)
            else:
                sols, t = self.solve_frontier([n.identifier for n in nodes], x0)
            return [unpack(s, t / len(sols)) for s in sols]

        def brancher(parent):
            return self._brancher(parent, branch_rule)

        depth = int(kwargs.pop('speculation_depth', 0))
        n_binaries = self.T * self.mld.nub

        def speculation(identifier):
            level, out = [identifier], []
            for _ in range(depth):
                level = [{**ident, **branch} for ident in level if len(ident) < n_binaries
                         for branch in branch_rule(ident, self.mld.nub)]
                out.extend(level)
            return out

        incumbent, leaves, qp_solves, solver_time = branch_and_bound(
            solver, search_rule, brancher, batch_solver=batch_solver,
            speculation=speculation if depth > 0 else None, **kwargs)
        if incumbent is None:
            return None, leaves, qp_solves, solver_time
        return incumbent.extra.primal, leaves, qp_solves, solver_time

    def _brancher(self, parent, branch_rule):
        '''
        Children of a solved node (controller.py:395-429): each child inherits
        the parent's dual solution, and its lower bound is the parent's plus
        the parent multiplier of the bound that the branch tightens.
        '''
        children = []
        dual = parent.extra.dual
        for branch in branch_rule(parent.identifier, self.mld.nub):
            lb = parent.lb
            for (t, i), v in branch.items():
                lb += dual.variables['nu_lb' if v == 1 else 'nu_ub'][t][i]
            child = Node({**parent.identifier, **branch}, lb, SubproblemSolution(None, dual, parent.extra.active_set))
            children.append(child)
        return children

    def feedback(self, x0, warm_start=None, e0=None, **kwargs):
        '''
        One closed-loop step (NOT in the reference API; it packages the loop
        body of notebooks/cart_pole_with_walls/plot_trajectory.py:19-45):
        solve from x0, take the first input, build the warm start for the
        next step assuming the model error ``e0`` (zero if None).

        Returns
        -------
        u0 : np.array (nu,) or None if the MIQP is infeasible
        warm_start : list of Node for the next call (None if infeasible)
        info : dict with 'solution', 'leaves', 'qp_solves', 'solver_time', 'x1'
        '''
        kwargs.setdefault('printing_period', None)
        solution, leaves, qp_solves, solver_time = self.feedforward(x0, warm_start=warm_start, **kwargs)
        info = dict(solution=solution, leaves=leaves, qp_solves=qp_solves, solver_time=solver_time, x1=None)
        if solution is None:
            return None, None, info
        uc0, ub0 = solution.variables['uc'][0], solution.variables['ub'][0]
        e0 = np.zeros(self.mld.nx) if e0 is None else np.asarray(e0, dtype=np.float64)
        next_ws = self.construct_warm_start(leaves, np.asarray(x0, dtype=np.float64), uc0, ub0, e0)[0]
        info['x1'] = solution.variables['x'][1] + e0
        return np.concatenate((uc0, ub0)), next_ws, info

    # ------------------------------------------------------------------
    # warm start
    # ------------------------------------------------------------------

    def _construct_warm_start_interstep(self, leaves, x0, uc0, ub0):
        '''
        Part of the warm start that does not need the model error
        (controller.py:431-501): drop the leaves that disagree with the
        applied binaries, shift identifier and multipliers one step back.
        '''
        u0 = np.concatenate((uc0, ub0))
        gc.disable()
        tic = time()
        warm_start = []
        for leaf in leaves:
            if not self._retain_leaf(leaf.identifier, ub0):
                continue
            shifted_identifier = {(t - 1, i): v for (t, i), v in leaf.identifier.items() if t > 0}
            old = leaf.extra.dual.variables
            new = self._shift_dual_variables(old)
            objective = leaf.extra.dual.objective + self._pi_sum(leaf.identifier, old, new, x0, u0)
            if getattr(leaf.extra.dual, 'weak', False):
                objective = -np.inf        # a ray that is no proof to tolerance is not carried over: the leaf reopens below
            warm_start.append(Node(shifted_identifier, leaf.lb, SubproblemSolution(None, DualSolution(new, objective))))
        toc = time() - tic
        gc.enable()
        return warm_start, toc

    def construct_warm_start(self, leaves, x0, uc0, ub0, e0):
        '''
        Warm start for the MIQP of the next time step (controller.py:503-564).

        leaves : leaves of the tree that proved optimality at this step
        x0 : state this step was solved from
        uc0, ub0 : applied continuous / binary inputs
        e0 : model error  x1 - A x0 - B u0

        Returns  list of Node, run-time construction time, inter-step time.
        '''
        warm_start, interstep_time = self._construct_warm_start_interstep(leaves, x0, uc0, ub0)
        gc.disable()
        tic = time()
        for node in warm_start:
            dual = node.extra.dual
            dual.objective = max(dual.objective - dual.variables['lam'][0].dot(e0), 0)
            if not np.isinf(node.lb):
                node.lb = dual.objective            # feasible before: shifted dual bound
            elif dual.objective <= 0.:
                node.lb = 0.                        # was infeasible, proof no longer holds
                node.extra.dual = None
        toc = time() - tic
        gc.enable()
        return warm_start, toc, interstep_time

    @staticmethod
    def _retain_leaf(identifier, ub0):
        '''True if the fixings of the leaf at time 0 agree with the applied binaries (controller.py:615-633).'''
        return all(v == ub0[i] for (t, i), v in identifier.items() if t == 0)

    def _shift_dual_variables(self, variables):
        '''
        Dual feasible point of the problem one step later (controller.py:635-666):
        drop time 0, pad with zeros; the last mu and rho are mapped through the
        precomputed updates so that stationarity still holds at the new last stage.
        '''
        shifted = {}
        for k in ('lam', 'nu_lb', 'nu_ub', 'sigma'):
            shifted[k] = variables[k][1:] + [np.zeros(variables[k][-1].shape)]
        for k in ('mu', 'rho'):
            shifted[k] = variables[k][1:-1] + [self._update[k].dot(variables[k][-1]),
                                               np.zeros(variables[k][-1].shape)]
        return shifted

    def _pi_sum(self, identifier, variables, shifted_variables, x0, u0):
        '''
        Change of the dual objective caused by the shift, without the term
        that needs the model error (controller.py:668-721).
        '''
        mld, T = self.mld, self.T
        sq = lambda v: v.dot(v)
        Qx0, Ru0 = self.Q.dot(x0), self.R.dot(u0)
        ub_lb, ub_ub = self._get_bound_binaries(identifier)
        Vu0 = mld.V.dot(u0)

        total = -sq(Qx0) - sq(Ru0)
        total += sq(.5 * variables['rho'][0] - Qx0) + sq(.5 * variables['sigma'][0] - Ru0)
        total -= (mld.F.dot(x0) + mld.G.dot(u0) - mld.h).dot(variables['mu'][0])
        total -= (ub_lb[0] - Vu0).dot(variables['nu_lb'][0])
        total -= (Vu0 - ub_ub[0]).dot(variables['nu_ub'][0])
        total += .25 * sq(variables['rho'][T]) - .25 * sq(shifted_variables['rho'][T - 1])
        total += self.h_Tm1.dot(variables['mu'][T - 1]) - mld.h.dot(shifted_variables['mu'][T - 2])
        return total

    def shift_binary_solution(self, ub):
        return np.vstack((ub[1:], np.zeros(self.mld.nub)))
