"""Many hybrid-MPC problems advanced in lockstep on one GPU.

The reference benchmarks its controller with a closed-loop Monte-Carlo study
(``notebooks/cart_pole_with_walls/statistical_analysis.py:93-207``): 100
simulations x 50 steps, one branch and bound at a time, one QP at a time.  The
simulations are independent, which is the natural data-parallel axis for a GPU
(SURVEY.md 8e/8f): here K branch-and-bound searches run side by side and every
round hands the candidates of ALL of them -- each node with its own initial
state -- to one kernel launch.

Semantics per instance are those of ``branch_and_bound`` + ``_brancher`` +
``construct_warm_start`` of this package (and therefore of the reference,
``branch_and_bound.py:408-499``, ``controller.py:395-429, 431-721``); only the
representation changes: the tree of an instance is a set of arrays (fixing
vectors, bounds, flat dual rows as the C ABI returns them) instead of Python
objects, and the warm-start shift is one vectorised pass over all leaves.
``tests/test_batched.py`` checks node for node that both forms agree.
"""
from time import perf_counter

import numpy as np


class NodeArrays(object):
    """Leaves of one branch-and-bound tree, in list order (the order matters for tie-breaking).

    fix   : int8 (N, T*nub)   -1 free / 0 / 1 (a chronological prefix under branch_in_time)
    lb    : float64 (N,)      lower bound (+inf: proved infeasible)
    dual  : float64 (N, n_dual) multipliers the node carries: its own if solved, else its parent's
    dobj  : float64 (N,)      dual objective that goes with ``dual``
    has_dual : bool (N,)      False for the root and for reopened warm-start nodes
    solved: bool (N,)
    """

    def __init__(self, fix, lb, dual, dobj, has_dual, solved=None):
        self.fix, self.lb, self.dual, self.dobj, self.has_dual = fix, lb, dual, dobj, has_dual
        self.solved = np.zeros(len(lb), dtype=bool) if solved is None else solved

    def __len__(self):
        return len(self.lb)

    @staticmethod
    def root(nfix, n_dual):
        return NodeArrays(np.full((1, nfix), -1, dtype=np.int8), np.array([-np.inf]), np.zeros((1, n_dual)),
                          np.zeros(1), np.zeros(1, dtype=bool))

    def identifiers(self, nub):
        return [{(k // nub, k % nub): float(v) for k, v in enumerate(row) if v >= 0} for row in self.fix]


class _Tree(object):
    """Growing arrays of one instance during the search."""

    def __init__(self, nodes):
        n = len(nodes)
        cap = max(64, 4 * n)
        self.n = n
        self.fix = np.empty((cap, nodes.fix.shape[1]), dtype=np.int8); self.fix[:n] = nodes.fix
        self.lb = np.empty(cap); self.lb[:n] = nodes.lb
        self.dual = np.empty((cap, nodes.dual.shape[1])); self.dual[:n] = nodes.dual
        self.dobj = np.empty(cap); self.dobj[:n] = nodes.dobj
        self.has_dual = np.zeros(cap, dtype=bool); self.has_dual[:n] = nodes.has_dual
        self.solved = np.zeros(cap, dtype=bool)
        self.alive = np.zeros(cap, dtype=bool); self.alive[:n] = True
        self.ub = np.inf
        self.incumbent = -1
        self.primal = None
        self.solves = 0

    def _grow(self, extra):
        if self.n + extra <= len(self.lb):
            return
        cap = 2 * (self.n + extra)
        for name in ('fix', 'lb', 'dual', 'dobj', 'has_dual', 'solved', 'alive'):
            old = getattr(self, name)
            new = np.zeros((cap,) + old.shape[1:], dtype=old.dtype)
            new[:self.n] = old[:self.n]
            setattr(self, name, new)

    def candidates(self, tol, width):
        idx = np.flatnonzero(self.alive[:self.n] & (self.lb[:self.n] < self.ub - tol))
        if idx.size > width:                      # repeated "argmin, first wins" == stable sort
            idx = idx[np.argsort(self.lb[idx], kind='stable')[:width]]
        elif idx.size > 1:
            idx = idx[np.argsort(self.lb[idx], kind='stable')]
        return idx

    def leaves(self):
        keep = np.flatnonzero(self.alive[:self.n])
        return NodeArrays(self.fix[keep].copy(), self.lb[keep].copy(), self.dual[keep].copy(), self.dobj[keep].copy(),
                          self.has_dual[keep].copy(), self.solved[keep].copy())


class BatchedMPC(object):
    """K independent hybrid-MPC searches per call on the controller's GPU backend."""

    def __init__(self, controller):
        self.c = controller
        lay = controller.layout
        self.T, self.nub, self.nx, self.nu, self.nuc = lay.T, lay.nub, lay.nx, lay.nu, lay.nuc
        self.nfix = self.T * self.nub
        cut = lay.dual_slices()
        self.o_lb = cut['nu_lb'][0].start
        self.o_ub = cut['nu_ub'][0].start
        self.cut = cut
        self.lay = lay
        # the GPU backend shifts the leaves of all trees in one kernel launch (hmpc_shift_batch, csrc/hmpc_shift.hip);
        # a backend without it (the CPU oracle used by the tests) takes the numpy form below
        self.device_shift = hasattr(controller.qp, 'shift_batch')
        if self.device_shift:
            controller.qp.set_shift_maps(controller._update['mu'], controller._update['rho'], controller.mld.V)

    # ------------------------------------------------------------------
    def feedforward_many(self, x0s, warm_starts=None, frontier_width=8, tol=0.):
        """Solves K MIQPs.  x0s: (K, nx).  warm_starts: list of NodeArrays (or None entries).

        Returns a list of dicts: objective (inf if infeasible), ub (T, nub) or None, x (T+1, nx),
        uc (T, nuc), leaves (NodeArrays), solves, and the total solver time of the call in 'time'.
        """
        x0s = np.atleast_2d(np.asarray(x0s, dtype=np.float64))
        K = x0s.shape[0]
        trees = []
        for k in range(K):
            ws = None if warm_starts is None else warm_starts[k]
            trees.append(_Tree(NodeArrays.root(self.nfix, self.lay.n_dual) if ws is None else ws))
        t_solver = 0.
        rounds = 0
        while True:
            picks = [tr.candidates(tol, frontier_width) for tr in trees]
            total = sum(p.size for p in picks)
            if total == 0:
                break
            fix = np.concatenate([tr.fix[p] for tr, p in zip(trees, picks) if p.size])
            x0 = np.concatenate([np.repeat(x0s[k:k + 1], p.size, axis=0) for k, p in enumerate(picks) if p.size])
            res = self.c.qp.solve_batch(x0, fix)
            t_solver += res['time']
            rounds += 1
            if np.any(res['status'] > 1):
                import os
                if os.environ.get('HMPC_DUMP_FAIL'):
                    bad = np.flatnonzero(res['status'] > 1)
                    np.savez(os.environ['HMPC_DUMP_FAIL'], x0=x0[bad], fix=fix[bad], status=res['status'][bad], iters=res['iters'][bad])
                raise RuntimeError('QP solver did not converge on %d nodes' % int((res['status'] > 1).sum()))
            o = 0
            for tr, p in zip(trees, picks):
                for j, node in enumerate(p):
                    self._consume(tr, node, res, o + j, tol)
                o += p.size
        out = []
        for tr in trees:
            d = dict(objective=tr.ub, ub=None, x=None, uc=None, leaves=tr.leaves(), solves=tr.solves, time=t_solver,
                     rounds=rounds)
            if tr.incumbent >= 0:
                row = tr.primal
                x = row[:(self.T + 1) * self.nx].reshape(self.T + 1, self.nx)
                u = row[(self.T + 1) * self.nx:].reshape(self.T, self.nu)
                d.update(x=x, uc=u[:, :self.nuc].copy(), ub=u[:, self.nuc:].copy())
            out.append(d)
        return out

    def _consume(self, tr, node, res, b, tol):
        """Prune / incumbent / branch for one solved node (branch_and_bound.py:476-489)."""
        tr.solves += 1
        obj = res['obj'][b]
        tr.lb[node] = obj
        tr.solved[node] = True
        tr.dual[node] = res['dual'][b]
        # an infeasibility ray that is no proof to tolerance (HMPC_ITERS_WEAK) prunes its node at this step only: with a
        # dual objective of -inf the shift reopens the leaf whatever the model error (controller.py:555-558)
        tr.dobj[node] = -np.inf if res['weak'][b] else res['dual_obj'][b]
        tr.has_dual[node] = True
        cutoff = tr.ub - tol
        if obj >= cutoff:
            return
        d = int((tr.fix[node] >= 0).sum())
        if d == self.nfix:                         # binary feasible: every binary fixed
            tr.ub, tr.incumbent, tr.primal = obj, node, res['primal'][b].copy()
            return
        # branch on the next binary in time (controller.py:13-44); child bound = parent bound + the
        # parent multiplier of the bound that the branch tightens (controller.py:419-422)
        tr._grow(2)
        for v in (0, 1):
            c = tr.n
            tr.fix[c] = tr.fix[node]
            tr.fix[c, d] = v
            tr.lb[c] = obj + (tr.dual[node, self.o_lb + d] if v == 1 else tr.dual[node, self.o_ub + d])
            tr.dual[c] = tr.dual[node]
            tr.dobj[c] = tr.dobj[node]
            tr.has_dual[c] = True
            tr.solved[c] = False
            tr.alive[c] = True
            tr.n += 1
        tr.alive[node] = False

    # ------------------------------------------------------------------
    def construct_warm_start(self, leaves, x0, uc0, ub0, e0):
        """Vectorised node shift (controller.py:431-564) for all leaves of one tree."""
        c, T, nub, nx, cut = self.c, self.T, self.nub, self.nx, self.cut
        mld = c.mld
        u0 = np.concatenate((uc0, ub0))
        first = leaves.fix[:, :nub]
        keep = np.all((first < 0) | (first == np.rint(ub0).astype(np.int8)), axis=1)   # _retain_leaf
        fix, lb, dual, dobj = leaves.fix[keep], leaves.lb[keep].copy(), leaves.dual[keep], leaves.dobj[keep]
        has_dual = leaves.has_dual[keep].copy()
        n = len(lb)
        new = np.zeros_like(dual)

        def seg(name, t):
            return cut[name][t]
        # lam, nu_lb, nu_ub, sigma: drop time 0, pad with zeros
        for name, last in (('lam', T), ('nu_lb', T - 1), ('nu_ub', T - 1), ('sigma', T - 1)):
            src = slice(cut[name][1].start, cut[name][last].stop)
            dst = slice(cut[name][0].start, cut[name][last - 1].stop)
            new[:, dst] = dual[:, src]
        # mu, rho: drop time 0, map the last block through the precomputed updates, pad with zeros
        if T > 2:
            new[:, cut['mu'][0].start:cut['mu'][T - 3].stop] = dual[:, cut['mu'][1].start:cut['mu'][T - 2].stop]
        new[:, seg('mu', T - 2)] = dual[:, seg('mu', T - 1)].dot(c._update['mu'].T)
        new[:, cut['rho'][0].start:cut['rho'][T - 2].stop] = dual[:, cut['rho'][1].start:cut['rho'][T - 1].stop]
        new[:, seg('rho', T - 1)] = dual[:, seg('rho', T)].dot(c._update['rho'].T)
        # change of the dual objective (controller.py:668-721)
        Qx0, Ru0 = c.Q.dot(x0), c.R.dot(u0)
        rho0, sig0 = dual[:, seg('rho', 0)], dual[:, seg('sigma', 0)]
        pi = -Qx0.dot(Qx0) - Ru0.dot(Ru0)
        pi = pi + np.sum((.5 * rho0 - Qx0) ** 2, axis=1) + np.sum((.5 * sig0 - Ru0) ** 2, axis=1)
        pi -= dual[:, seg('mu', 0)].dot(mld.F.dot(x0) + mld.G.dot(u0) - mld.h)
        lo = np.where(first[keep] >= 0, first[keep], 0).astype(np.float64)
        hi = np.where(first[keep] >= 0, first[keep], 1).astype(np.float64)
        Vu0 = mld.V.dot(u0)
        pi -= np.sum((lo - Vu0) * dual[:, seg('nu_lb', 0)], axis=1)
        pi -= np.sum((Vu0 - hi) * dual[:, seg('nu_ub', 0)], axis=1)
        pi += .25 * np.sum(dual[:, seg('rho', T)] ** 2, axis=1) - .25 * np.sum(new[:, seg('rho', T - 1)] ** 2, axis=1)
        pi += dual[:, seg('mu', T - 1)].dot(c.h_Tm1) - new[:, seg('mu', T - 2)].dot(mld.h)
        obj = dobj + pi
        # run-time part (controller.py:541-558): model error, clipping, reopening
        obj = np.maximum(obj - new[:, seg('lam', 0)].dot(e0), 0.)
        finite = ~np.isinf(lb)
        lb[finite] = obj[finite]
        reopen = (~finite) & (obj <= 0.)
        lb[reopen] = 0.
        has_dual[reopen] = False
        new_fix = np.concatenate((fix[:, nub:], np.full((n, nub), -1, dtype=np.int8)), axis=1)
        return NodeArrays(new_fix, lb, new, obj, has_dual)

    def construct_warm_start_many(self, leaves_list, x0s, u0s, e0s):
        """Node shift for the leaves of K trees at once: one kernel launch on the GPU backend, else the numpy form per tree.

        leaves_list : list of NodeArrays;  x0s, e0s : (K, nx);  u0s : (K, nu) applied inputs (uc, ub).
        Returns a list of NodeArrays (the warm starts).
        """
        if not self.device_shift:
            return [self.construct_warm_start(lv, x0s[k], u0s[k][:self.nuc], u0s[k][self.nuc:], e0s[k])
                    for k, lv in enumerate(leaves_list)]
        sizes = [len(lv) for lv in leaves_list]
        owner = np.repeat(np.arange(len(leaves_list), dtype=np.int32), sizes)
        cat = lambda name: np.concatenate([getattr(lv, name) for lv in leaves_list])
        r = self.c.qp.shift_batch(owner, x0s, u0s, e0s, cat('fix'), cat('lb'), cat('dual'), cat('dobj'))
        has_dual = cat('has_dual') & ~r['reopened']
        out, o = [], 0
        for n in sizes:
            k = np.flatnonzero(r['keep'][o:o + n]) + o
            out.append(NodeArrays(r['fix'][k], r['lb'][k], r['dual'][k], r['dual_obj'][k], has_dual[k]))
            o += n
        return out

    # ------------------------------------------------------------------
    def closed_loop(self, x0, n_steps, e_sd=0., seeds=(0,), x_max=None, frontier_width=8, cold_too=False, log=None, errors=None):
        """Closed-loop Monte-Carlo study in the shape of statistical_analysis.py:93-207.

        One simulation per seed, all advanced in lockstep.  The disturbance of simulation i at step t is
        ``e_sd * RandomState(i).randn(nx) * x_max`` -- the stream of ``np.random.seed(i)`` that the
        reference draws from (statistical_analysis.py:73,176).  A simulation whose MIQP becomes infeasible
        stops (the reference discards it, :99-108).  ``errors`` (len(seeds), n_steps, nx), if given, replaces the
        random stream: the disturbances applied in the reference's published runs
        (``notebooks/cart_pole_with_walls/data/errors_sd_*.npy``) can be replayed step by step.

        Returns dict: nodes_ws, nodes_cs (per sim, per step), len_ws, costs, alive steps, wall time, steps/s.
        """
        K = len(seeds)
        rngs = [np.random.RandomState(s) for s in seeds]
        x_max = np.ones(self.nx) if x_max is None else np.asarray(x_max, dtype=np.float64)
        xs = np.repeat(np.asarray(x0, dtype=np.float64)[None], K, axis=0)
        ws = [None] * K
        active = list(range(K))
        stats = dict(nodes_ws=[[] for _ in range(K)], nodes_cs=[[] for _ in range(K)], len_ws=[[] for _ in range(K)],
                     costs=[[] for _ in range(K)], errors=[[] for _ in range(K)], reopened=[[] for _ in range(K)], cost_mismatches=[])
        tic = perf_counter()
        steps_done = 0
        for t in range(n_steps):
            if not active:
                break
            cold = self.feedforward_many(xs[active], None, frontier_width) if cold_too else None
            warm = self.feedforward_many(xs[active], [ws[k] for k in active], frontier_width)
            still, shifted = [], []
            for j, k in enumerate(active):
                r = warm[j]
                if cold is not None:
                    stats['nodes_cs'][k].append(cold[j]['solves'])
                    # the reference asserts np.isclose(cost_cs, cost_ws) (statistical_analysis.py:171); here a
                    # disagreement is counted and reported instead of ending the study
                    if not np.isclose(r['objective'], cold[j]['objective'], rtol=1e-5, atol=1e-8):
                        stats['cost_mismatches'].append((seeds[k], t, cold[j]['objective'], r['objective']))
                stats['nodes_ws'][k].append(r['solves'])
                if not np.isfinite(r['objective']):
                    continue                        # infeasible: the simulation ends here
                e_t = e_sd * rngs[k].randn(self.nx) * x_max if errors is None else np.asarray(errors[k][t], dtype=np.float64)
                shifted.append((j, k, e_t))
            if shifted:
                new_ws = self.construct_warm_start_many(
                    [warm[j]['leaves'] for j, _, _ in shifted], np.array([xs[k] for _, k, _ in shifted]),
                    np.array([np.concatenate((warm[j]['uc'][0], warm[j]['ub'][0])) for j, _, _ in shifted]),
                    np.array([e for _, _, e in shifted]))
            for (j, k, e_t), w in zip(shifted, new_ws if shifted else []):
                r = warm[j]
                ws[k] = w
                stats['len_ws'][k].append(len(ws[k]))
                stats['reopened'][k].append(int((~ws[k].has_dual).sum()))   # infeasibility proofs lost in the shift
                stats['costs'][k].append(r['objective'])
                stats['errors'][k].append(e_t)
                if log is not None:
                    cs = '(cs: {}, {:.3f}) '.format(cold[j]['solves'], cold[j]['time']) if cold is not None else ''
                    log.write('sim %d Time step %d %s(ws: %d, %.3f) (ws info: %d) (e: %.3f, %s)\n'
                              % (seeds[k], t, cs, r['solves'], r['time'], len(ws[k]), np.linalg.norm(e_t), e_t))
                xs[k] = r['x'][1] + e_t
                still.append(k)
                steps_done += 1
            active = still
        wall = perf_counter() - tic
        stats.update(wall=wall, steps=steps_done, steps_per_sec=steps_done / wall if wall > 0 else 0., survivors=len(active))
        return stats
