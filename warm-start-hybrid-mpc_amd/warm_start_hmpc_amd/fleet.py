"""K closed loops in lockstep through the C ABI's fleet driver (``hmpc_fleet_*``, ``csrc/hmpc_fleet.hip``).

The reference runs its closed-loop study one simulation, one step, one node at a time
(``notebooks/cart_pole_with_walls/statistical_analysis.py:93-196``).  ``batched.BatchedMPC`` runs K simulations
side by side but keeps the trees in numpy and the bookkeeping in Python; here the trees live behind the library
handle -- topology and bounds in C++, every multiplier row in HBM -- and one call advances all loops by one step.
Same per-tree semantics as ``BatchedMPC.feedforward_many`` / ``construct_warm_start_many``
(``tests/test_fleet.py`` compares the two step by step).
"""
import ctypes
from time import perf_counter

import numpy as np


class FleetMPC(object):

    def __init__(self, controller, K, handdown=True):
        """handdown: children solved in a later round than their parent receive the parent's record (``hmpc_warm``: the
        parent's active set is tried before the first interior-point iteration; the reference hands down the simplex
        basis, controller.py:260-264)."""
        qp = controller.qp
        if not hasattr(qp, 'handle'):
            raise RuntimeError('FleetMPC needs the HIP backend (the product path has no CPU fallback).')
        self.c, self.qp, self.K = controller, qp, int(K)
        self.handdown = bool(handdown)
        qp.set_shift_maps(controller._update['mu'], controller._update['rho'], controller.mld.V)
        self.nx, self.nu = controller.mld.nx, controller.mld.nu
        self._f = ctypes.c_void_p()
        qp._check(qp.lib.hmpc_fleet_create(qp.handle, self.K, ctypes.byref(self._f)))
        qp._check(qp.lib.hmpc_fleet_handdown(self._f, int(bool(handdown)), None))

    def __del__(self):
        f = getattr(self, '_f', None)
        if f:
            self.qp.lib.hmpc_fleet_destroy(f)
            self._f = None

    def reset(self, k=-1):
        self.qp._check(self.qp.lib.hmpc_fleet_reset(self._f, int(k)))

    def stop(self, k):
        """Loop k has ended: its tree is dropped and it takes no part in further steps (no launches, no pool rows) until
        ``reset`` makes it cold again."""
        self.qp._check(self.qp.lib.hmpc_fleet_stop(self._f, int(k)))

    def rows(self):
        """(rows of the multiplier pools in use, rows allocated).  Rows nobody references are reclaimed: ``shift``
        compacts, and ``solve`` starts the pools from zero when every tree is cold."""
        used, cap = ctypes.c_int64(), ctypes.c_int64()
        self.qp._check(self.qp.lib.hmpc_fleet_rows(self._f, ctypes.byref(used), ctypes.byref(cap)))
        return used.value, cap.value

    def solve(self, x0s, frontier_width=8, tol=0., speculation=0):
        """One MIQP per loop from x0s (K, nx), warm-started from the loop's tree.  Returns dict of arrays:
        cost (K,), u0 (K, nu), x1 (K, nx) -- the model's next state --, solves, leaves (K,).
        ``speculation=k``: descendants through the next k binaries ride in the launch of every node that has to be
        solved (same results, fewer launches per step; pays for few loops)."""
        x0s = np.ascontiguousarray(x0s, dtype=np.float64)
        assert x0s.shape == (self.K, self.nx)
        out = dict(cost=np.empty(self.K), u0=np.empty((self.K, self.nu)), x1=np.empty((self.K, self.nx)),
                   solves=np.empty(self.K, dtype=np.int32), leaves=np.empty(self.K, dtype=np.int32))
        self.qp._check(self.qp.lib.hmpc_fleet_solve(self._f, x0s.ctypes.data, int(frontier_width), int(speculation), float(tol), out['cost'].ctypes.data,
                                                    out['u0'].ctypes.data, out['x1'].ctypes.data, out['solves'].ctypes.data,
                                                    out['leaves'].ctypes.data))
        return out

    def shift(self, e0s):
        """Turns every tree into the warm start of the next step given the model errors e0s (K, nx).
        Returns (cover sizes, reopened leaves), int32 (K,)."""
        e0s = np.ascontiguousarray(e0s, dtype=np.float64)
        assert e0s.shape == (self.K, self.nx)
        cover, reopened = np.empty(self.K, dtype=np.int32), np.empty(self.K, dtype=np.int32)
        self.qp._check(self.qp.lib.hmpc_fleet_shift(self._f, e0s.ctypes.data, cover.ctypes.data, reopened.ctypes.data))
        return cover, reopened

    def stats(self):
        r, n, v = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.qp._check(self.qp.lib.hmpc_fleet_stats(self._f, ctypes.byref(r), ctypes.byref(n)))
        self.qp._check(self.qp.lib.hmpc_fleet_handdown(self._f, -1, ctypes.byref(v)))
        t = (ctypes.c_double * 5)()
        self.qp._check(self.qp.lib.hmpc_fleet_timing(self._f, t))
        u, w = ctypes.c_int64(), ctypes.c_int64()
        self.qp._check(self.qp.lib.hmpc_fleet_uncertified(self._f, ctypes.byref(u), ctypes.byref(w)))
        # (uncertified: nodes pruned on the collapse of tau alone, HMPC_ITERS_UNCERTIFIED; resting: searches whose optimum rests on one)
        return dict(rounds=r.value, launched=n.value, handed=v.value, uncertified=u.value, resting_on_uncertified=w.value,
                    seconds=dict(zip(('select', 'stage', 'device', 'consume', 'shift'), [float(x) for x in t])))

    def closed_loop(self, x0, n_steps, errors, frontier_width=8, speculation=0, cold_speculation=0, cold_frontier_width=None):
        """K closed loops from the same x0 under prescribed model errors (K, n_steps, nx) -- the shape of
        ``BatchedMPC.closed_loop(errors=...)``.  Returns dict: costs, nodes_ws, len_ws, reopened (K, n_steps; NaN / 0
        after a loop has ended), wall, wall_first_step (the cold start's share of it), steps, steps_per_sec."""
        K = self.K
        errors = np.asarray(errors, dtype=np.float64)
        xs = np.repeat(np.asarray(x0, dtype=np.float64)[None], K, axis=0)
        st = dict(costs=np.full((K, n_steps), np.nan), nodes_ws=np.zeros((K, n_steps), dtype=np.int64),
                  len_ws=np.zeros((K, n_steps), dtype=np.int64), reopened=np.zeros((K, n_steps), dtype=np.int64))
        self.reset()
        steps = 0
        tic = perf_counter()
        for t in range(n_steps):
            r = self.solve(xs, (cold_frontier_width or frontier_width) if t == 0 else frontier_width,
                           speculation=cold_speculation if t == 0 else speculation)
            ok = np.isfinite(r['cost'])
            cover, reopened = self.shift(errors[:, t])
            st['costs'][:, t] = r['cost']
            st['nodes_ws'][:, t] = r['solves']
            st['len_ws'][:, t] = cover
            st['reopened'][:, t] = reopened
            xs = np.where(ok[:, None], r['x1'] + errors[:, t], xs)
            steps += int(ok.sum())
            if t == 0:
                first = perf_counter() - tic
        wall = perf_counter() - tic
        st.update(wall=wall, wall_first_step=first if n_steps > 0 else 0., steps=steps, steps_per_sec=steps / wall if wall > 0 else 0.)
        return st


def closed_loop_study(controller, x0, errors, frontier_width=1, cold_too=True, speculation=0, log=None, sim_ids=None,
                      handdown=True):
    """The reference's closed-loop study (``notebooks/cart_pole_with_walls/statistical_analysis.py:93-196``) on the fleet
    driver: K simulations in lockstep under prescribed model errors ``errors`` (K, n_steps, nx); at every step a
    cold-started search (a second fleet, reset before every step; ``cold_too``) and a warm-started one from the same
    state, their costs compared (the reference's assertion at :171; a disagreement is recorded, not raised), the next
    warm start built from the warm search's leaves (:180-187).  A simulation ends at the step whose MIQP has no solution
    (:165-166).  ``frontier_width=1`` is the reference's node order.  The input applied between two steps is the warm
    search's u0; the reference takes uc / ub[0] from the COLD solution and x[1] from the warm one (:180-194) -- the same
    thing wherever the optimum is unique (a tie of the MIQP is the one event in which they can differ: three steps of
    the published study, tests/test_reference_replay.py).  ``handdown=False`` runs the searches as the reference's published
    runs did (no parent -> child hand-down: where multipliers are not unique a handed-down solve may return another
    optimal choice, and solve counts move by a few).

    Returns the dictionary of ``BatchedMPC.closed_loop``: per simulation the lists nodes_cs, nodes_ws, len_ws, reopened,
    costs (one entry per step the simulation was alive for; len_ws / reopened / costs only for steps with a solution),
    cost_mismatches [(simulation, step, cold cost, warm cost)], wall, steps, steps_per_sec, survivors.
    """
    errors = np.asarray(errors, dtype=np.float64)
    K, n_steps = errors.shape[:2]
    sim_ids = list(range(K)) if sim_ids is None else list(sim_ids)
    warm = FleetMPC(controller, K, handdown=handdown)
    cold = FleetMPC(controller, K, handdown=handdown) if cold_too else None
    xs = np.repeat(np.asarray(x0, dtype=np.float64)[None], K, axis=0)
    alive = np.ones(K, dtype=bool)
    stopped = np.zeros(K, dtype=bool)
    st = dict(nodes_ws=[[] for _ in range(K)], nodes_cs=[[] for _ in range(K)], len_ws=[[] for _ in range(K)],
              costs=[[] for _ in range(K)], reopened=[[] for _ in range(K)], cost_mismatches=[])
    steps = 0
    tic = perf_counter()
    for t in range(n_steps):
        if not alive.any():
            break
        rc = None
        if cold is not None:
            for k in range(K):                          # (an ended simulation takes no part: no launches, no pool rows)
                if alive[k]:
                    cold.reset(k)
                elif not stopped[k]:
                    cold.stop(k)
                    stopped[k] = True
            rc = cold.solve(xs, frontier_width, speculation=speculation)
        rw = warm.solve(xs, frontier_width, speculation=speculation)
        ok = alive & np.isfinite(rw['cost'])
        cover, reopened = warm.shift(errors[:, t])
        for k in np.flatnonzero(alive):
            st['nodes_ws'][k].append(int(rw['solves'][k]))
            if rc is not None:
                st['nodes_cs'][k].append(int(rc['solves'][k]))
                if not (np.isclose(rw['cost'][k], rc['cost'][k], rtol=1e-5, atol=1e-8) or
                        (np.isinf(rw['cost'][k]) and np.isinf(rc['cost'][k]))):
                    st['cost_mismatches'].append((sim_ids[k], t, float(rc['cost'][k]), float(rw['cost'][k])))
            if ok[k]:
                st['len_ws'][k].append(int(cover[k]))
                st['reopened'][k].append(int(reopened[k]))
                st['costs'][k].append(float(rw['cost'][k]))
                if log is not None:
                    cs = '(cs: %d) ' % rc['solves'][k] if rc is not None else ''
                    log.write('sim %d Time step %d %s(ws: %d) (ws info: %d) (e: %.3f, %s)\n'
                              % (sim_ids[k], t, cs, rw['solves'][k], cover[k], np.linalg.norm(errors[k, t]), errors[k, t]))
        xs = np.where(ok[:, None], rw['x1'] + errors[:, t], xs)
        steps += int(ok.sum())
        alive = ok
    wall = perf_counter() - tic
    st.update(wall=wall, steps=steps, steps_per_sec=steps / wall if wall > 0 else 0., survivors=int(alive.sum()),
              handed=warm.stats()['handed'] + (cold.stats()['handed'] if cold else 0),
              rounds=warm.stats()['rounds'] + (cold.stats()['rounds'] if cold else 0),
              launched=warm.stats()['launched'] + (cold.stats()['launched'] if cold else 0))
    return st
