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

    def __init__(self, controller, K):
        qp = controller.qp
        if not hasattr(qp, 'handle'):
            raise RuntimeError('FleetMPC needs the HIP backend (the product path has no CPU fallback).')
        self.c, self.qp, self.K = controller, qp, int(K)
        qp.set_shift_maps(controller._update['mu'], controller._update['rho'], controller.mld.V)
        self.nx, self.nu = controller.mld.nx, controller.mld.nu
        self._f = ctypes.c_void_p()
        qp._check(qp.lib.hmpc_fleet_create(qp.handle, self.K, ctypes.byref(self._f)))

    def __del__(self):
        f = getattr(self, '_f', None)
        if f:
            self.qp.lib.hmpc_fleet_destroy(f)
            self._f = None

    def reset(self, k=-1):
        self.qp._check(self.qp.lib.hmpc_fleet_reset(self._f, int(k)))

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
        r, n = ctypes.c_int64(), ctypes.c_int64()
        self.qp._check(self.qp.lib.hmpc_fleet_stats(self._f, ctypes.byref(r), ctypes.byref(n)))
        return dict(rounds=r.value, launched=n.value)

    def closed_loop(self, x0, n_steps, errors, frontier_width=8, speculation=0, cold_speculation=0):
        """K closed loops from the same x0 under prescribed model errors (K, n_steps, nx) -- the shape of
        ``BatchedMPC.closed_loop(errors=...)``.  Returns dict: costs, nodes_ws, len_ws, reopened (K, n_steps; NaN / 0
        after a loop has ended), wall, steps, steps_per_sec."""
        K = self.K
        errors = np.asarray(errors, dtype=np.float64)
        xs = np.repeat(np.asarray(x0, dtype=np.float64)[None], K, axis=0)
        st = dict(costs=np.full((K, n_steps), np.nan), nodes_ws=np.zeros((K, n_steps), dtype=np.int64),
                  len_ws=np.zeros((K, n_steps), dtype=np.int64), reopened=np.zeros((K, n_steps), dtype=np.int64))
        self.reset()
        steps = 0
        tic = perf_counter()
        for t in range(n_steps):
            r = self.solve(xs, frontier_width, speculation=cold_speculation if t == 0 else speculation)
            ok = np.isfinite(r['cost'])
            cover, reopened = self.shift(errors[:, t])
            st['costs'][:, t] = r['cost']
            st['nodes_ws'][:, t] = r['solves']
            st['len_ws'][:, t] = cover
            st['reopened'][:, t] = reopened
            xs = np.where(ok[:, None], r['x1'] + errors[:, t], xs)
            steps += int(ok.sum())
        wall = perf_counter() - tic
        st.update(wall=wall, steps=steps, steps_per_sec=steps / wall if wall > 0 else 0.)
        return st
