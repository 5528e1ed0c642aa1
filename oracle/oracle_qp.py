"""ctypes front end of the CPU oracle (oracle/hsde_qp.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py import this module.  It offers the same
``solve_batch(x0, fix)`` interface as the product's HIP backend so that the
host logic (controller, branch and bound, warm start) can be exercised on a
machine without a GPU and the GPU results can be checked against it.
"""
import ctypes
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'liboracle_qp%s.so' % os.environ.get('ORACLE_LIBRARY_SUFFIX', ''))   # (_asan: the sanitizer build, tests/test_sanitizers.py)


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def _load():
    if not os.path.exists(LIB):
        build()
    lib = ctypes.CDLL(LIB)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_solve_batch.restype = ctypes.c_int
    lib.oracle_solve_batch.argtypes = [ctypes.c_int] * 9 + [dp] * 11 + [dp, ctypes.c_int, ctypes.c_int,
                                                                       ctypes.POINTER(ctypes.c_int8),
                                                                       ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                                       dp, dp, ctypes.POINTER(ctypes.c_int32),
                                                                       dp, dp, ctypes.POINTER(ctypes.c_int),
                                                                       ctypes.POINTER(ctypes.c_int), dp, dp, ctypes.POINTER(ctypes.c_int)]
    return lib


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class OracleBatchedQP(object):
    """CPU oracle behind the backend interface of the controller."""

    @staticmethod
    def lp_solve_batch(A, c, b, relax=None, **kw):
        """The facet LPs of the offline ingredients on the LP oracle (oracle/dense_lp.c)."""
        from oracle.oracle_lp import lp_solve_batch
        return lp_solve_batch(A, c, b, relax=relax, **kw)

    def __init__(self, problem, tol=1e-8, tol_inf=1e-6, max_iter=100, threads=1, lazy_terminal=True, refine=True, polish=True, polish_tol=1e-4):
        self.lib = _load()
        c = lambda M: np.ascontiguousarray(np.atleast_2d(M), dtype=np.float64)
        self.p = {k: (c(v) if k not in ('nx', 'nu', 'nub', 'T', 'h', 'h_Tm1') else v) for k, v in problem.items()}
        self.p['h'] = np.ascontiguousarray(problem['h'], dtype=np.float64)
        self.p['h_Tm1'] = np.ascontiguousarray(problem['h_Tm1'], dtype=np.float64)
        self.tol, self.tol_inf, self.max_iter, self.threads = tol, tol_inf, max_iter, threads
        self.lazy_terminal = int(lazy_terminal)
        self.refine = int(refine)
        self.polish = int(polish)
        self.polish_tol = float(polish_tol)
        p = self.p
        self.sizes = (p['nx'], p['nu'], p['nub'], p['T'], p['h'].size, p['h_Tm1'].size,
                      p['Q'].shape[0], p['R'].shape[0], p['Q_T'].shape[0])
        nx, nu, nub, T, nc, ncL, nq, nr, nqT = self.sizes
        self.n_primal = (T + 1) * nx + T * nu
        self.n_dual = (T + 1) * nx + (T - 1) * nc + ncL + 2 * T * nub + T * nq + nqT + T * nr

    def solve_batch(self, x0, fix, warm=None):
        """warm: optional (primal rows, dual rows, index) -- per node the row of its PARENT's record in the two arrays
        (-1: none), the hand-down of hmpc_warm (include/hmpc.h)."""
        p = self.p
        fix = np.ascontiguousarray(fix, dtype=np.int8)
        B = fix.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        stride = 0 if x0.ndim == 1 else p['nx']
        out = dict(obj=np.empty(B), dual_obj=np.empty(B), status=np.empty(B, dtype=np.int32),
                   iters=np.empty(B, dtype=np.int32), primal=np.empty((B, self.n_primal)),
                   dual=np.empty((B, self.n_dual)), polished=np.zeros(B, dtype=np.int32))
        if warm is not None:
            wp = np.ascontiguousarray(warm[0], dtype=np.float64).reshape(-1, self.n_primal)
            wd = np.ascontiguousarray(warm[1], dtype=np.float64).reshape(-1, self.n_dual)
            wi = np.ascontiguousarray(warm[2], dtype=np.int32)
            assert wi.shape == (B,) and wp.shape[0] == wd.shape[0] and (wi.max(initial=-1) < wp.shape[0])
            wargs = (_d(wp), _d(wd), wi.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        else:
            wargs = (None, None, None)
        tic = time.perf_counter()
        rc = self.lib.oracle_solve_batch(
            *self.sizes, _d(p['A']), _d(p['B']), _d(p['F']), _d(p['G']), _d(p['h']),
            _d(p['F_Tm1']), _d(p['G_Tm1']), _d(p['h_Tm1']), _d(p['Q']), _d(p['R']), _d(p['Q_T']),
            _d(x0), stride, B, fix.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
            self.tol, self.tol_inf, self.max_iter, self.threads, self.lazy_terminal, self.refine, self.polish, self.polish_tol,
            *wargs,
            _d(out['obj']), _d(out['dual_obj']), out['status'].ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
            out['iters'].ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _d(out['primal']), _d(out['dual']),
            out['polished'].ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        if rc != 0:
            raise RuntimeError('oracle_solve_batch failed with code %d' % rc)
        out['time'] = time.perf_counter() - tic
        out['weak'] = (out['polished'] >> 8) & 1        # infeasible, but the ray is no proof to tolerance (HMPC_ITERS_WEAK)
        out['second'] = (out['polished'] >> 9) & 1      # the terminal-set rows were needed: the node was solved twice (lazy terminal set)
        out['uncertified'] = (out['polished'] >> 10) & 1  # weak, and pruned on the collapse of tau alone (HMPC_ITERS_UNCERTIFIED)
        out['polished'] &= 0xff
        return out
