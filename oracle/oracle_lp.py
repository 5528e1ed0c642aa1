"""ctypes front end of the CPU oracle for the facet LPs (oracle/dense_lp.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this
module.  Same call as the product's ``qp_backend.lp_solve_batch`` so that the offline terminal ingredients
(``terminal_set.mcais / remove_redundant_inequalities / update_mu``) can be run on a machine without a GPU and
the GPU results checked against it.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'liboracle_lp%s.so' % os.environ.get('ORACLE_LIBRARY_SUFFIX', ''))
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(['make', '-s', '-C', HERE])
        lib = ctypes.CDLL(LIB)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        lib.oracle_lp_batch.restype = ctypes.c_int
        lib.oracle_lp_batch.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, dp, ctypes.c_int, ip, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, dp, dp, dp, ip, ip]
        _lib = lib
    return _lib


def lp_solve_batch(A, c, b, relax=None, tol=1e-9, max_iter=100, threads=1):
    """maximise c_k'x s.t. A x <= b_k (+1 on row relax[k]); c: [n] or [B, n], b: [m] or [B, m].

    Returns dict(obj[B], x[B, n], z[B, m], status[B], iters[B]); status 0 optimal, 1 empty set, 4 unbounded.
    """
    lib = _load()
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    c = np.ascontiguousarray(c, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    B = max(c.shape[0] if c.ndim == 2 else 1, b.shape[0] if b.ndim == 2 else 1, 0 if relax is None else len(relax))
    cs = n if c.ndim == 2 else 0
    bs = m if b.ndim == 2 else 0
    if (c.ndim == 2 and c.shape != (B, n)) or (c.ndim == 1 and c.size != n) or (b.ndim == 2 and b.shape != (B, m)) or (b.ndim == 1 and b.size != m):
        raise ValueError('lp_solve_batch: inconsistent sizes')
    rl = None
    if relax is not None:
        rl = np.ascontiguousarray(relax, dtype=np.int32)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    obj = np.empty(B); x = np.empty((B, n)); z = np.empty((B, m))
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
    d = lambda a: a.ctypes.data_as(dp)
    rc = lib.oracle_lp_batch(n, m, d(A), d(c), cs, d(b), bs, rl.ctypes.data_as(ip) if rl is not None else None, B,
                             tol, max_iter, threads, d(obj), d(x), d(z), status.ctypes.data_as(ip), iters.ctypes.data_as(ip))
    if rc != 0:
        raise ValueError('oracle_lp_batch: bad arguments')
    return dict(obj=obj, x=x, z=z, status=status, iters=iters)
