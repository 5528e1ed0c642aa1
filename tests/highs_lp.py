"""The batched LP call of ``terminal_set`` on top of HiGHS (scipy) -- an independent solver the LP oracle and the HIP
LP kernel are checked against, and the one the committed fixtures were generated with.  TEST INFRASTRUCTURE."""
import numpy as np
from scipy.optimize import linprog


def lp_solve_batch(A, c, b, relax=None, **_):
    A = np.asarray(A, dtype=float)
    m, n = A.shape
    c = np.asarray(c, dtype=float)
    b = np.asarray(b, dtype=float)
    B = max(c.shape[0] if c.ndim == 2 else 1, b.shape[0] if b.ndim == 2 else 1, 0 if relax is None else len(relax))
    obj = np.full(B, np.nan); x = np.full((B, n), np.nan); z = np.zeros((B, m)); status = np.zeros(B, dtype=np.int32)
    for k in range(B):
        ck = c[k] if c.ndim == 2 else c
        bk = np.array(b[k] if b.ndim == 2 else b)
        if relax is not None:
            bk[relax[k]] += 1.
        res = linprog(-ck, A_ub=A, b_ub=bk, bounds=(None, None), method='highs')
        if res.status == 0:
            obj[k], x[k], z[k] = -res.fun, res.x, -res.ineqlin.marginals
        else:
            status[k] = {2: 1, 3: 4}.get(res.status, 3)
    return dict(obj=obj, x=x, z=z, status=status, iters=np.zeros(B, dtype=np.int32))
