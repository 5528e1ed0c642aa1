"""Diagnostic (GPU box): how far the multipliers of the binaries' bounds -- what child bounds are made of
(controller.py:419-422) -- differ between the HIP kernel and the oracle on the optimal nodes of a real tree.
    python tests/gpu_nu_diff.py"""
import numpy as np
from conftest import ROOT  # noqa: F401  (path setup)
from helpers import make_controller, real_tree_with_parents

X0 = np.array([0., 0., 1., 0.])
for name, T in (('cart_pole_with_walls', 20), ('cart_pole_with_walls', 10), ('cart_pole_one_wall', 40)):
    hip = make_controller(name, T=T, backend='hip')
    orc = make_controller(name, T=T, backend='oracle', threads=8)
    x0 = np.array([0., 0., .5, 0.]) if T == 10 else X0
    fix, _ = real_tree_with_parents(orc, x0)
    a, b = hip.qp.solve_batch(x0, fix), orc.qp.solve_batch(x0, fix)
    cut = hip.layout.dual_slices()
    lo, hi = cut['nu_lb'][0].start, cut['nu_ub'][-1].stop
    opt = (a['status'] == 0) & (b['status'] == 0)
    d = np.abs(a['dual'][opt][:, lo:hi] - b['dual'][opt][:, lo:hi])
    net = np.abs((a['dual'][opt][:, cut['nu_ub'][0].start:hi] - a['dual'][opt][:, lo:cut['nu_ub'][0].start]) -
                 (b['dual'][opt][:, cut['nu_ub'][0].start:hi] - b['dual'][opt][:, lo:cut['nu_ub'][0].start]))
    mu = np.abs(a['dual'][opt][:, cut['mu'][0].start:cut['mu'][-1].stop] - b['dual'][opt][:, cut['mu'][0].start:cut['mu'][-1].stop])
    print('%s N=%d: %d optimal nodes; |nu_lb, nu_ub| differ by at most %.3e (entries above 1e-6: %d of %d), nu_ub - nu_lb by %.3e, mu by %.3e; statuses equal %s, iterations equal on %.0f %%'
          % (name, T, opt.sum(), d.max(), (d > 1e-6).sum(), d.size, net.max(), mu.max(), np.array_equal(a['status'], b['status']),
             100 * np.mean(a['iters'] == b['iters'])))
