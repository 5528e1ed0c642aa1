"""Experiment on the CPU oracle (hand-run): would a leaf of the shifted tree verify from ITS OWN shifted record?

The hand-down (hmpc_warm) tries a handed record's active set before the first interior-point iteration; inside a step a
child receives its parent's record.  The leaves a warm-started search re-solves first have no parent record in the new step --
but they carry shifted multipliers (their own if they were solved in the old step, else their parent's).  Here: closed loop on
the oracle, at every step the leaves of the shifted tree with a finite bound are solved at the next state (a) cold and (b) handed
their shifted multipliers (primal row: zeros with the fixed binaries in place -- the primal only sets the proximal centre, weight
1e-10); counted: how many verify (polished == 64), and that verified records equal the cold ones.

    python tests/cpu_self_handdown.py [steps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.batched import BatchedMPC

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
bm = BatchedMPC(ctrl)
qp = ctrl.qp
x_max = load_fixture('cart_pole_with_walls')['x_max']
T, nub, nx, nu = ctrl.T, ctrl.mld.nub, ctrl.mld.nx, ctrl.mld.nu
nuc = nu - nub
for sd in (0.001, 0.01):
    rng = np.random.RandomState(0)
    x = np.array([0., 0., 1., 0.])
    ws = None
    tot = dict(leaves=0, finite=0, verified=0, first8=0, first8_verified=0, equal=0)
    for step in range(steps):
        res = bm.feedforward_many(x[None], None if ws is None else [ws], frontier_width=8)[0]
        e0 = sd * rng.randn(nx) * x_max
        ws = bm.construct_warm_start(res['leaves'], x, res['uc'][0], res['ub'][0], e0)
        x = res['x'][1] + e0
        fin = np.flatnonzero(np.isfinite(ws.lb) & ws.has_dual)
        if not len(fin):
            continue
        fix = ws.fix[fin]
        primal = np.zeros((len(fin), qp.n_primal))
        for j in range(len(fin)):                      # fixed binaries in place (the hand-down skips a record whose binaries lie far from the node's)
            for t in range(T):
                for b in range(nub):
                    if fix[j, t * nub + b] >= 0:
                        primal[j, (T + 1) * nx + t * nu + nuc + b] = fix[j, t * nub + b]
        cold = qp.solve_batch(x, fix)
        warm = qp.solve_batch(x, fix, warm=(primal, ws.dual[fin], np.arange(len(fin), dtype=np.int32)))
        ver = warm['polished'] == 64
        assert np.array_equal(cold['status'], warm['status'])
        opt = cold['status'] == 0
        ok = np.allclose(cold['obj'][opt], warm['obj'][opt], rtol=1e-8, atol=1e-10)
        order = np.argsort(ws.lb[fin], kind='stable')[:8]
        tot['leaves'] += len(ws.lb); tot['finite'] += len(fin); tot['verified'] += int(ver.sum())
        tot['first8'] += len(order); tot['first8_verified'] += int(ver[order].sum()); tot['equal'] += int(ok)
        print('sd %.3f step %d: cover %d, finite bounds %d (optimal at the next state %d), verified from the own shifted record %d; of the 8 lowest bounds %d; objectives equal %s; iterations cold %.1f'
              % (sd, step, len(ws.lb), len(fin), opt.sum(), ver.sum(), ver[order].sum(), ok, cold['iters'].mean()), flush=True)
    print('sd %.3f TOTAL %s' % (sd, tot), flush=True)
