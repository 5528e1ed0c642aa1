"""Large randomized GPU <-> oracle sweep (diagnostic): many frontiers, random per-node initial states."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier
# DBG_SYSTEM / DBG_T select the system and horizon of the random-depth family (default: the headline system, N=20);
# DBG_REPS the number of frontiers; DBG_SKIP_WIDE=1 leaves the second family out
NAME = os.environ.get('DBG_SYSTEM', 'cart_pole_with_walls')
T = int(os.environ.get('DBG_T', 20))
hip = make_controller(NAME, T=T, backend='hip')
orc = make_controller(NAME, T=T, backend='oracle', threads=16)
# the oracle run tighter than the product (tol 1e-10, polish from a converged iterate): kernel and default oracle are the same
# algorithm, a defect they share shows only against this one
tight = make_controller(NAME, T=T, backend='oracle', threads=16, tol=1e-10, polish_tol=1e-8)
NUB, NU = hip.mld.nub, hip.mld.nu
rng = np.random.default_rng(123)
tot = bad_status = unpolished = 0
worst = worst_fc = worst_tight = 0.
nbig = nbig_tight = tight_unpolished = 0
for rep in range(int(os.environ.get('DBG_REPS', 24))):
    B = int(rng.choice([64, 300, 700, 2048, 4096]))
    p_one = float(rng.choice([0.02, 0.1, 0.3, 0.5]))
    fix = random_prefix_frontier(T, NUB, B, p_one=p_one, seed0=100000 + 5000 * rep)
    x0 = rng.uniform(-1, 1, (B, 4)) * np.array([.3, .2, 1.0, .8])
    a, b = hip.qp.solve_batch(x0, fix), orc.qp.solve_batch(x0, fix)
    tot += B
    ns = int((a['status'] != b['status']).sum())
    bad_status += ns
    fin = (a['status'] == 0) & (b['status'] == 0)
    if fin.any():
        xa, xb = a['primal'][fin][:, :(T + 1) * 4], b['primal'][fin][:, :(T + 1) * 4]
        dev = np.max(np.abs(xa - xb), axis=1) / np.maximum(1e-2, np.max(np.abs(xb), axis=1))
        worst = max(worst, float(dev.max()))
        nbig += int((dev > 1e-5).sum())
        fa, fb = a['primal'][fin][:, (T + 1) * 4:].reshape(-1, T, NU)[:, :, 0], b['primal'][fin][:, (T + 1) * 4:].reshape(-1, T, NU)[:, :, 0]
        worst_fc = max(worst_fc, float((np.max(np.abs(fa - fb), axis=1) / np.maximum(1e-2, np.max(np.abs(fb), axis=1))).max()))
        unpolished += int((a['polished'][fin] == 0).sum())
        c = tight.qp.solve_batch(x0, fix)
        xc = c['primal'][fin][:, :(T + 1) * 4]
        devt = np.max(np.abs(xa - xc), axis=1) / np.maximum(1e-2, np.max(np.abs(xc), axis=1))
        # (only where the tight run ends on a polished vertex: an interior-point iterate, even of gap 1e-10, is ~sqrt(gap) off in the
        # trajectory -- rep 71 of the 96-frontier sweep: the tight run leaves one node unpolished, 5.1e-5 from the vertex that
        # kernel and oracle both return, with an objective 4e-9 ABOVE theirs)
        tp = c['polished'][fin] > 0
        tight_unpolished += int((~tp).sum())
        if tp.any():
            worst_tight = max(worst_tight, float(devt[tp].max()))
            nbig_tight += int((devt[tp] > 1e-5).sum())
    print('rep %2d B %4d p %.2f: status mismatches %d, not converged hip %d oracle %d, feasible %d, worst dev so far %.1e, > 1e-5: %d'
          % (rep, B, p_one, ns, int((a['status'] > 1).sum()), int((b['status'] > 1).sum()), int(fin.sum()), worst, nbig), flush=True)
print('%s N=%d: ' % (NAME, T), end='')
print('TOTAL nodes %d, status mismatches %d, worst state-trajectory deviation %.2e (penalised input %.2e), nodes above 1e-5: %d, optimal nodes left unpolished by the kernel: %d'
      % (tot, bad_status, worst, worst_fc, nbig, unpolished))
print('against the tight oracle (where it ends polished; it leaves %d optimal nodes unpolished): worst deviation %.2e, nodes above 1e-5: %d' % (tight_unpolished, worst_tight, nbig_tight))

# ---- second family: shallow prefixes, wide initial states (feasible-heavy; the nodes whose active sets need the second
# penalty level of the polish live here), every system; the kernel against the oracle and -- independent of both --
# against the dense active-set solve of tests/dense_qp.py on the first 300 optimal nodes of each batch
import dense_qp
for name, T2, B, width, depth, p1 in () if os.environ.get('DBG_SKIP_WIDE') else (('cart_pole_one_wall', 40, 12000, .6, 30, .15), ('cart_pole_with_walls', 20, 6000, .5, 40, .3),
                                     ('cart_pole_with_walls', 40, 3000, .5, 40, .3)):
    h2 = make_controller(name, T=T2, backend='hip')
    o2 = make_controller(name, T=T2, backend='oracle', threads=16)
    nub = h2.mld.nub
    r2 = np.random.RandomState(3)
    fix = np.full((B, T2 * nub), -1, np.int8)
    for k in range(B):
        dep = r2.randint(0, depth)
        fix[k, :dep] = r2.rand(dep) < p1
    from helpers import load_fixture
    x0 = (r2.rand(B, 4) - .5) * 2 * load_fixture(name)['x_max'] * width
    a, b = h2.qp.solve_batch(x0, fix), o2.qp.solve_batch(x0, fix)
    fin = (a['status'] == 0) & (b['status'] == 0)
    xa, xb = a['primal'][fin][:, :(T2 + 1) * 4], b['primal'][fin][:, :(T2 + 1) * 4]
    dev = np.max(np.abs(xa - xb), axis=1) / np.maximum(1e-2, np.max(np.abs(xb), axis=1))
    dq = dense_qp.dense_qp(h2)
    wd = 0.
    for i in np.flatnonzero(fin & (a['polished'] > 0))[:300]:
        w, _ = dense_qp.active_set_primal(h2, dq, x0[i], fix[i], a['dual'][i])
        X = a['primal'][i][:(T2 + 1) * 4]
        wd = max(wd, float(np.abs(X - w[:(T2 + 1) * 4]).max() / (1 + np.abs(X).max())))
    print('%s N=%d shallow/wide: nodes %d, status mismatches %d, optimal %d, unpolished hip %d oracle %d, worst dev hip-oracle %.2e, hip-dense %.2e, iterations per optimal node %.2f'
          % (name, T2, B, int((a['status'] != b['status']).sum()), int(fin.sum()), int((a['polished'][fin] == 0).sum()),
             int((b['polished'][fin] == 0).sum()), float(dev.max()), wd, float((a['iters'][fin] & 0xFFFF).mean())), flush=True)
