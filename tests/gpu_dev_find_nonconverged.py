"""Diagnostic (GPU box): replays the first steps of the published sd = .003 study with cold searches on the HIP backend and
dumps every node the kernel does not converge on (status MAXITER / NUMERICAL) with its state: gpurun_out/nonconverged.npz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
from helpers import make_controller, load_fixture

ref = load_fixture('reference_closed_loop')
errors = ref['errors_0003']
hip = make_controller('cart_pole_with_walls', backend='hip')
bad_x, bad_f, bad_s = [], [], []
inner = hip.qp.solve_batch


def watching(x0, fix, warm=None):
    r = inner(x0, fix, warm=warm)
    for b in np.flatnonzero(r['status'] > 1):
        bad_x.append(np.array(x0 if np.ndim(x0) == 1 else x0[b])); bad_f.append(np.array(fix[b])); bad_s.append(int(r['status'][b]))
        r['status'][b] = 1            # (let the search go on: treat as pruned)
        r['obj'][b] = np.inf
    return r


hip.qp.solve_batch = watching
X0 = np.array([0., 0., 1., 0.])
for sim in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    x = X0.copy()
    for t in range(5):
        try:
            sol, leaves, solves, _ = hip.feedforward(x, printing_period=None)
        except Exception as e:
            print('sim', sim, 'step', t, 'raised', repr(e)[:200])
            break
        if sol is None:
            break
        x = sol.variables['x'][1] + errors[sim, t]
print('non-converged nodes:', len(bad_x), bad_s)
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/nonconverged.npz', x0=np.array(bad_x), fix=np.array(bad_f), status=np.array(bad_s))
