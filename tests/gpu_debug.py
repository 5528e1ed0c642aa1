"""Ad-hoc GPU-vs-oracle comparison used while bringing the kernel up (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller, random_prefix_frontier

T = int(os.environ.get('DBG_T', 20))
nb = int(os.environ.get('DBG_B', 64))
co = make_controller(T=T, backend='oracle')
ch = make_controller(T=T, backend='hip')
x0 = np.array([0., 0., 1., 0.])
fix = random_prefix_frontier(T, 4, nb)
fix[0, :] = -1
ro = co.qp.solve_batch(x0, fix)
t = time.time(); rh = ch.qp.solve_batch(x0, fix); t = time.time() - t
print('hip time', t, 'grid/lds', ch.qp.launch_info())
print('status oracle', np.bincount(ro['status'], minlength=4), 'hip', np.bincount(rh['status'], minlength=4))
print('iters oracle', ro['iters'][:16], 'hip', rh['iters'][:16])
mism = np.flatnonzero(ro['status'] != rh['status'])
print('status mismatches', mism[:20])
fin = np.isfinite(ro['obj']) & np.isfinite(rh['obj'])
if fin.any():
    print('max rel obj diff', np.max(np.abs(ro['obj'][fin] - rh['obj'][fin]) / (1e-12 + np.abs(ro['obj'][fin]))))
    print('max primal diff', np.max(np.abs(ro['primal'][fin] - rh['primal'][fin])))
    print('max dual diff', np.max(np.abs(ro['dual'][fin] - rh['dual'][fin])), 'scale', np.max(np.abs(ro['dual'][fin])))
inf = (ro['status'] == 1) & (rh['status'] == 1)
if inf.any():
    print('farkas obj oracle/hip', ro['dual_obj'][inf][:4], rh['dual_obj'][inf][:4])
    print('max farkas dual diff', np.max(np.abs(ro['dual'][inf] - rh['dual'][inf])))
print('obj', ro['obj'][:6], rh['obj'][:6])
