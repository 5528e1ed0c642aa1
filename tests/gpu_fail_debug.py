import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import make_controller
d = np.load(sys.argv[1])
print('status', d['status'], 'iters', d['iters'], 'x0', d['x0'], 'fixed', (d['fix'] >= 0).sum(axis=1))
co = make_controller(backend='oracle')
ro = co.qp.solve_batch(d['x0'], d['fix'])
print('oracle status', ro['status'], 'iters', ro['iters'], 'obj', ro['obj'])
ch = make_controller(backend='hip')
rh = ch.qp.solve_batch(d['x0'], d['fix'])
print('hip status', rh['status'], 'iters', rh['iters'], 'obj', rh['obj'])
