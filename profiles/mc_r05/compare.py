"""Puts the arrays of this directory (written by `python -m warm_start_hmpc_amd.monte_carlo ... --width 1` on the GPU,
fleet driver, final kernels of round 5, searches WITHOUT the parent -> child hand-down) beside the reference's published ones (tests/golden/reference_closed_loop.npz).

    python profiles/mc_r04/compare.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ref = np.load(os.path.join(HERE, '..', '..', 'tests', 'golden', 'reference_closed_loop.npz'))
print('| sd | steps compared | covers equal to published | cold solves: here / published (equal, within 3) | warm solves (steps >= 1): here / published | proofs lost per shift |')
print('|---|---|---|---|---|---|')
for sd, tag in (('0.001', '0001'), ('0.003', '0003'), ('0.010', '0010')):
    lw, cs, ws, ro = (np.load(os.path.join(HERE, '%s_sd_%s.npy' % (k, sd))) for k in ('len_ws', 'nodes_cs', 'nodes_ws', 'reopened'))
    steps = ref['steps_' + tag] if 'steps_' + tag in ref.files else np.full(100, 50)
    full = np.flatnonzero(steps == 50)                       # the arrays hold the completed simulations, in order
    plw, pcs, pws = (ref['nodes_%s_%s' % (k, tag)][full] for k in ('len_ws', 'cs', 'ws'))
    assert lw.shape == plw.shape
    diff = np.argwhere(lw != plw)
    print('| %s | %d | %d of %d%s | %.2f / %.2f (%.0f %%, %.2f %%) | %.2f / %.2f | %.2f |'
          % (sd, lw.size, (lw == plw).sum(), lw.size, '' if not len(diff) else ' (simulation %d, steps %s: %s here, %s published)'
             % (full[diff[0, 0]], diff[:, 1].tolist(), lw[tuple(diff.T)].tolist(), plw[tuple(diff.T)].tolist()),
             cs.mean(), pcs.mean(), 100 * (cs == pcs).mean(), 100 * (np.abs(cs - pcs) <= 3).mean(),
             ws[:, 1:].mean(), pws[:, 1:].mean(), ro.mean()))
