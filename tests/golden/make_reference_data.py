"""Regenerates tests/golden/reference_closed_loop.npz from the reference's PUBLISHED experiment output.

Data only (no reference code): the model errors the reference applied in its closed-loop Monte-Carlo study and the
node counts it recorded per MPC step,
    /root/reference/notebooks/cart_pole_with_walls/data/errors_sd_{0.001,0.003}.npy        (100 x 50 x 4 float64)
    /root/reference/notebooks/cart_pole_with_walls/data/nodes_{cs,ws,len_ws}_sd_{...}.npy  (100 x 50 int64)
written by notebooks/cart_pole_with_walls/statistical_analysis.py:199-207 (cs = QP solves of the cold-started branch
and bound at that step, ws = of the warm-started one, len_ws = size of the warm start built after the step).
The first N_SIMS simulations of each noise level are kept.  Runs in the build container only (the reference does
not travel); tests replay these disturbances through this repository's controller
(tests/test_reference_replay.py).

    python tests/golden/make_reference_data.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = '/root/reference/notebooks/cart_pole_with_walls/data'
N_SIMS = 12

if __name__ == '__main__':
    out = {}
    for sd in ('0.001', '0.003'):
        tag = sd.replace('.', '')
        for key in ('errors', 'nodes_cs', 'nodes_ws', 'nodes_len_ws'):
            a = np.load(os.path.join(DATA, '%s_sd_%s.npy' % (key, sd)), allow_pickle=False)
            assert a.shape[:2] == (100, 50)
            out['%s_%s' % (key, tag)] = a[:N_SIMS]
        cs, ws, lw = (np.load(os.path.join(DATA, 'nodes_%s_sd_%s.npy' % (k, sd))) for k in ('cs', 'ws', 'len_ws'))
        print('sd', sd, 'published over 100 simulations: cold %.1f (%d..%d), warm (steps >= 1) %.2f (%d..%d), cover %d..%d'
              % (cs.mean(), cs.min(), cs.max(), ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max(), lw.min(), lw.max()))
        # summary of ALL 100 published simulations, for the distribution checks
        out['summary_%s' % tag] = np.array([cs.mean(), cs.min(), cs.max(), ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max(), lw.min(), lw.max()])
    np.savez_compressed(os.path.join(HERE, 'reference_closed_loop.npz'), **out)
