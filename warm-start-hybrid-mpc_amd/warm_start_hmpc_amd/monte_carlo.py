"""Closed-loop Monte-Carlo study on the GPU path -- counterpart of the reference's
``notebooks/cart_pole_with_walls/statistical_analysis.py`` (same initial state, horizon, error model
``e_t = sd * randn(nx) * x_max`` drawn from ``np.random.seed(simulation index)``, one cold-started and
one warm-started branch and bound per step, equal-cost assertion), with all simulations advanced in
lockstep so that their branch-and-bound rounds share kernel launches.

    python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz \
        --sims 100 --steps 50 --sd 0.003 --out gpurun_out/mc

writes ``solve_log_sd_<sd>.log`` (one line per simulation and step, fields named as in the reference's
log: ``cs`` cold solves, ``ws`` warm solves, ``ws info`` cover size, ``e`` error norm) and
``nodes_{cs,ws,len_ws}_sd_<sd>.npy`` so that the numbers can be put next to
``notebooks/cart_pole_with_walls/data/`` of the reference.  Gurobi's own MIQP columns (``grb``,
``grb_fair``) have no counterpart here.
"""
import argparse
import os
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--fixture', required=True)
    ap.add_argument('--sims', type=int, default=100)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--sd', type=float, default=0.0)
    ap.add_argument('--width', type=int, default=8)
    ap.add_argument('--no-cold', action='store_true')
    ap.add_argument('--out', default='.')
    args = ap.parse_args(argv)

    from .mld_system import MLDSystem
    from .controller import HybridModelPredictiveController
    from .batched import BatchedMPC
    d = np.load(args.fixture)
    mld = MLDSystem([d['A'], d['B']], [d['F'], d['G'], d['h']], int(d['nub']))
    ctrl = HybridModelPredictiveController(mld, int(d['T']), [d['Q'], d['R'], d['Q_T']], [d['F_T'], d['h_T']])
    bm = BatchedMPC(ctrl)
    os.makedirs(args.out, exist_ok=True)
    tag = 'sd_{:.3f}'.format(args.sd)
    with open(os.path.join(args.out, 'solve_log_%s.log' % tag), 'w') as log:
        log.write('Error standard deviation {:.3f}\n\n'.format(args.sd))
        st = bm.closed_loop(np.array([0., 0., 1., 0.]), args.steps, e_sd=args.sd, seeds=tuple(range(args.sims)),
                            x_max=d['x_max'], frontier_width=args.width, cold_too=not args.no_cold, log=log)
    full = [k for k in range(args.sims) if len(st['nodes_ws'][k]) == args.steps and len(st['costs'][k]) == args.steps]
    for key in ('nodes_ws', 'nodes_cs', 'len_ws'):
        rows = [st[key][k] for k in full if len(st[key][k]) == args.steps]
        if rows:
            np.save(os.path.join(args.out, '%s_%s.npy' % (key, tag)), np.array(rows))
    ws = np.array([st['nodes_ws'][k] for k in full]) if full else np.zeros((0, args.steps))
    print('simulations %d (completed %d), steps %d, sd %.3f' % (args.sims, len(full), st['steps'], args.sd))
    if full:
        print('warm solves/step (steps >= 1): mean %.2f min %d max %d' % (ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max()))
        if not args.no_cold:
            cs = np.array([st['nodes_cs'][k] for k in full])
            print('cold solves/step: mean %.2f min %d max %d' % (cs.mean(), cs.min(), cs.max()))
        lw = np.array([st['len_ws'][k] for k in full])
        print('cover size: min %d max %d' % (lw.min(), lw.max()))
    print('warm/cold cost disagreements (np.isclose rtol 1e-5): %d of %d steps %s'
          % (len(st['cost_mismatches']), st['steps'], st['cost_mismatches'][:3]))
    print('wall %.2f s, %.1f MPC steps/s (all simulations)' % (st['wall'], st['steps_per_sec']))
    return 0


if __name__ == '__main__':
    sys.exit(main())
