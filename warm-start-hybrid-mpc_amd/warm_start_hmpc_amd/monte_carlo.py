"""Closed-loop Monte-Carlo study on the GPU path -- counterpart of the reference's
``notebooks/cart_pole_with_walls/statistical_analysis.py`` (same initial state, horizon, error model
``e_t = sd * randn(nx) * x_max`` drawn from ``np.random.seed(simulation index)``, one cold-started and
one warm-started branch and bound per step, equal-cost assertion), with all simulations advanced in
lockstep so that their branch-and-bound rounds share kernel launches.  On the GPU the study runs on the C++ fleet
driver behind the C ABI (``fleet.closed_loop_study`` over ``hmpc_fleet_*``: trees behind the handle, multiplier rows
resident in HBM); ``--errors FILE.npz --errors-key KEY`` replays prescribed disturbances instead of drawing them (the
reference's published ones: tests/golden/reference_closed_loop.npz, keys errors_0001 / errors_0003 / errors_0010).

    python -m warm_start_hmpc_amd.monte_carlo --fixture tests/golden/cart_pole_with_walls.npz \
        --sims 100 --steps 50 --sd 0.003 --out gpurun_out/mc

writes ``solve_log_sd_<sd>.log`` (one line per simulation and step, fields named as in the reference's
log: ``cs`` cold solves, ``ws`` warm solves, ``ws info`` cover size, ``e`` error norm) and
``nodes_{cs,ws,len_ws}_sd_<sd>.npy`` so that the numbers can be put next to
``notebooks/cart_pole_with_walls/data/`` of the reference.  Gurobi's own MIQP columns (``grb``,
``grb_fair``) have no counterpart here.

The simulations are independent: launched under ``torch.distributed.run`` with N ranks (one per GPU),
rank r advances the simulations r, r + N, r + 2N, ... on its own GPU with no communication during the
study; at the end the per-simulation statistics are gathered on rank 0, which writes the arrays and the
summary (the per-step log lines of every rank go to ``solve_log_<sd>.rank<r>.log``).
"""
import argparse
import os
import sys

import numpy as np


def main(argv=None, backend_factory=None):
    """``backend_factory`` (problem data -> batched QP backend) is a hook for the CPU tests; the default is the HIP
    library on this rank's GPU."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--fixture', required=True)
    ap.add_argument('--sims', type=int, default=100)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--sd', type=float, default=0.0)
    ap.add_argument('--width', type=int, default=8)
    ap.add_argument('--no-cold', action='store_true')
    ap.add_argument('--out', default='.')
    ap.add_argument('--errors', default=None, help='.npz with prescribed model errors (sims, steps, nx)')
    ap.add_argument('--errors-key', default='errors')
    ap.add_argument('--no-handdown', action='store_true',
                    help='searches without the parent -> child hand-down of active sets, as in the reference\'s published runs '
                         '(with it, solve counts move by a few where multipliers are not unique)')
    args = ap.parse_args(argv)

    from .mld_system import MLDSystem
    from .controller import HybridModelPredictiveController
    from .batched import BatchedMPC
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group('nccl' if (backend_factory is None and torch.cuda.is_available()) else 'gloo')
    d = np.load(args.fixture)
    mld = MLDSystem([d['A'], d['B']], [d['F'], d['G'], d['h']], int(d['nub']))
    data = dict(T=int(d['T']), objective=[d['Q'], d['R'], d['Q_T']], terminal_set=[d['F_T'], d['h_T']])
    if backend_factory is None:
        ctrl = HybridModelPredictiveController(mld, data['T'], data['objective'], data['terminal_set'],
                                               solver_params={'device': local if world > 1 else -1})
    else:
        class _Late(object):          # the controller wants a backend at construction; bind it right after
            pass
        ctrl = HybridModelPredictiveController(mld, data['T'], data['objective'], data['terminal_set'], backend=_Late())
        ctrl.qp = backend_factory(ctrl.problem_data())
    os.makedirs(args.out, exist_ok=True)
    tag = 'sd_{:.3f}'.format(args.sd)
    seeds = tuple(range(rank, args.sims, world))           # this rank's simulations
    # the disturbance of simulation i at step t: e_sd * randn(nx) * x_max from the stream of np.random.seed(i)
    # (statistical_analysis.py:73,176), or the prescribed one
    if args.errors is not None:
        errors = np.load(args.errors)[args.errors_key][list(seeds), :args.steps]
    else:
        errors = np.array([[args.sd * np.multiply(rng.randn(mld.nx), d['x_max']) for _ in range(args.steps)]
                           for rng in (np.random.RandomState(s) for s in seeds)]).reshape(len(seeds), args.steps, mld.nx)
    x0 = np.array([0., 0., 1., 0.])
    log_name = 'solve_log_%s.log' % tag if world == 1 else 'solve_log_%s.rank%d.log' % (tag, rank)
    with open(os.path.join(args.out, log_name), 'w') as log:
        log.write('Error standard deviation {:.3f}\n\n'.format(args.sd))
        if backend_factory is None:                        # the product path: the fleet driver behind the C ABI
            from .fleet import closed_loop_study
            st = closed_loop_study(ctrl, x0, errors, frontier_width=args.width, cold_too=not args.no_cold, log=log, sim_ids=seeds,
                                   handdown=not args.no_handdown)
        else:
            st = BatchedMPC(ctrl).closed_loop(x0, args.steps, seeds=seeds, frontier_width=args.width, cold_too=not args.no_cold,
                                              log=log, errors=errors)
    if world > 1:
        # per-simulation lists back into global simulation order; counters summed; wall = slowest rank
        parts = [None] * world
        dist.all_gather_object(parts, {k: st[k] for k in ('nodes_ws', 'nodes_cs', 'len_ws', 'reopened', 'costs', 'cost_mismatches', 'steps', 'wall')})
        merged = {k: [None] * args.sims for k in ('nodes_ws', 'nodes_cs', 'len_ws', 'reopened', 'costs')}
        for r, part in enumerate(parts):
            for j, sim in enumerate(range(r, args.sims, world)):
                for k in merged:
                    merged[k][sim] = part[k][j]
        st = dict(merged, cost_mismatches=sum((p['cost_mismatches'] for p in parts), []), steps=sum(p['steps'] for p in parts),
                  wall=max(p['wall'] for p in parts))
        st['steps_per_sec'] = st['steps'] / st['wall'] if st['wall'] > 0 else 0.
        if rank != 0:
            return 0
    full = [k for k in range(args.sims) if len(st['nodes_ws'][k]) == args.steps and len(st['costs'][k]) == args.steps]
    for key in ('nodes_ws', 'nodes_cs', 'len_ws', 'reopened'):
        rows = [st[key][k] for k in full if len(st[key][k]) == args.steps]
        if rows:
            np.save(os.path.join(args.out, '%s_%s.npy' % (key, tag)), np.array(rows))
    ws = np.array([st['nodes_ws'][k] for k in full]) if full else np.zeros((0, args.steps))
    print('simulations %d (completed %d), steps %d, sd %.3f, hand-down %s' % (args.sims, len(full), st['steps'], args.sd, 'off' if args.no_handdown else 'on'))
    if full:
        print('warm solves/step (steps >= 1): mean %.2f min %d max %d' % (ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max()))
        if not args.no_cold:
            cs = np.array([st['nodes_cs'][k] for k in full])
            print('cold solves/step: mean %.2f min %d max %d' % (cs.mean(), cs.min(), cs.max()))
        lw = np.array([st['len_ws'][k] for k in full])
        print('cover size: min %d max %d' % (lw.min(), lw.max()))
        print('infeasibility proofs lost per shift: mean %.2f' % np.mean([st['reopened'][k] for k in full]))
    ended = [(k, len(st['costs'][k])) for k in range(args.sims) if k not in full]
    if ended:
        print('simulations that left the feasible set (simulation, step): %s' % ended)
    print('warm/cold cost disagreements (np.isclose rtol 1e-5): %d of %d steps %s'
          % (len(st['cost_mismatches']), st['steps'], st['cost_mismatches'][:3]))
    print('wall %.2f s, %.1f MPC steps/s (all simulations)' % (st['wall'], st['steps_per_sec']))
    return 0


if __name__ == '__main__':
    sys.exit(main())
