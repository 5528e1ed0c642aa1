"""N > 1 path on CPU: two gloo ranks shard a frontier / a warm-start cover and exchange the
incumbent.  The QP backend here is the CPU oracle (test infrastructure); what is tested is the
sharding, the collectives and that all ranks agree with the single-process answer."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import make_controller, random_prefix_frontier
    from warm_start_hmpc_amd.distributed import solve_frontier_sharded, feedforward_sharded, shard_indices
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle')
    x0 = np.array([0., 0., .5, 0.])
    # 1) synthetic frontier: disjoint shards, global incumbent
    fix = random_prefix_frontier(10, 4, 40, p_one=0.05, seed0=7000)
    fix[3, :] = 0                                  # one fully fixed, feasible node (all binaries 0)
    idx, res, ub = solve_frontier_sharded(ctrl, fix, x0)
    assert np.array_equal(idx, shard_indices(40, rank, world))
    # 2) sharded branch and bound from a cover
    sol, leaves, _, _ = ctrl.feedforward(x0, printing_period=None)
    uc0, ub0 = sol.variables['uc'][0], sol.variables['ub'][0]
    cover = ctrl.construct_warm_start(leaves, x0, uc0, ub0, np.zeros(4))[0]
    x1 = sol.variables['x'][1]
    obj, assign, my_leaves, my_solves = feedforward_sharded(ctrl, x1, cover)
    q.put((rank, idx.tolist(), res['obj'].tolist(), ub, obj, assign.tolist(), my_solves))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_with_single_process():
    from helpers import make_controller, random_prefix_frontier
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle')
    x0 = np.array([0., 0., .5, 0.])
    fix = random_prefix_frontier(10, 4, 40, p_one=0.05, seed0=7000)
    fix[3, :] = 0
    ref = ctrl.qp.solve_batch(x0, fix)
    merged = np.full(40, np.nan)
    for rank, idx, obj, ub, _, _, _ in got:
        merged[idx] = obj
    assert np.array_equal(merged, ref['obj'])                       # shards are disjoint, complete, identical
    full = (fix >= 0).all(axis=1) & (ref['status'] == 0)
    assert got[0][3] == got[1][3] == ref['obj'][full].min()          # same global incumbent on both ranks

    sol, leaves, _, _ = ctrl.feedforward(x0, printing_period=None)
    cover = ctrl.construct_warm_start(leaves, x0, sol.variables['uc'][0], sol.variables['ub'][0], np.zeros(4))[0]
    single = ctrl.feedforward(sol.variables['x'][1], printing_period=None, warm_start=cover)
    for rank, _, _, _, obj, assign, solves in got:
        assert obj == single[0].objective
        assert np.array_equal(np.array(assign), np.array(single[0].variables['ub']))


def _failing_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import make_controller
    from warm_start_hmpc_amd.distributed import feedforward_sharded, PeerFailure
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle')
    x0 = np.array([0., 0., .5, 0.])
    sol, leaves, _, _ = ctrl.feedforward(x0, printing_period=None)
    cover = ctrl.construct_warm_start(leaves, x0, sol.variables['uc'][0], sol.variables['ub'][0], np.zeros(4))[0]
    if rank == 1:                                  # this rank's solver breaks down in its third round
        good, calls = ctrl.qp.solve_batch, [0]

        def solve_batch(x, fix, **kw):
            calls[0] += 1
            res = good(x, fix, **kw)
            if calls[0] == 3:
                res['status'][:] = 3               # NUMERICAL: the controller raises
            return res
        ctrl.qp.solve_batch = solve_batch
    try:
        feedforward_sharded(ctrl, sol.variables['x'][1], cover, frontier_width=2)
        q.put((rank, 'no error'))
    except PeerFailure:
        q.put((rank, 'peer'))
    except RuntimeError:
        q.put((rank, 'own'))
    dist.barrier()                                 # both ranks are out of the search: nobody hangs
    dist.destroy_process_group()


def test_a_failing_rank_does_not_leave_the_other_in_a_collective():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == {0: 'peer', 1: 'own'}


def _mc_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from warm_start_hmpc_amd import monte_carlo
    from oracle.oracle_qp import OracleBatchedQP
    rc = monte_carlo.main(['--fixture', os.path.join(ROOT, 'tests', 'golden', 'cart_pole_with_walls.npz'), '--sims', '3',
                           '--steps', '3', '--sd', '0.003', '--no-cold', '--out', out],
                          backend_factory=lambda data: OracleBatchedQP(data, threads=2))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(rc)


def test_monte_carlo_shards_simulations_over_ranks(tmp_path):
    # simulations are the communication-free axis; two ranks must reproduce the single-process study
    from warm_start_hmpc_amd import monte_carlo
    from oracle.oracle_qp import OracleBatchedQP
    one, two = str(tmp_path / 'one'), str(tmp_path / 'two')
    fixture = os.path.join(ROOT, 'tests', 'golden', 'cart_pole_with_walls.npz')
    assert monte_carlo.main(['--fixture', fixture, '--sims', '3', '--steps', '3', '--sd', '0.003', '--no-cold', '--out', one],
                            backend_factory=lambda data: OracleBatchedQP(data, threads=4)) == 0
    ctx = mp.get_context('spawn')
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_mc_worker, args=(r, 2, port, two)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for key in ('nodes_ws', 'len_ws'):
        a = np.load(os.path.join(one, '%s_sd_0.003.npy' % key))
        b = np.load(os.path.join(two, '%s_sd_0.003.npy' % key))
        assert a.shape == (3, 3) and np.array_equal(a, b), key
    assert os.path.exists(os.path.join(two, 'solve_log_sd_0.003.rank1.log'))


def _strong_scaling_worker(rank, world, port, q):
    # bench.py --gpus N --frontier-total 256 (BASELINE configs[2] as specified: ONE frontier split over the ranks, node k
    # to rank k mod N, one MIN all-reduce of the incumbent per step), rehearsed with the oracle backend under gloo: the
    # frontier construction and the sharding are bench.py's own functions
    for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from helpers import make_controller, load_fixture
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=2)
    x0_all, fix_all, _ = bench.real_tree_frontier(ctrl, 256, 0, load_fixture('cart_pole_with_walls')['x_max'], x_center=np.array([0., 0., .5, 0.]))
    mine = bench.shard(256, world, rank)
    res = ctrl.qp.solve_batch(np.ascontiguousarray(x0_all[mine]), np.ascontiguousarray(fix_all[mine]))
    full = (fix_all[mine] >= 0).all(axis=1) & (res['status'] == 0)
    ub = torch.tensor([res['obj'][full].min() if full.any() else np.inf], dtype=torch.float64)
    dist.all_reduce(ub, op=dist.ReduceOp.MIN)
    q.put((rank, mine.tolist(), res['obj'].tolist(), float(ub.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_frontier_of_the_bench_under_gloo():
    import bench
    from helpers import make_controller, load_fixture
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_strong_scaling_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=4)
    x0_all, fix_all, _ = bench.real_tree_frontier(ctrl, 256, 0, load_fixture('cart_pole_with_walls')['x_max'], x_center=np.array([0., 0., .5, 0.]))
    ref = ctrl.qp.solve_batch(x0_all, fix_all)
    merged = np.full(256, np.nan)
    for _, idx, obj, _ in got:
        assert np.all(np.isnan(merged[idx]))                            # disjoint shards ...
        merged[idx] = obj
    assert np.array_equal(merged, ref['obj'])                           # ... that cover the frontier; same records
    full = (fix_all >= 0).all(axis=1) & (ref['status'] == 0)
    assert got[0][3] == got[1][3] == ref['obj'][full].min()             # the same global incumbent on both ranks


class _LockstepOnTheOracle(object):
    """What bench.sharded_fleet_rate needs of a fleet (closed_loop -> nodes_ws, len_ws as arrays), on the numpy lockstep
    driver over the CPU oracle: the rehearsal of the sharded closed-loop leg of the bench line."""

    def __init__(self, ctrl, K):
        from warm_start_hmpc_amd.batched import BatchedMPC
        self.driver, self.K = BatchedMPC(ctrl), K

    def closed_loop(self, x0, n_steps, errors, frontier_width=8, **_):
        st = self.driver.closed_loop(x0, n_steps, seeds=tuple(range(self.K)), frontier_width=frontier_width, errors=errors)
        return dict(nodes_ws=np.array(st['nodes_ws']), len_ws=np.array(st['len_ws']), wall=st['wall'])


def _sharded_fleet_worker(rank, world, port, q):
    # the `mpc_steps_per_sec.fleet_sharded` key of a multi-GPU bench line (bench.sharded_fleet_rate: simulations sharded over the
    # ranks, barriers around the timed regions, MAX of the wall times, SUM of the steps), rehearsed under gloo on the oracle
    for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from helpers import make_controller
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='oracle', threads=2)
    r = bench.sharded_fleet_rate(ctrl, world, rank, dist.barrier, dist, 'cpu', total_loops=5, steps=2, fleet_factory=_LockstepOnTheOracle,
                                 T_state=[0., 0., .5, 0.])
    q.put((rank, r))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_closed_loops_of_the_bench_under_gloo():
    import bench
    assert bench.fleet_shard(5, 2, 0) == [0, 2, 4] and bench.fleet_shard(5, 2, 1) == [1, 3]
    assert sorted(s for r in range(8) for s in bench.fleet_shard(1024, 8, r)) == list(range(1024))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_fleet_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = got[0][1], got[1][1]
    # every rank reports the whole job: 5 loops, the same rate, the published cover of the warm-started steps
    assert a['loops_total'] == b['loops_total'] == 5 and a['loops_per_rank'] == 3 and b['loops_per_rank'] == 2
    assert a['value'] == b['value'] > 0 and a['steps_timed'] == 2
    assert a['warm_solves_per_step_mean'] == b['warm_solves_per_step_mean'] and 5 <= a['warm_solves_per_step_mean'] <= 30
    assert a['cover_min_max'] == b['cover_min_max']
