#!/usr/bin/env python
"""bench.py -- QP subproblems/s of the batched B&B-node solver on MI355X.

A "step" is one pass of the hot path over one synthetic frontier: every rank
solves ``--frontier`` nodes of the cart-pole-with-walls MIQP (N=20, 4
binaries/step; BASELINE.json configs[1]) with inputs already resident in HBM,
then -- when more than one rank runs -- all ranks exchange the incumbent upper
bound with one RCCL all-reduce(min) of 8 bytes.  Frontier nodes are independent,
so ranks hold disjoint shards and the scaling is weak (fixed work per GPU).

The default frontier (``--frontier-kind real_tree``) is SURVEY.md 8(d) C2's
"replay frontier" -- what a branch and bound actually solves: every node a
cold-started search from x0 = [0, 0, 1, 0] solves plus its leaves, tiled to the
frontier size; a third of the nodes optimal, the rest infeasible; each node
solved cold (no hand-down from its parent: every record is what a stand-alone
solve returns).  ``--states distinct`` takes the trees of distinct closed-loop
states instead (x0 + 0.05 N(0,1) x_max per tree: 6 % of those nodes need the
terminal set and are solved twice); ``--frontier-kind random_prefix`` is the
random-binary generator of the same section (p = 0.5: 98 % infeasible, the easy
mix; p = 0.1).  All of these, the hand-down variants and the other configs are
secondary keys of the default run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frontier B]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def progress(msg):
    """One line on stderr per stage: a run that is silent for minutes is taken to be hung by the GPU pool."""
    sys.stderr.write('[bench %7.1f s] %s\n' % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()
if os.environ.get('BENCH_WATCHDOG'):     # diagnostic: where is the interpreter every N seconds
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ['BENCH_WATCHDOG']), repeat=True, file=sys.stderr)


def cpu_baseline(ctrl, x0, fix):
    """The CPU oracle (a float64 port of the same QP statement, oracle/hsde_qp.c) timed on the
    host cores of this box, on the same frontier.  A reported baseline, not the thing shipped."""
    from oracle.oracle_qp import OracleBatchedQP
    # the box gives each GPU a share of the host cores (16 per GPU); never oversubscribe beyond it
    cores = min(len(os.sched_getaffinity(0)), 16)
    orc = OracleBatchedQP(ctrl.problem_data(), threads=cores)

    def timed(solver, x, f, warm=3, reps=10):       # 3 warm-up batches, median of >= 10
        for _ in range(warm):
            solver.solve_batch(x, f)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            solver.solve_batch(x, f)
            ts.append(time.perf_counter() - t)
        return float(np.median(ts))
    best = timed(orc, x0, fix)
    one = OracleBatchedQP(ctrl.problem_data(), threads=1)
    sub = fix[:min(256, len(fix))]
    t1 = timed(one, x0, sub)
    out = {'value': len(fix) / best, 'unit': 'QP subproblems/s', 'cores': cores, 'kind': 'port',
           'sample': '%d-node frontier of this run, OpenMP over nodes, 3 warm-ups, median of 10 batches (single thread: first %d nodes)' % (len(fix), len(sub)),
           'single_thread_value': len(sub) / t1}
    # BASELINE.md section 3 / SURVEY 8d(ii): the reference's own engine beside it, if it exists on this host.  The QP path
    # over Gurobi is tests/gurobi_reference.py (the model of controller.py:119-184, the optimize / Farkas sequence of
    # bounded_qp.py:200-228, one node at a time on one thread as the reference); where it runs, IT is the CPU baseline
    # (kind "gurobi") and the oracle port moves to the key `port`.
    try:
        import gurobi_reference
        if not gurobi_reference.available():
            raise ImportError('gurobipy')
        grb = gurobi_reference.GurobiBatchedQP(ctrl.problem_data(), gurobi_params={'Threads': 1})
        gsub = fix[:min(512, len(fix))]                      # (~0.3 k QP/s published: a bounded sample, 2 - 10 s)
        grb.solve_batch(x0 if np.ndim(x0) == 1 else x0[:len(gsub)], gsub[:32])
        r = grb.solve_batch(x0 if np.ndim(x0) == 1 else x0[:len(gsub)], gsub)
        out = {'value': len(gsub) / r['time'], 'unit': 'QP subproblems/s', 'cores': 1, 'kind': 'gurobi',
               'sample': 'first %d nodes of the frontier of this run, one Gurobi solve per node on one thread (as the reference), wall time; '
                         'Gurobi\'s own Runtime: %.0f QP/s' % (len(gsub), len(gsub) / r['solver_time']),
               'status_equal_to_port': bool(np.array_equal(r['status'], orc.solve_batch(x0 if np.ndim(x0) == 1 else x0[:len(gsub)], gsub)['status'])),
               'port': out}
    except Exception as e:
        out['gurobi'] = 'unavailable on this host (%s): the CPU baseline is the oracle port' % (type(e).__name__ + ': ' + str(e))[:160]
    return out


def shift_bandwidth(ctrl, dev, leaves=65536, trees=64, reps=10):
    """The warm-start node shift (hmpc_shift_batch_device, csrc/hmpc_shift.hip): an HBM-bound kernel that reads and
    writes one dual row per leaf.  Device-resident synthetic leaves, HIP events on the launch stream."""
    import torch
    qp = ctrl.qp
    qp.set_shift_maps(ctrl._update['mu'], ctrl._update['rho'], ctrl.mld.V)
    g = torch.Generator(device=dev).manual_seed(0)
    B, K, nd, nf = leaves, trees, qp.n_dual, qp.nfix
    owner = torch.randint(0, K, (B,), device=dev, dtype=torch.int32, generator=g)
    x0 = torch.rand(K, qp.nx, device=dev, dtype=torch.float64, generator=g)
    u0 = torch.zeros(K, qp.nu, device=dev, dtype=torch.float64)
    e0 = 1e-3 * torch.rand(K, qp.nx, device=dev, dtype=torch.float64, generator=g)
    fix = torch.full((B, nf), -1, device=dev, dtype=torch.int8)
    dual = torch.rand(B, nd, device=dev, dtype=torch.float64, generator=g)
    lb = torch.rand(B, device=dev, dtype=torch.float64, generator=g)
    dobj = torch.rand(B, device=dev, dtype=torch.float64, generator=g)
    out = dict(fix=torch.empty_like(fix), lb=torch.empty_like(lb), dual=torch.empty_like(dual), dual_obj=torch.empty_like(dobj),
               flags=torch.empty(B, device=dev, dtype=torch.uint8))
    for _ in range(2):
        qp.shift_batch_device(owner, x0, u0, e0, fix, lb, dual, dobj, out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        qp.shift_batch_device(owner, x0, u0, e0, fix, lb, dual, dobj, out)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    assert int((out['flags'] & 1).sum().item()) == B
    bytes_per_leaf = 2 * (8 * nd + nf + 16) + 4 + 1          # row, identifier, bound and objective in and out, owner, flags
    gbs = bytes_per_leaf * B / (ms * 1e-3) / 1e9
    return {'leaves': B, 'trees': K, 'kernel_ms_avg': ms, 'algorithmic_bytes_per_leaf': bytes_per_leaf,
            'achieved_GBs': gbs, 'peak_GBs': HBM_PEAK_GBS, 'frac': gbs / HBM_PEAK_GBS, 'bound': 'hbm',
            'kernel': 'hmpc_shift_tree_kernel + hmpc_shift_row_kernel (rows staged in LDS by global_load_lds; both launches inside the timed region)'}


def _device_rate(qp, x0_h, fix_h, dev, reps=5, warm=2, parent=None):
    """QP/s of one frontier, inputs resident in HBM, HIP events on the launch stream; statuses and iteration counts.
    parent (int32 [B], -1: none): every node is handed the record of its parent (hmpc_warm), taken from one untimed
    cold pass over the same frontier."""
    import torch
    B = fix_h.shape[0]
    fix = torch.from_numpy(np.ascontiguousarray(fix_h)).to(dev)
    x0 = torch.from_numpy(np.ascontiguousarray(x0_h)).to(dev)
    out = dict(obj=torch.empty(B, dtype=torch.float64, device=dev), dual_obj=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev),
               primal=torch.empty(B, qp.n_primal, dtype=torch.float64, device=dev),
               dual=torch.empty(B, qp.n_dual, dtype=torch.float64, device=dev))
    hand = None
    if parent is not None:
        par = dict(out, primal=torch.empty_like(out['primal']), dual=torch.empty_like(out['dual']))
        qp.solve_batch_device(x0, fix, par)
        torch.cuda.synchronize()
        st, itf = par['status'].cpu().numpy(), par['iters'].cpu().numpy()
        good = (parent >= 0) & (st[np.maximum(parent, 0)] == 0) & (((itf[np.maximum(parent, 0)] >> 16) & 1) > 0)
        hand = (par['primal'], par['dual'], torch.from_numpy(np.where(good, parent, -1).astype(np.int32)).to(dev))
    for _ in range(warm):
        qp.solve_batch_device(x0, fix, out, warm=hand)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        qp.solve_batch_device(x0, fix, out, warm=hand)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    status = out['status'].cpu().numpy()
    raw = out['iters'].cpu().numpy()
    grid, lds = qp.launch_info()
    r = {'nodes': B, 'kernel_ms_avg': ms, 'qp_per_s': B / (ms * 1e-3), 'optimal': int((status == 0).sum()),
         'infeasible': int((status == 1).sum()), 'not_converged': int((status > 1).sum()),
         'polished': int(((raw >> 16) & 1).sum()), 'ipm_iters_mean': float((raw & 0xFFFF).mean()),
         'ipm_iters_mean_optimal': float((raw & 0xFFFF)[status == 0].mean()) if (status == 0).any() else None,
         'grid': grid, 'lds_bytes_per_wg': lds}
    if hand is not None:
        r['handed_down'] = int((hand[2] >= 0).sum().item())
        r['handed_down_verified'] = int(((raw >> 18) & 1).sum())
    return r, status


def secondary_frontiers(ctrl, dev, x_max):
    """The other frontiers of SURVEY 8(d) C2 on the headline system (the headline line is the real-tree frontier solved
    cold): the same frontier with every node handed its parent's record (hmpc_warm), the random-prefix generator at
    p = 0.5 (98 % infeasible: round 2's headline) and p = 0.1, the 1024-node size of BASELINE configs[2], optimal and
    infeasible nodes apart, and the run-time-sized kernel forced onto the same system (what an MLD without a
    compile-time instantiation gets)."""
    from helpers import random_prefix_frontier, make_controller
    T, nub = ctrl.T, ctrl.mld.nub
    x0_1 = np.array([0., 0., 1., 0.])
    out = {}
    x0_n, fix_n, par_n = real_tree_frontier(ctrl, 4096, 0, x_max, spread=0.)
    r, _ = _device_rate(ctrl.qp, x0_n, fix_n, dev, parent=par_n)
    out['replay_frontier_4096_handdown'] = r               # the headline frontier, every node handed its parent's record
    x0_t, fix_t, par_t = real_tree_frontier(ctrl, 4096, 0, x_max)
    r, _ = _device_rate(ctrl.qp, x0_t, fix_t, dev)
    r['note'] = 'trees of distinct closed-loop states: 6 % of the nodes need the terminal set (two solves), the launch is as long as its tail (DESIGN.md 5)'
    out['real_trees_distinct_states_4096'] = r
    r, _ = _device_rate(ctrl.qp, x0_t, fix_t, dev, parent=par_t)
    out['real_trees_distinct_states_4096_handdown'] = r
    r, _ = _device_rate(ctrl.qp, x0_n[:1024], fix_n[:1024], dev)
    out['replay_frontier_1024_configs2_size'] = r
    r, _ = _device_rate(ctrl.qp, x0_n[:1024], fix_n[:1024], dev, parent=np.where(par_n[:1024] < 1024, par_n[:1024], -1))
    out['replay_frontier_1024_handdown'] = r
    r, _ = _device_rate(ctrl.qp, x0_1, random_prefix_frontier(T, nub, 4096, p_one=0.5), dev)
    out['random_prefix_p0.5_4096'] = r
    r, _ = _device_rate(ctrl.qp, x0_1, random_prefix_frontier(T, nub, 4096, p_one=0.1), dev)
    out['random_prefix_p0.1_4096'] = r
    r, _ = _device_rate(ctrl.qp, x0_1, random_prefix_frontier(T, nub, 1024, p_one=0.5), dev)
    out['random_prefix_p0.5_1024_configs2_size'] = r
    # optimal and infeasible nodes apart (the same count of each, drawn from p = 0.1 frontiers)
    pool = random_prefix_frontier(T, nub, 32768, p_one=0.1, seed0=200000)
    _, status = _device_rate(ctrl.qp, x0_1, pool, dev, reps=1, warm=0)
    n = min(int((status == 0).sum()), 2048)
    r, _ = _device_rate(ctrl.qp, x0_1, pool[status == 0][:n], dev)
    out['optimal_nodes_only'] = r
    r, _ = _device_rate(ctrl.qp, x0_1, pool[status == 1][:n], dev)
    out['infeasible_nodes_only'] = r
    # the run-time-sized kernel (sparse lists in LDS, row state in a global slab) on the same system and frontier
    os.environ['HMPC_FORCE_GENERIC'] = '1'
    try:
        gen = make_controller('cart_pole_with_walls', backend='hip')
    finally:
        del os.environ['HMPC_FORCE_GENERIC']
    r, _ = _device_rate(gen.qp, x0_n, fix_n, dev, reps=3, warm=1)
    r['kernel'] = 'hmpc_qp_kernel<0,...> (run-time-sized: any MLD that fits LDS; HMPC_FORCE_GENERIC)'
    out['replay_frontier_4096_generic_kernel'] = r
    return out


def other_configs(dev):
    """BASELINE configs[3] (N = 40) and configs[4] (random MLD nx=20, nu=6+8, N=30) with their own algorithmic bytes."""
    from helpers import make_controller, random_prefix_frontier, random_mld, _NoBackend
    from warm_start_hmpc_amd.controller import HybridModelPredictiveController
    from warm_start_hmpc_amd.qp_backend import HipBatchedQP
    out = {}
    c40 = make_controller('cart_pole_with_walls', T=40, backend='hip')
    # configs[3]: the replay frontier of ITS OWN tree (320 solved nodes + 161 leaves from x0 = [0, 0, 1, 0]),
    # tiled to 2048 -- what a search at N = 40 solves; the random-prefix frontiers (95 - 99 % infeasible) stay as secondary keys
    x40, f40, _ = real_tree_frontier(c40, 2048, 0, None, spread=0.)
    for name, x, f in (('cart_pole_N40_replay_frontier_2048', x40, f40),
                       ('cart_pole_N40_random_prefix_p0.1_2048', np.array([0., 0., 1., 0.]), random_prefix_frontier(40, 4, 2048, p_one=0.1)),
                       ('cart_pole_N40_random_prefix_p0.5_2048', np.array([0., 0., 1., 0.]), random_prefix_frontier(40, 4, 2048, p_one=0.5))):
        r, _ = _device_rate(c40.qp, x, f, dev)
        r['algorithmic_bytes_per_qp'] = c40.layout.bytes_per_qp()
        r['achieved_GBs'] = r['algorithmic_bytes_per_qp'] * r['nodes'] / (r['kernel_ms_avg'] * 1e-3) / 1e9
        r['kernel'] = 'hmpc_qp_kernel<4,7,4,...> (static row map, %d bytes of LDS per node)' % r['lds_bytes_per_wg']
        out[name] = r
    # a shape without a built-in instantiation: the register kernel compiled for it at hmpc_create (csrc/hmpc_jit.h)
    # against the run-time-sized kernel that served such shapes until round 4 (HMPC_JIT=0), same 2048-node frontier
    try:
        mldj, objj, x0j = random_mld(nx=6, nuc=2, nub=3, seed=3)
        Tj = 12
        cj = HybridModelPredictiveController(mldj, Tj, objj, None, backend=_NoBackend())
        fj = random_prefix_frontier(Tj, 3, 2048, p_one=0.3)
        fj[0, :] = -1
        spec = HipBatchedQP(cj.problem_data())
        os.environ['HMPC_JIT'] = '0'
        try:
            gen = HipBatchedQP(cj.problem_data())
        finally:
            del os.environ['HMPC_JIT']
        rs, st_s = _device_rate(spec, x0j, fj, dev)
        rg, st_g = _device_rate(gen, x0j, fj, dev)
        # (status ARRAYS, node by node -- round 4 compared counts of optimal / infeasible nodes and missed that the kernel
        # compiled for the problem left one node undecided and four unpolished; both figures go to `parity_flags`)
        out['generic_vs_specialised'] = {'system': 'random MLD nx=6 nu=2+3 N=12, 2048 random prefixes (p_one 0.3)', 'kernel_kinds_specialised': list(spec.kernel_info()),
                                         'kernel_kinds_generic': list(gen.kernel_info()), 'specialised': rs, 'generic': rg,
                                         'speedup': rs['qp_per_s'] / rg['qp_per_s'], 'statuses_equal': bool(np.array_equal(st_s, st_g)),
                                         'polished_equal': rs['polished'] == rg['polished'], 'compiled_kernels_dropped': int(spec.jit_stats()[0])}
    except Exception as e:
        out['generic_vs_specialised'] = {'error': repr(e)}
    # configs[4]: frontier = prefixes of a dive to a feasible leaf, every other one with a flipped binary (random
    # prefixes are all infeasible for this generator), tiled
    mld, objective, x0 = random_mld()
    T, nub, nx = 30, 8, 20
    c4 = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    c4.qp = HipBatchedQP(c4.problem_data())
    Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
    leaf = np.full((1, T * nub), -1, np.int8)
    for t in range(T):
        rr = c4.qp.solve_batch(x0, leaf)
        leaf[0, t * nub:(t + 1) * nub] = (rr['primal'][0][:(T + 1) * nx].reshape(T + 1, nx)[t] @ Cj.T >= 0)
    f = dive_frontier(leaf[0], 4096, 0)
    r, _ = _device_rate(c4.qp, x0, f, dev, reps=3, warm=1)
    r['algorithmic_bytes_per_qp'] = c4.layout.bytes_per_qp()
    r['achieved_GBs'] = r['algorithmic_bytes_per_qp'] * r['nodes'] / (r['kernel_ms_avg'] * 1e-3) / 1e9
    r['kernel'] = ('hmpc_qp_kernel<-1,...> (generic, streaming form: factor in a global slab; panel elimination on wave 0, tiles on the matrix cores), '
                   'compiled with this problem\'s sizes at hmpc_create (csrc/hmpc_jit.h)')
    r['kernel_kinds'] = list(c4.qp.kernel_info())
    r['frontier'] = '4096 distinct nodes: prefixes of a dive to a feasible leaf, every other one with one binary flipped (random prefixes of this generator are all infeasible)'
    out['random_mld_nx20_nu14_N30_dive_frontier_4096'] = r
    # the same launch on the shipped run-time-sized build of the streaming form (what served the problem until round 4)
    os.environ['HMPC_JIT_SIZED'] = '0'
    try:
        plain = HipBatchedQP(c4.problem_data())
    finally:
        del os.environ['HMPC_JIT_SIZED']
    r, _ = _device_rate(plain, x0, f, dev, reps=2, warm=1)
    r['kernel_kinds'] = list(plain.kernel_info())
    out['random_mld_nx20_nu14_N30_dive_frontier_4096_run_time_sized_kernel'] = r
    del plain
    # the same problem with the parent -> child hand-down (hmpc_warm): the tree a dive leaves behind (prefix chain of the
    # leaf + one-flip siblings, 481 nodes, every node but the root with its parent in the frontier), tiled to 4096; the
    # parents' records come from one untimed cold pass (as replay_frontier_4096_handdown on the headline system)
    ft, pt = dive_tree(leaf[0])
    m = len(ft)
    reps_t = -(-4096 // m)
    ftile = np.tile(ft, (reps_t, 1))[:4096]
    ptile = np.concatenate([np.where(pt >= 0, pt + c * m, -1) for c in range(reps_t)])[:4096].astype(np.int32)
    for key, par in (('random_mld_nx20_nu14_N30_dive_tree_4096', None), ('random_mld_nx20_nu14_N30_dive_tree_4096_handdown', ptile)):
        r, _ = _device_rate(c4.qp, x0, ftile, dev, reps=3, warm=1, parent=par)
        r['algorithmic_bytes_per_qp'] = c4.layout.bytes_per_qp()
        r['frontier'] = 'dive tree: the %d nodes a depth-first dive has solved at its first leaf (prefix chain + siblings), tiled to 4096' % m
        out[key] = r
    return out


def offline_lps():
    """the terminal ingredients of the cart-pole controller (mcais.py:44-184, controller.py:186-227)
    on the batched LP kernel -- one launch per sweep of the reference's one-LP-at-a-time loops.  Wall time of the whole
    construction (host copies and Python included); cpu_baseline: the same calls on the LP oracle, one host core."""
    from helpers import load_fixture
    from warm_start_hmpc_amd import terminal_set as ts
    from warm_start_hmpc_amd.qp_backend import lp_solve_batch
    d = load_fixture('cart_pole_with_walls')
    A_cl, D = d['A'] + d['B'][:, :1].dot(d['K']), d['F'] + d['G'][:, :1].dot(d['K'])
    count = {'lps': 0, 'launches': 0}

    def counted(lp):
        def call(A, c, b, relax=None):
            r = lp(A, c, b, relax=relax)
            count['lps'] += r['obj'].size
            count['launches'] += 1
            return r
        return call

    def build(lp):
        F_T, h_T = ts.mcais(A_cl, D, d['h'], lp=lp)
        M = ts.update_mu(d['F'], d['G'], d['h'], np.vstack((d['F'], F_T.dot(d['A']))), np.vstack((d['G'], F_T.dot(d['B']))), lp=lp)
        return F_T, M

    build(lp_solve_batch)                       # warm-up (module load)
    t0 = time.perf_counter()
    F_T, M = build(counted(lp_solve_batch))
    gpu_s = time.perf_counter() - t0
    out = {'lps': count['lps'], 'launches': count['launches'], 'facets': int(F_T.shape[0]), 'wall_ms': gpu_s * 1e3,
           'lps_per_sec': count['lps'] / gpu_s, 'same_set_as_fixture': bool(F_T.shape == d['F_T'].shape and np.allclose(F_T, d['F_T'], atol=1e-12))}
    try:
        from oracle.oracle_lp import lp_solve_batch as cpu_lp     # cpu_baseline leg
        t0 = time.perf_counter()
        build(cpu_lp)
        cpu_s = time.perf_counter() - t0
        out['cpu_baseline'] = {'wall_ms': cpu_s * 1e3, 'lps_per_sec': count['lps'] / cpu_s, 'cores': 1, 'kind': 'port'}
    except Exception as e:
        out['cpu_baseline'] = {'error': repr(e)}
    return out


def mpc_steps_per_sec(ctrl, steps=10, sims=64):
    """Closed-loop MPC steps/s (warm-started B&B, sigma = 0.001), the second figure of BASELINE.json's
    metric: (a) one loop alone (latency bound: a handful of sequential B&B rounds per step) and (b) `sims`
    independent loops advanced in lockstep, their rounds sharing kernel launches (the Monte-Carlo shape of
    the reference's statistical_analysis.py).  Host-pointer API, Python B&B and warm-start shift included;
    step 0 (the cold start) is excluded as in the reference's tables."""
    from warm_start_hmpc_amd.batched import BatchedMPC
    from helpers import load_fixture
    x_max = load_fixture('cart_pole_with_walls')['x_max']
    bm = BatchedMPC(ctrl)
    out = {}
    for label, seeds in (('single_loop', (0,)), ('lockstep_%d_loops' % sims, tuple(range(sims)))):
        warm = bm.closed_loop(np.array([0., 0., 1., 0.]), 1, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=8)
        t0 = time.perf_counter()
        st = bm.closed_loop(np.array([0., 0., 1., 0.]), steps + 1, e_sd=0.001, seeds=seeds, x_max=x_max, frontier_width=8)
        dt = time.perf_counter() - t0 - warm['wall']          # subtract one cold-start step
        ws = np.array([v[1:] for v in st['nodes_ws']])
        out[label] = {'value': len(seeds) * steps / dt, 'warm_solves_per_step_mean': float(ws.mean()),
                      'cover_min_max': [int(min(min(v) for v in st['len_ws'])), int(max(max(v) for v in st['len_ws']))]}
    # (c) one loop through the reference-shaped API (controller.feedback = feedforward + construct_warm_start),
    # with speculative expansion: the descendants of every selected node down to one stage of binaries are
    # solved in the same launch, so a warm-started step needs a couple of launches instead of one per level
    for label, depth in (('single_loop_feedback_api', 0), ('single_loop_feedback_api_speculative', ctrl.mld.nub)):
        best = None
        for attempt in range(3):   # best of 3: a lone loop leaves the GPU idle between launches and its clocks drift
            np.random.seed(0)
            x, ws, dt, rounds, solves = np.array([0., 0., 1., 0.]), None, 0., [], []
            for k in range(steps + 1):
                e = 0.001 * np.random.randn(4) * x_max
                st = {}
                t0 = time.perf_counter()
                u, ws, info = ctrl.feedback(x, warm_start=ws, e0=e, frontier_width=8, speculation_depth=depth, stats=st)
                if k > 0:
                    dt += time.perf_counter() - t0
                    rounds.append(st['rounds'])
                    solves.append(info['qp_solves'])
                x = info['x1']
            if best is None or dt < best[0]:
                best = (dt, rounds, solves)
        dt, rounds, solves = best
        out[label] = {'value': steps / dt, 'launches_per_step_mean': float(np.mean(rounds)),
                      'warm_solves_per_step_mean': float(np.mean(solves)), 'sample': 'best of 3 runs of %d steps' % steps}
    # (d) the C++ fleet driver (hmpc_fleet_*, csrc/hmpc_fleet.hip): trees behind the handle, multiplier rows resident in
    # HBM, one call per step for all loops; same disturbances for every driver (sigma = 0.001, seed = loop index)
    from warm_start_hmpc_amd.fleet import FleetMPC
    # (one loop: the cold start with dive prediction -- speculation < 0 --, warm steps with the subtree of the entering stage)
    for K, spec, hand in ((1, 4, True), (1, 4, False), (64, 2, True), (256, 0, True), (1024, 0, True), (1024, 0, False)):
        errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in range(K)])
        progress('fleet of %d loops (hand-down %s)' % (K, hand))
        fl = FleetMPC(ctrl, K, handdown=hand)
        kw = dict(frontier_width=8, speculation=spec, cold_speculation=-1 if K == 1 else spec, cold_frontier_width=2 if K == 1 else 8)
        fl.closed_loop(np.array([0., 0., 1., 0.]), 2, errs[:, :2], **kw)   # warm-up (allocations)
        sa = fl.stats()
        cold = fl.closed_loop(np.array([0., 0., 1., 0.]), 1, errs[:, :1], **kw)
        s0 = fl.stats()
        st = fl.closed_loop(np.array([0., 0., 1., 0.]), steps + 1, errs, **kw)
        s1 = fl.stats()
        if K >= 256:        # (a step of many loops is a handful of launches and host phases of a millisecond: the run is taken twice, the faster one counts)
            st2 = fl.closed_loop(np.array([0., 0., 1., 0.]), steps + 1, errs, **kw)
            if st2['wall'] - st2['wall_first_step'] < st['wall'] - st['wall_first_step']:
                st = st2
        warm_only = {k: ((s1[k] - s0[k]) - (s0[k] - sa[k])) / float(steps) for k in ('rounds', 'launched', 'handed')}   # (the run of steps + 1 steps minus its cold step)
        dt = st['wall'] - st['wall_first_step']                                                   # the warm steps of the run (its own cold start taken out)
        out['fleet_%d_loops%s' % (K, '' if hand else '_no_handdown')] = {
            'value': K * steps / dt, 'warm_solves_per_step_mean': float(st['nodes_ws'][:, 1:].mean()),
            'cover_min_max': [int(st['len_ws'].min()), int(st['len_ws'].max())], 'speculation_depth': spec,
            'cold_start': 'dive prediction' if K == 1 else 'as the warm steps', 'handdown': hand, 'handed_down_verified_per_step': (s1['handed'] - s0['handed']) / (K * (steps + 1.0)),
            'cold_step_ms': 1e3 * cold['wall'], 'warm_step_ms': 1e3 * dt / steps,
            'launches_per_step_incl_cold_mean': (s1['rounds'] - s0['rounds']) / (steps + 1.0),
            'per_warm_step': {'launches': warm_only['rounds'], 'nodes_launched_per_loop': warm_only['launched'] / K,
                              'handed_down_verified_per_loop': warm_only['handed'] / K},
            'sample': 'the faster of two runs of %d steps' % (steps + 1) if K >= 256 else 'one run of %d steps' % (steps + 1),
            'driver': 'hmpc_fleet_* (C++, trees behind the handle, multiplier rows resident in HBM)'}
        del fl
    out['note'] = 'closed loop sigma=0.001, warm-started B&B, frontier_width=8, reference (published, Gurobi): 26.8 steps/s'
    return out


def fleet_shard(total_loops, world, rank):
    """Simulations of a study of `total_loops` closed loops that rank `rank` of `world` runs: s = rank, rank + world, ...
    (simulation s on rank s mod world -- monte_carlo.py; the loops of the reference's study are independent)."""
    return list(range(rank, total_loops, world))


def sharded_fleet_rate(ctrl, world, rank, barrier, dist, dev, total_loops=1024, steps=10, fleet_factory=None, T_state=None):
    """MPC steps/s of `total_loops` closed loops sharded over the ranks by simulation (simulation s on rank s mod world, as
    monte_carlo.py and the reference's study, statistical_analysis.py:93-196): one fleet (hmpc_fleet_*) per rank, no
    communication inside the timed region.  Every rank returns; the figure is the same on all (sum of steps / max wall)."""
    import torch
    from helpers import load_fixture
    x_max = load_fixture('cart_pole_with_walls')['x_max']
    sims = fleet_shard(total_loops, world, rank)
    K = len(sims)
    errs = np.array([0.001 * np.random.RandomState(s).randn(steps + 1, 4) * x_max for s in sims])
    if fleet_factory is None:
        from warm_start_hmpc_amd.fleet import FleetMPC
        fl = FleetMPC(ctrl, K, handdown=True)
    else:
        fl = fleet_factory(ctrl, K)            # (the CPU rehearsal of tests/test_distributed.py: the lockstep driver on the oracle)
    kw = dict(frontier_width=8, speculation=0, cold_speculation=0, cold_frontier_width=8)
    x0 = np.array([0., 0., 1., 0.]) if T_state is None else np.asarray(T_state, dtype=np.float64)
    fl.closed_loop(x0, 2, errs[:, :2], **kw)              # warm-up (allocations)
    barrier()
    t0 = time.perf_counter()
    cold = fl.closed_loop(x0, 1, errs[:, :1], **kw)       # the cold-start step alone ...
    barrier()
    t_cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    st = fl.closed_loop(x0, steps + 1, errs, **kw)         # ... and the same step followed by `steps` warm-started ones
    barrier()
    t_all = time.perf_counter() - t0
    walls = torch.tensor([t_cold, t_all, float(K), float(st['nodes_ws'][:, 1:].sum()), float(st['len_ws'].min()), float(st['len_ws'].max())],
                         dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    mx, sm = walls.clone(), walls.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    mn = walls.clone()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    loops = int(sm[2].item())
    dt = float(mx[1].item()) - float(mx[0].item())
    del fl
    return {'value': loops * steps / dt, 'unit': 'MPC steps/s (whole job)', 'loops_total': loops, 'loops_per_rank': K, 'steps_timed': steps,
            'warm_step_ms': 1e3 * dt / steps, 'cold_step_ms': 1e3 * float(mx[0].item()),
            'warm_solves_per_step_mean': float(sm[3].item()) / (loops * steps), 'cover_min_max': [int(mn[4].item()), int(mx[5].item())],
            'simulations_of_this_rank': sims[:4] + (['...'] if len(sims) > 4 else []),
            'sharding': 'simulation s on rank s mod N, one fleet (hmpc_fleet_*) per rank, no collective inside the timed region',
            'timing': 'barrier + synchronize on both sides, max over the ranks; cold-start step timed apart and subtracted'}


def parity_flags(line):
    """Everything in a bench line that says a kernel did not do what its sibling or the contract says: nodes left undecided
    (`not_converged` > 0 anywhere), status arrays of two kernels on one workload that differ (`statuses_equal` false), polished
    counts that differ, compiled kernels the nets dropped.  `ok` is what __graft_entry__.smoke() and the tests hold."""
    undecided, unequal, dropped = {}, [], {}

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                here = path + [str(k)]
                if k == 'not_converged' and isinstance(v, (int, float)) and v > 0:
                    undecided['.'.join(path)] = int(v)
                elif k in ('statuses_equal', 'polished_equal') and v is False:
                    unequal.append('.'.join(here))
                elif k == 'compiled_kernels_dropped' and isinstance(v, (int, float)) and v > 0:
                    dropped['.'.join(path)] = int(v)
                else:
                    walk(v, here)
    walk(line, [])
    return {'ok': not undecided and not unequal and not dropped, 'not_converged': undecided, 'statuses_or_polished_counts_differ': unequal,
            'compiled_kernels_dropped': dropped}


def real_tree_frontier(ctrl, B, rank, x_max, spread=0.05, x_center=None):
    """B nodes of real branch-and-bound trees: cold-started searches from x0 = [0, 0, 1, 0] + spread * N(0, 1) * x_max
    (seed: rank, tree), every node they solve plus their leaves, until B nodes are collected; spread = 0: the one tree
    of the nominal state, tiled (SURVEY 8(d) C2 as written).  Returns (x0 [B, nx], fix [B, T nub], parent [B]: row of the
    node's parent in this frontier or -1)."""
    x_center = np.array([0., 0., 1., 0.]) if x_center is None else np.asarray(x_center, dtype=np.float64)
    if spread == 0:
        from helpers import real_tree_with_parents
        fix1, par1 = real_tree_with_parents(ctrl, x_center, leaves_too=True, frontier_width=8)
        reps = B // len(fix1) + 1
        fix = np.tile(fix1, (reps, 1))[:B]
        parent = np.concatenate([np.where(par1 >= 0, par1 + r * len(fix1), -1) for r in range(reps)])[:B].astype(np.int32)
        return np.ascontiguousarray(np.repeat(x_center[None], B, axis=0)), np.ascontiguousarray(fix), parent
    from helpers import real_tree_with_parents
    xs, fixes, parents, n, j = [], [], [], 0, 0
    while n < B:
        rng = np.random.RandomState(7919 * rank + j)
        x0 = x_center + (spread * rng.randn(4) * x_max if (rank or j) else 0.)
        j += 1
        fix, parent = real_tree_with_parents(ctrl, x0, leaves_too=True, frontier_width=8)
        if not len(fix):
            continue
        parents.append(np.where(parent >= 0, parent + n, -1))
        fixes.append(fix)
        xs.append(np.repeat(x0[None], len(fix), axis=0))
        n += len(fix)
    fix, parent, x0 = np.concatenate(fixes)[:B], np.concatenate(parents)[:B].astype(np.int32), np.concatenate(xs)[:B]
    return np.ascontiguousarray(x0), np.ascontiguousarray(fix), parent


def dive_frontier(leaf, count, seed):
    """BASELINE configs[4] frontier: `count` DISTINCT nodes -- prefixes (random depth) of a dive to a feasible leaf, every
    other one with one of its fixed binaries flipped (random prefixes of the random MLD are all infeasible)."""
    rng = np.random.default_rng(seed)
    n = leaf.size
    seen, rows = {bytes(np.full(n, -1, np.int8))}, [np.full(n, -1, np.int8)]
    while len(rows) < count:
        d = int(rng.integers(1, n + 1))
        row = np.full(n, -1, np.int8)
        row[:d] = leaf[:d]
        if rng.random() < 0.5 or len(seen) > n // 2:   # (there are only n distinct prefixes without a flip)
            j = int(rng.integers(0, d))
            row[j] = 1 - row[j]
        if row.tobytes() not in seen:
            seen.add(row.tobytes())
            rows.append(row)
    return np.array(rows)


def dive_tree(leaf):
    """The tree a dive leaves behind: the chain of prefixes of the leaf (the root first) and, beside every prefix but the
    root, its sibling -- the same prefix with its last binary flipped.  Returns (nodes [2 n + 1, n], parent index of each,
    -1 for the root): what a depth-first search has solved when it reaches its first leaf, and the parents whose
    records the hand-down gives to the children."""
    n = leaf.size
    rows, parent = [np.full(n, -1, np.int8)], [-1]
    for d in range(1, n + 1):
        p = np.full(n, -1, np.int8)
        p[:d] = leaf[:d]
        q = p.copy()
        q[d - 1] = 1 - q[d - 1]
        up = 2 * (d - 1) - 1 if d > 1 else 0          # the prefix one shorter (rows: root, then (prefix, sibling) pairs)
        rows += [p, q]
        parent += [up, up]
    return np.array(rows), np.array(parent, dtype=np.int32)


def shard(total, world, rank):
    """Strong scaling (BASELINE configs[2]): node k of a frontier of `total` nodes goes to rank k mod world."""
    return np.arange(rank, total, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frontier', type=int, default=4096, help='nodes per GPU per step')
    ap.add_argument('--frontier-kind', default='real_tree', choices=('real_tree', 'random_prefix'),
                    help='real_tree: the nodes real searches solve (default); random_prefix: SURVEY 8(d) C2 generator with --p-one')
    ap.add_argument('--states', default='nominal', choices=('nominal', 'distinct'),
                    help='real_tree: the tree of x0 = [0, 0, 1, 0] tiled (SURVEY 8(d) C2), or the trees of distinct closed-loop states')
    ap.add_argument('--p-one', type=float, default=0.5)
    ap.add_argument('--handdown', action='store_true',
                    help='hand every node the record of its parent (hmpc_warm): the parents come from one untimed cold pass')
    ap.add_argument('--frontier-total', type=int, default=0,
                    help='strong scaling: this many nodes IN TOTAL, split over the ranks (BASELINE configs[2]: 1024 over 8 GPUs); '
                         'overrides --frontier')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary frontiers / configs / closed-loop figures')
    ap.add_argument('--workload', default='cart_pole_n20', choices=('cart_pole_n20', 'cart_pole_n40', 'random_mld'),
                    help='cart_pole_n20: the headline (BASELINE configs[1]/[2]); the others (configs[3], configs[4]) are for profiling '
                         'their kernels with the same harness -- their numbers are secondary keys of the default run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='N > 1 ranks all on cuda:0 with a gloo group: rehearses the multi-rank path on a one-GPU box '
                         '(the number it prints is not a scaling measurement)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Called the way the one-GPU run is called (`python bench.py --gpus N ...`): start the N ranks as a CHILD process --
        # one per GPU under torch.distributed.run, rendezvous on 127.0.0.1 -- before anything here touches the GPU (a
        # process that has initialised the GPU must never be replaced by another), relay rank 0's JSON line (the ranks
        # inherit stdout) and leave with the child's exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        progress('spawning %d ranks: %s' % (args.gpus, ' '.join(cmd)))
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))

    import torch
    import torch.distributed as dist
    from helpers import make_controller, random_prefix_frontier, load_fixture

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or call bench.py --gpus N without a launcher: it starts the ranks itself)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rehearse_on_one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    B = args.frontier
    if args.frontier_total > 0:
        if args.frontier_total % world:
            raise SystemExit('--frontier-total must be a multiple of the number of ranks')
        B = args.frontier_total // world
    if args.workload == 'random_mld':
        # BASELINE configs[4]: frontier = prefixes of a dive to a feasible leaf, every other one with a flipped binary
        from helpers import random_mld, _NoBackend
        from warm_start_hmpc_amd.controller import HybridModelPredictiveController
        from warm_start_hmpc_amd.qp_backend import HipBatchedQP
        mld, objective, x0_h = random_mld()
        ctrl = HybridModelPredictiveController(mld, 30, objective, None, backend=_NoBackend())
        ctrl.qp = HipBatchedQP(ctrl.problem_data(), device=local)
        T, nub = ctrl.T, ctrl.mld.nub
        Cj = np.array([mld.F[52 + 4 * j] for j in range(nub)])
        leaf = np.full((1, T * nub), -1, np.int8)
        for t in range(T):
            rr = ctrl.qp.solve_batch(x0_h, leaf)
            leaf[0, t * nub:(t + 1) * nub] = (rr['primal'][0][:(T + 1) * 20].reshape(T + 1, 20)[t] @ Cj.T >= 0)
        fix_h = dive_frontier(leaf[0], B, rank)
    else:
        ctrl = make_controller('cart_pole_with_walls', T=40 if args.workload == 'cart_pole_n40' else None, backend='hip', device=local)
        T, nub = ctrl.T, ctrl.mld.nub
        parent_h = None
        spread = 0.05 if args.states == 'distinct' else 0.
        if args.frontier_kind == 'real_tree':
            if args.frontier_total > 0:    # strong scaling: ONE frontier, node k to rank k mod world (every rank builds it)
                x0_all, fix_all, _ = real_tree_frontier(ctrl, args.frontier_total, 0, load_fixture('cart_pole_with_walls')['x_max'], spread=spread)
                mine = shard(args.frontier_total, world, rank)
                x0_h, fix_h = np.ascontiguousarray(x0_all[mine]), np.ascontiguousarray(fix_all[mine])
            else:                          # weak scaling: every rank its own trees (seeded by the rank)
                x0_h, fix_h, parent_h = real_tree_frontier(ctrl, B, rank, load_fixture('cart_pole_with_walls')['x_max'], spread=spread)
        else:
            # disjoint shards: rank r takes seeds 1000 + r*B .. 1000 + (r+1)*B - 1 (strong scaling: of the one frontier)
            fix_h = random_prefix_frontier(T, nub, B, p_one=args.p_one, seed0=1000 + rank * B)
            x0_h = np.array([0., 0., 1., 0.])
    fix = torch.from_numpy(fix_h).to(dev)
    x0 = torch.from_numpy(x0_h).to(dev)
    out = dict(obj=torch.empty(B, dtype=torch.float64, device=dev), dual_obj=torch.empty(B, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), iters=torch.empty(B, dtype=torch.int32, device=dev),
               primal=torch.empty(B, ctrl.qp.n_primal, dtype=torch.float64, device=dev),
               dual=torch.empty(B, ctrl.qp.n_dual, dtype=torch.float64, device=dev))
    fully_fixed = torch.from_numpy((fix_h >= 0).all(axis=1)).to(dev)
    ub = torch.full((1,), float('inf'), dtype=torch.float64, device=dev)
    warm = None
    if args.handdown and args.workload == 'cart_pole_n20' and args.frontier_kind == 'real_tree' and parent_h is not None:
        # one untimed cold pass produces the parents' records; a node is handed its parent's if that is an optimal vertex
        par = dict(out, primal=torch.empty_like(out['primal']), dual=torch.empty_like(out['dual']))
        ctrl.qp.solve_batch_device(x0, fix, par)
        torch.cuda.synchronize()
        st, itf = par['status'].cpu().numpy(), par['iters'].cpu().numpy()
        good = (parent_h >= 0) & (st[np.maximum(parent_h, 0)] == 0) & (((itf[np.maximum(parent_h, 0)] >> 16) & 1) > 0)
        warm = (par['primal'], par['dual'], torch.from_numpy(np.where(good, parent_h, -1).astype(np.int32)).to(dev))

    def step():
        ctrl.qp.solve_batch_device(x0, fix, out, warm=warm)
        # incumbent upper bound = best objective among binary-feasible (fully fixed) nodes,
        # shared over xGMI so that every rank prunes against the global best
        cand = torch.where(fully_fixed, out['obj'], torch.full_like(out['obj'], float('inf')))
        torch.minimum(ub, cand.min().reshape(1), out=ub)
        if world > 1:
            dist.all_reduce(ub, op=dist.ReduceOp.MIN)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    progress('rank %d: frontier of %d nodes built' % (rank, B))
    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        ctrl.qp.solve_batch_device(x0, fix, out, warm=warm)
        ev[k][1].record()
        cand = torch.where(fully_fixed, out['obj'], torch.full_like(out['obj'], float('inf')))
        torch.minimum(ub, cand.min().reshape(1), out=ub)
        if world > 1:
            dist.all_reduce(ub, op=dist.ReduceOp.MIN)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # BASELINE configs[2] as written (ONE frontier of 1024 nodes over the ranks, node k to rank k mod N; strong scaling)
    # beside the weak default, in the same multi-GPU run: same step (solve the shard, all-reduce the incumbent), same
    # bracketing, max over the ranks
    strong = None
    if world > 1 and args.frontier_total == 0 and args.workload == 'cart_pole_n20' and args.frontier_kind == 'real_tree' and 1024 % world == 0:
        xs_all, fs_all, _ = real_tree_frontier(ctrl, 1024, 0, load_fixture('cart_pole_with_walls')['x_max'], spread=spread)
        mine = shard(1024, world, rank)
        xs, fs = torch.from_numpy(np.ascontiguousarray(xs_all[mine])).to(dev), torch.from_numpy(np.ascontiguousarray(fs_all[mine])).to(dev)
        Bs = len(mine)
        outs = {k: v[:Bs] for k, v in out.items()}
        ffs = torch.from_numpy((fs_all[mine] >= 0).all(axis=1)).to(dev)
        ubs = torch.full((1,), float('inf'), dtype=torch.float64, device=dev)

        def strong_step():
            ctrl.qp.solve_batch_device(xs, fs, outs)
            cand = torch.where(ffs, outs['obj'], torch.full_like(outs['obj'], float('inf')))
            torch.minimum(ubs, cand.min().reshape(1), out=ubs)
            dist.all_reduce(ubs, op=dist.ReduceOp.MIN)
        for _ in range(args.warmup):
            strong_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            strong_step()
        barrier()
        ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        strong = {'frontier_total': 1024, 'nodes_per_gpu': Bs, 'ms_per_step': 1e3 * float(ts.item()) / args.steps,
                  'qp_per_s': 1024 * args.steps / float(ts.item()), 'scaling': 'strong',
                  'incumbent': float(ubs.item()), 'note': 'BASELINE configs[2]: 1024-node replay frontier sharded over the ranks, incumbent all-reduce every step'}

    # The other half of BASELINE's metric at N > 1: MPC steps/s of the closed-loop study (statistical_analysis.py:93-196:
    # independent simulations), sharded by simulation -- simulation s on rank s mod N, no communication (monte_carlo.py) --:
    # 1024 loops in total, one fleet of 1024 / N loops per rank, disturbances seeded by the GLOBAL simulation index; same
    # bracketing as the QP metric (barrier + synchronize on both sides, max over the ranks), steps summed over the ranks.
    fleet_sharded = None
    if world > 1 and args.workload == 'cart_pole_n20' and not args.no_secondary:
        fleet_sharded = sharded_fleet_rate(ctrl, world, rank, barrier, dist, dev)

    status = out['status'].cpu().numpy()
    raw_iters = out['iters'].cpu().numpy()
    iters = raw_iters & 0xFFFF                           # bits 16..18 flag polished / weak / handed-down records
    if rank == 0:
        bytes_per_qp = ctrl.layout.bytes_per_qp()
        value = world * B * args.steps / elapsed
        achieved = bytes_per_qp * B / (kernel_ms * 1e-3) / 1e9
        grid, lds = ctrl.qp.launch_info()
        # HBM bytes per launch from the PMC passes of this same command (FETCH_SIZE x2 + WRITE_SIZE, gfx950
        # corrections of MI355X_MICROARCH.md); counters need their own rocprofv3 runs, so the figure is read
        # from the committed summary (profiles/collect.sh + profiles/summarise.py), valid for the default frontier
        traffic, traffic_src = None, None
        try:
            if B == 4096 and args.frontier_kind == 'real_tree' and args.states == 'nominal' and not args.handdown and args.workload == 'cart_pole_n20':
                with open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')) as fh:
                    traffic = json.load(fh)['hbm_traffic_bytes_per_launch']
                traffic_src = 'profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'
        except (OSError, KeyError, ValueError):
            pass
        polished = int(((raw_iters >> 16) & 1).sum())
        kind = ('%s%s' % ('replay frontier of SURVEY 8d C2: every node a cold-started search from x0 = [0, 0, 1, 0] solves plus its leaves, tiled'
                          if args.states == 'nominal' else 'nodes of real trees from distinct closed-loop states (every node cold-started searches '
                          'from [0, 0, 1, 0] + 0.05 N(0,1) x_max solve, plus their leaves)',
                          ', each handed its parent\'s record (hmpc_warm)' if warm is not None else ', each solved cold')
                if args.frontier_kind == 'real_tree' else 'random-prefix frontier (SURVEY 8d C2), p_one=%.2f' % args.p_one)
        line = {
            'metric': 'QP subproblems/sec, cart-pole-with-walls N=20 synthetic frontier' if args.workload == 'cart_pole_n20' else 'QP subproblems/sec, ' + args.workload,
            'value': value, 'unit': 'QP subproblems/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': ('strong' if args.frontier_total > 0 else 'weak') if not args.rehearse_on_one_gpu else 'rehearsal: all ranks on one GPU, not a measurement',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'rccl_ranks': (dist.get_world_size() if world > 1 and dist.get_backend() == 'nccl' else (0 if world > 1 else 1)),
            'config': {'workload': ('cart_pole_with_walls N=%d, 4 binaries/step, %s' % (T, kind))
                       if args.workload != 'random_mld' else 'random MLD nx=20 nu=6+8 N=30 (SURVEY 8d C4), dive frontier',
                       'frontier_nodes_per_gpu': B, 'x0': x0_h.tolist() if x0_h.ndim == 1 else ([0., 0., 1., 0.] if args.states == 'nominal' and args.frontier_kind == 'real_tree'
                                                                                                     else 'one initial state per tree: [0, 0, 1, 0] + 0.05 N(0,1) x_max'),
                       'parallelism': 'frontier sharded by node, '
                       'one RCCL all-reduce(min) of the incumbent per step' if world > 1 else 'single GPU',
                       'solver': 'HSDE interior point + Riccati, tol 1e-8, lazy terminal set, <= 2 refinement steps, active-set polish'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': bytes_per_qp * B,
                         'algorithmic_bytes_per_qp': bytes_per_qp, 'kernel': 'hmpc_qp_kernel', 'kernel_kinds_1_2_4_waves': list(ctrl.qp.kernel_info()), 'ilp_schedule_1_2_4_waves': list(ctrl.qp.kernel_recipe()),
                         'kernel_ms_avg': kernel_ms, 'grid': grid, 'lds_bytes_per_wg': lds},
            # secondary: algorithmic flops (per interior-point iteration T (7/3) (nx+nu)^3 + 2 nnz(A_c) (nx+nu)
            # = 1.2e5 for this workload) against the f64 vector peak (half of MI355X_MICROARCH.md's 157.3 TFLOP/s FP32)
            'flops': {'per_iteration': 1.2e5, 'iterations_per_qp': float(iters.mean()),
                      'achieved_TFLOPs': 1.2e5 * float(iters.mean()) * B / (kernel_ms * 1e-3) / 1e12, 'peak_TFLOPs_f64_vector': 78.6,
                      'frac': 1.2e5 * float(iters.mean()) * B / (kernel_ms * 1e-3) / 1e12 / 78.6,
                      'note': 'neither bandwidth nor flops bound: sequential stage recursions, one wave per SIMD (DESIGN.md 5)'},
            'nodes': {'optimal': int((status == 0).sum()), 'infeasible': int((status == 1).sum()),
                      'not_converged': int((status > 1).sum()), 'polished': polished, 'ipm_iters_mean': float(iters.mean()),
                      'ipm_iters_mean_optimal': float(iters[status == 0].mean()) if (status == 0).any() else None,
                      'handed_down_verified': int(((raw_iters >> 18) & 1).sum())},
        }
        if strong is not None:
            line['configs2_strong_scaling_1024'] = strong
        if fleet_sharded is not None:
            line['mpc_steps_per_sec'] = {'fleet_sharded': fleet_sharded,
                                         'note': 'closed loop sigma=0.001, warm-started B&B, simulations sharded over the ranks (simulation s on rank s mod N), reference (published, Gurobi): 26.8 steps/s'}
        if world == 1 and args.workload == 'cart_pole_n20':
            try:  # second kernel of the path (HBM bound), a few milliseconds
                line['warm_start_shift'] = shift_bandwidth(ctrl, dev)
            except Exception as e:
                line['warm_start_shift'] = {'error': str(e)}
        progress('timed region done: %.3f ms per step' % (1e3 * elapsed / args.steps))
        if not args.no_cpu_baseline and args.workload != 'random_mld':
            # (rank 0 only; at N > 1 the other ranks have finished their timed regions and wait in the final barrier)
            line['cpu_baseline'] = cpu_baseline(ctrl, x0_h, fix_h)
            progress('cpu baseline done')
        if world == 1 and not args.no_secondary and not args.no_cpu_baseline and args.workload == 'cart_pole_n20':
            for key, fn in (('frontiers', lambda: secondary_frontiers(ctrl, dev, load_fixture('cart_pole_with_walls')['x_max'])), ('other_configs', lambda: other_configs(dev)),
                            ('mpc_steps_per_sec', lambda: mpc_steps_per_sec(ctrl)), ('offline_lps', offline_lps)):
                try:
                    line[key] = fn()
                except Exception as e:  # secondary figures never hide the main line
                    line[key] = {'error': repr(e)}
                progress('%s done' % key)
        line['nodes']['compiled_kernels_dropped'] = int(ctrl.qp.jit_stats()[0])
        line['parity_flags'] = parity_flags(line)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
