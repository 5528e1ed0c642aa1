"""The C++ fleet driver (hmpc_fleet_*, csrc/hmpc_fleet.hip: K closed loops in lockstep, trees behind the handle, multiplier
rows resident in HBM) against the numpy lockstep driver (batched.BatchedMPC) and the reference's published runs."""
import os

import numpy as np
import pytest

from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.batched import BatchedMPC

pytestmark = pytest.mark.gpu
X0 = np.array([0., 0., 1., 0.])


def test_fleet_walks_the_walk_of_the_numpy_driver():
    from warm_start_hmpc_amd.fleet import FleetMPC
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='hip')
    K, steps = 6, 12
    errors = ref['errors_0003'][:K, :steps]
    fl = FleetMPC(ctrl, K).closed_loop(X0, steps, errors, frontier_width=8)
    py = BatchedMPC(ctrl).closed_loop(X0, steps, seeds=tuple(range(K)), frontier_width=8, errors=errors)
    np.testing.assert_allclose(fl['costs'], np.array(py['costs']), rtol=1e-9, atol=1e-12)     # same optimum at every step
    assert np.array_equal(fl['len_ws'], np.array(py['len_ws']))                                 # same covers
    assert np.array_equal(fl['reopened'], np.array(py['reopened']))
    # solve counts: the same search up to the order in which equal bounds are met (the two drivers batch differently,
    # so a node may be solved by the 1-, 2- or 4-wave kernel: last-digit differences in the multipliers)
    assert np.max(np.abs(fl['nodes_ws'] - np.array(py['nodes_ws']))) <= 6
    assert abs(fl['nodes_ws'][:, 1:].mean() - np.array(py['nodes_ws'])[:, 1:].mean()) < 0.5


def test_fleet_replays_the_published_runs():
    # the reference's own disturbances (tests/golden/reference_closed_loop.npz), frontier_width 1 = its node order
    from warm_start_hmpc_amd.fleet import FleetMPC
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='hip')
    for tag in ('0001', '0003'):
        st = FleetMPC(ctrl, 12).closed_loop(X0, 50, ref['errors_' + tag][:12], frontier_width=1)
        assert st['steps'] == 600
        assert np.array_equal(st['len_ws'], ref['nodes_len_ws_' + tag][:12])                      # published cover sizes, every step
        ws, pws = st['nodes_ws'][:, 1:], ref['nodes_ws_' + tag][:12, 1:]
        calm = st['len_ws'][:, :-1] == 77
        assert np.median(ws[calm] - st['reopened'][:, :-1][calm]) == 9                         # dive + lost proofs (test_reference_replay.py)
        assert ws.mean() <= pws.mean() and ws.mean() >= 8.0
        assert abs(st['nodes_ws'][:, 0].mean() - ref['nodes_cs_' + tag][:12, 0].mean()) <= 3    # step 0 is a cold start


def test_speculative_expansion_in_the_fleet_changes_launches_not_results():
    from warm_start_hmpc_amd.fleet import FleetMPC
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='hip')
    errors = ref['errors_0003'][:2, :8]
    a, b = FleetMPC(ctrl, 2), FleetMPC(ctrl, 2)
    plain = a.closed_loop(X0, 8, errors, frontier_width=1)
    spec = b.closed_loop(X0, 8, errors, frontier_width=1, speculation=4, cold_speculation=4)
    np.testing.assert_allclose(spec['costs'], plain['costs'], rtol=1e-9, atol=1e-12)
    # (kernel variant differs with the batch size -- 1 / 2 / 4 waves per node --, and which infeasibility proofs survive a
    # shift depends on the ray: a leaf whose shifted dual objective lies at the margin is reopened by one variant and not
    # by the other; every reopened leaf is one more solve of the next step)
    assert np.array_equal(spec['len_ws'], plain['len_ws']) and np.max(np.abs(spec['reopened'] - plain['reopened'])) <= 2
    assert np.max(np.abs(spec['nodes_ws'] - plain['nodes_ws'])) <= 4
    assert b.stats()['rounds'] < a.stats()['rounds'] / 2 and b.stats()['launched'] > a.stats()['launched']
    # dive prediction (speculation < 0): the rest of a dive, predicted from the parent's rounded relaxed binaries, and the
    # sibling of every step ride along -- linear in the depth; the cold start takes a handful of launches
    c = FleetMPC(ctrl, 2)
    dive = c.closed_loop(X0, 8, errors, frontier_width=1, speculation=-1, cold_speculation=-1)
    np.testing.assert_allclose(dive['costs'], plain['costs'], rtol=1e-9, atol=1e-12)
    # (which infeasibility proofs survive a shift depends on the ray, and the ray of a node on whether it was handed its
    # parent's record and on the kernel variant the batch size selects: a proof within rounding of zero may fall either way)
    assert np.array_equal(dive['len_ws'], plain['len_ws']) and np.max(np.abs(dive['reopened'] - plain['reopened'])) <= 1
    assert np.max(np.abs(dive['nodes_ws'] - plain['nodes_ws'])) <= 4
    assert c.stats()['rounds'] < a.stats()['rounds'] / 3


def test_fleet_stops_a_loop_whose_miqp_is_infeasible_and_resets():
    from warm_start_hmpc_amd.fleet import FleetMPC
    ctrl = make_controller('cart_pole_with_walls', backend='hip')
    fl = FleetMPC(ctrl, 3)
    x0s = np.array([X0, [0., 0., 5., 0.], X0 * 0.5])          # the second state is outside the feasible set
    r = fl.solve(x0s)
    assert np.isfinite(r['cost'][0]) and np.isinf(r['cost'][1]) and np.isfinite(r['cost'][2])
    assert np.all(np.isnan(r['u0'][1]))
    cover, _ = fl.shift(np.zeros((3, 4)))
    assert cover[0] == 77 and cover[1] == 0
    r2 = fl.solve(np.array([r['x1'][0], X0, r['x1'][2]]))
    assert np.isinf(r2['cost'][1]) and r2['solves'][1] == 0       # ended loops stay ended ...
    fl.reset(1)
    r3 = fl.solve(np.array([r2['x1'][0], X0, r2['x1'][2]]))
    assert np.isfinite(r3['cost'][1]) and r3['solves'][1] > 150   # ... until reset: a cold start
    single = ctrl.feedforward(X0, printing_period=None)
    assert abs(r3['cost'][1] - single[0].objective) < 1e-9
    assert fl.stats()['rounds'] > 0


def test_incumbent_allreduce_through_the_c_abi():
    # hmpc_allreduce_incumbent (include/hmpc.h): RCCL communicator of one rank on this box -- the exchange must be the
    # identity; with N ranks the same call is MIN over (ub, -open) (warm_start_hmpc_amd/distributed.py is the tested
    # N-rank form, gloo world size 2 in tests/test_distributed.py)
    import ctypes
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='hip')
    lib, h = ctrl.qp.lib, ctrl.qp.handle
    uid = ctypes.create_string_buffer(128)
    assert lib.hmpc_comm_unique_id(uid) == 0, lib.hmpc_last_error()
    comm = ctypes.c_void_p()
    assert lib.hmpc_comm_create(h, 1, 0, uid, ctypes.byref(comm)) == 0, lib.hmpc_last_error()
    ub, n_open = ctypes.c_double(0.125), ctypes.c_int32(7)
    assert lib.hmpc_allreduce_incumbent(comm, ctypes.byref(ub), ctypes.byref(n_open)) == 0, lib.hmpc_last_error()
    assert ub.value == 0.125 and n_open.value == 7
    ub.value = float('inf')
    n_open.value = 0
    assert lib.hmpc_allreduce_incumbent(comm, ctypes.byref(ub), ctypes.byref(n_open)) == 0
    assert ub.value == float('inf') and n_open.value == 0
    assert lib.hmpc_comm_create(h, 2, 5, uid, ctypes.byref(ctypes.c_void_p())) == -1      # rank out of range
    assert lib.hmpc_comm_destroy(comm) == 0


def test_row_pools_of_a_fleet_that_is_never_shifted_stay_bounded():
    # a fleet that is reset and solved at every step, never shifted -- the cold searches of
    # fleet.closed_loop_study -- kept every multiplier row ever written (only hmpc_fleet_shift compacted).  Rows nobody
    # references are reclaimed now: hmpc_fleet_solve starts the pools from zero when every tree is cold; an ended loop
    # (hmpc_fleet_stop) holds nothing and is not searched.
    from warm_start_hmpc_amd.fleet import FleetMPC
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='hip')
    K = 4
    cold = FleetMPC(ctrl, K)
    xs = np.repeat(np.array([[0., 0., .5, 0.]]), K, axis=0)
    used = []
    for step in range(3):
        cold.reset()
        r = cold.solve(xs, 4)
        used.append(cold.rows()[0])
    assert used[0] > 0 and max(used[1:]) <= used[0] + 8      # the same searches: the same rows again, not more
    assert np.all(np.isfinite(r['cost'])) and np.all(r['solves'] > 0)
    cold.stop(2)                                            # loop 2 has ended
    for k in (0, 1, 3):
        cold.reset(k)
    r = cold.solve(xs, 4)
    assert r['solves'][2] == 0 and np.isinf(r['cost'][2]) and np.all(r['solves'][[0, 1, 3]] > 0)
    assert cold.rows()[0] <= used[0]                        # (the stopped loop holds no rows: the pools started from zero again)
    launched = cold.stats()['launched']
    for k in (0, 1, 3):
        cold.stop(k)
    cold.solve(xs, 4)                                       # nothing is running: nothing is launched
    assert cold.stats()['launched'] == launched


def test_incumbent_exchange_over_rccl_between_two_gpus(tmp_path):
    # Two FRESH processes, one per GPU, communicator from hmpc_comm_unique_id: MIN semantics, the -inf abort path, owner and
    # broadcast of the winning assignment over real RCCL (tests/rccl_two_ranks.py).  Needs two GPUs: on the one-GPU boxes of
    # this pool it skips -- the N-rank path is then covered by the gloo tests (test_distributed.py) and the one-rank RCCL
    # test above only.
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (this box has %d): RCCL between ranks is not exercised here' % torch.cuda.device_count())
    here = os.path.dirname(os.path.abspath(__file__))
    ident = str(tmp_path / 'rccl_id')
    procs = [subprocess.Popen([sys.executable, os.path.join(here, 'rccl_two_ranks.py'), str(r), '2', ident],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    seen = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    for s in seen:
        assert s['min'] == [2.0, 5]                                     # smallest bound, largest number of open candidates
        assert s['min_device'] == [2.0, 5] and s['poisoned'] == -1      # (HMPC_EINVAL on EVERY rank)
        assert s['abort'][0] == '-Infinity' or s['abort'][0] == float('-inf')
        assert s['publish'][0] == 2.0 and s['publish'][1] == 1 and s['publish'][2] == [1] * 40
        assert s['tie'][0] == 1.0 and s['tie'][1] == 0 and s['tie'][2] == [10] * 40
        assert s['none'][1] == -1
        assert s['root_status'] == 0


def test_incumbent_publication_over_rccl_with_one_rank(tmp_path):
    # the same script as the two-GPU test with a communicator of one rank: every entry point of the exchange goes through
    # RCCL on this box (all-reduce, all-reduce of the owner, broadcast), in a fresh process
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'rccl_two_ranks.py'), '0', '1', str(tmp_path / 'rccl_id')],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s = json.loads(r.stdout.strip().splitlines()[-1])
    assert s['min'] == [3.0, 5] and s['abort'][0] == float('-inf')
    assert s['min_device'] == [3.0, 5] and s['poisoned'] == -1
    assert s['publish'][:2] == [2.5, 0] and s['publish'][2] == [0] * 40
    assert s['tie'][:2] == [1.0, 0] and s['none'][1] == -1 and s['root_status'] == 0
