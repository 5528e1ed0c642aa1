"""The tree search and the warm-start construction against traces recorded on the REFERENCE'S OWN code.

tests/golden/bb_traces.npz (made by tests/golden/make_bb_traces.py in the build container) holds what the reference's
`branch_and_bound` (warm_start_hmpc/branch_and_bound.py:408-499), its selection rules (:501-563), `branch_in_time` and
`_brancher` (controller.py:13-44, 395-429) and `construct_warm_start` (controller.py:431-721) did -- imported from
/root/reference and run unmodified -- when driven through the reference's solver callback by this repository's CPU QP
oracle: the identifiers in the order they were solved, the leaves with their bounds, the incumbent, and the warm start
built for the next step, for six cases (N = 10 / 20 / 40, with and without terminal set, one-wall system, best-first
and depth-first) over one cold and two warm-started MPC steps each.

Here this repository's driver (`frontier_width=1`), brancher and warm-start construction replay the same cases:
  * on the oracle backend (CPU) everything must be EQUAL -- solve order, leaves, bounds bit for bit (same QP records,
    same arithmetic), the shifted multipliers to rounding (the reference sums the pi terms in another association);
  * on the HIP backend (GPU) the incumbent (cost, binary assignment) must be equal and the search within a few solves
    and leaves of the trace: multipliers of dependent active rows are not unique, so kernel and oracle -- like any two
    solvers -- may meet equal bounds in another order (see the comment in _replay_case).  What the HIP path does is
    pinned exactly (HIP_PINS: the trace's own numbers on 11 of 16 steps), and where its cover has the trace's size it
    must be the trace's cover.
"""
import numpy as np
import pytest

from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.branch_and_bound import best_first, depth_first

CASES = ('n10', 'n10free', 'n20', 'n20depth', 'n40', 'onewall')
# What the HIP path does on these cases, pinned (the +-3 envelope alone pinned the solve order to
# nothing): (solves, leaves, size of the next cover) per case and step, as measured on the kernels of round 5 under BOTH
# instruction schedules (profiles/r05_bb_deviation.txt) -- equal to the reference's trace on 11 of 16 steps (every step of n20,
# n20depth and n40), one solve off on n10 (81 / 9 against 80 / 10), one node more on the one-wall system (157 / 102 / 100 against
# 156 / 101 / 99, carried through its warm start), the tie of n10free.  The cause is the non-unique multipliers of dependent
# active rows (see _replay_case); a kernel change that moves these numbers has changed which optimal multiplier the polish
# returns somewhere, and has to say so here.
HIP_PINS = {('n10', 0): (81, 41, 37), ('n10', 1): (9, 41, 37), ('n10', 2): (9, 41, None),
            ('n10free', 0): (87, 45, None),
            ('n20', 0): (160, 81, 77), ('n20', 1): (12, 81, 77), ('n20', 2): (11, 81, None),
            ('n20depth', 0): (161, 81, 77), ('n20depth', 1): (13, 81, 77), ('n20depth', 2): (11, 81, None),
            ('n40', 0): (320, 161, 157), ('n40', 1): (43, 162, 158), ('n40', 2): (42, 164, None),
            ('onewall', 0): (157, 102, 100), ('onewall', 1): (147, 193, 191), ('onewall', 2): (9, 194, None)}
RULES = {'best_first': best_first, 'depth_first': depth_first}


def _fix_rows(ctrl, nodes):
    return np.array([ctrl._fix_vector(n.identifier) for n in nodes]).reshape(len(nodes), ctrl.T * ctrl.mld.nub)


def _replay_case(tr, name, backend, exact):
    T, nub, terminal = (int(v) for v in tr[name + '_meta'])
    ctrl = make_controller(str(tr[name + '_fixture']), T=T, terminal=bool(terminal), backend=backend)
    rule = RULES[str(tr[name + '_rule'])]
    warm_start = None
    for s in range(3):
        key = '%s_s%d_' % (name, s)
        if key + 'order' not in tr.files:
            break
        x = tr[key + 'x0']
        order = []
        inner = ctrl.solve_frontier

        def recording(identifiers, x0, _inner=inner, **kw):
            order.extend(ctrl._fix_vector(i) for i in identifiers)
            return _inner(identifiers, x0, **kw)

        ctrl.solve_frontier = recording
        try:
            sol, leaves, solves, _ = ctrl.feedforward(x, search_rule=rule, warm_start=warm_start, printing_period=None,
                                                      frontier_width=1)
        finally:
            ctrl.solve_frontier = inner
        if not exact:
            # HIP backend: the multipliers of dependent active rows are not unique and kernel and
            # oracle -- like any two solvers, the reference's published counts wobble 158..161 for it -- may return
            # different optimal choices; child bounds (parent bound + multiplier) then meet in another order.  What must
            # hold: the same incumbent (cost and binary assignment), the search within a few solves and leaves of the
            # trace, the leaves a disjoint cover, the next cover of the same size.
            from kkt_checks import is_disjoint_cover
            assert abs(solves - int(tr[key + 'solves'])) <= 3, (name, s, solves, int(tr[key + 'solves']))
            assert abs(len(leaves) - len(tr[key + 'leaves_lb'])) <= 3
            assert (solves, len(leaves)) == HIP_PINS[(name, s)][:2], (name, s, (solves, len(leaves)), 'pinned', HIP_PINS[(name, s)][:2],
                                                                        'trace', (int(tr[key + 'solves']), len(tr[key + 'leaves_lb'])))
            assert is_disjoint_cover(ctrl, leaves)
            np.testing.assert_allclose(sol.objective, float(tr[key + 'cost']), rtol=1e-7, atol=1e-12)
            ub = np.concatenate(sol.variables['ub']).round().astype(np.int8)
            if name == 'n10free' and not np.array_equal(ub, tr[key + 'incumbent_fix']):
                # this MIQP has an exact tie (the damper binary of the last stages is free of charge without a terminal
                # set: test_gpu_parity.py::test_golden_vectors proves it): either assignment is the optimum, and the
                # trajectories of the two differ from here on
                break
            assert np.array_equal(ub, tr[key + 'incumbent_fix']), (name, s)
            if key + 'ws_fix' not in tr.files:
                break
            u0, e0 = tr[key + 'u0'], tr[key + 'e0']
            nuc = ctrl.mld.nu - ctrl.mld.nub
            np.testing.assert_allclose(np.concatenate((sol.variables['uc'][0], sol.variables['ub'][0])), u0, rtol=1e-5, atol=1e-6)
            ws, _, _ = ctrl.construct_warm_start(leaves, x, u0[:nuc], u0[nuc:], e0)
            assert abs(len(ws) - len(tr[key + 'ws_lb'])) <= 3
            assert len(ws) == HIP_PINS[(name, s)][2], (name, s, len(ws), HIP_PINS[(name, s)][2], len(tr[key + 'ws_lb']))
            if len(ws) == len(tr[key + 'ws_lb']):       # (where the cover has the trace's size it IS the trace's cover, and the same leaves reopen)
                assert np.array_equal(_fix_rows(ctrl, ws), tr[key + 'ws_fix']), (name, s)
                has_dual = np.array([n.extra.dual is not None for n in ws])
                assert np.mean(has_dual == tr[key + 'ws_has_dual']) >= 0.98, (name, s)
            warm_start = ws
            continue
        # the reference's driver solved the same nodes in the same order ...
        assert solves == int(tr[key + 'solves']), (name, s, solves, int(tr[key + 'solves']))
        assert np.array_equal(np.array(order), tr[key + 'order']), (name, s)
        # ... and ended with the same leaves, in the same list order, with the same bounds
        assert np.array_equal(_fix_rows(ctrl, leaves), tr[key + 'leaves_fix']), (name, s)
        lb, want = np.array([l.lb for l in leaves]), tr[key + 'leaves_lb']
        assert np.array_equal(np.isinf(lb), np.isinf(want))
        fin = np.isfinite(want)
        if exact:
            assert np.array_equal(lb[fin], want[fin]), (name, s, np.max(np.abs(lb[fin] - want[fin])))
            assert sol.objective == float(tr[key + 'cost'])
        else:
            np.testing.assert_allclose(lb[fin], want[fin], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(sol.objective, float(tr[key + 'cost']), rtol=1e-7, atol=1e-12)
        ub = np.concatenate(sol.variables['ub']).round().astype(np.int8)
        assert np.array_equal(ub, tr[key + 'incumbent_fix']), (name, s)              # binary assignment of the incumbent
        if key + 'ws_fix' not in tr.files:
            break
        # the next warm start: this repository's construct_warm_start against the reference's, same inputs
        u0, e0 = tr[key + 'u0'], tr[key + 'e0']
        nuc = ctrl.mld.nu - ctrl.mld.nub
        if not exact:                                      # (applied input: the recorded one, so that both shifts see the same data)
            np.testing.assert_allclose(np.concatenate((sol.variables['uc'][0], sol.variables['ub'][0])), u0, rtol=1e-5, atol=1e-6)
        ws, _, _ = ctrl.construct_warm_start(leaves, x, u0[:nuc], u0[nuc:], e0)
        assert np.array_equal(_fix_rows(ctrl, ws), tr[key + 'ws_fix']), (name, s)     # retain rule + identifier shift
        has_dual = np.array([n.extra.dual is not None for n in ws])
        lb, want = np.array([n.lb for n in ws]), tr[key + 'ws_lb']
        if exact:
            assert np.array_equal(has_dual, tr[key + 'ws_has_dual']), (name, s)      # the same leaves reopen
            assert np.array_equal(np.isinf(lb), np.isinf(want))
            fin = np.isfinite(want)
            np.testing.assert_allclose(lb[fin], want[fin], rtol=1e-12, atol=1e-13)   # pi-sum, model error, clipping
            dobj = np.array([n.extra.dual.objective if n.extra.dual is not None else 0. for n in ws])
            np.testing.assert_allclose(dobj, tr[key + 'ws_dobj'], rtol=1e-12, atol=1e-13)
            if key + 'ws_dual' in tr.files:                                          # the shifted multipliers themselves
                lay, cut = ctrl.layout, ctrl.layout.dual_slices()
                for j, n in enumerate(ws):
                    if n.extra.dual is None:
                        continue
                    row = np.zeros(lay.n_dual)
                    for k, blocks in cut.items():
                        for t, sl in enumerate(blocks):
                            row[sl] = n.extra.dual.variables[k][t]
                    np.testing.assert_allclose(row, tr[key + 'ws_dual'][j], rtol=1e-13, atol=1e-14)
        else:
            assert len(ws) == len(want)
            same = has_dual == tr[key + 'ws_has_dual']                                # (a proof within rounding of zero may fall either way)
            assert np.mean(same) >= 0.97, (name, s, np.mean(same))
            ok = same & np.isfinite(want) & np.isfinite(lb)
            np.testing.assert_allclose(lb[ok], want[ok], rtol=1e-5, atol=1e-7)
        # the next step starts from this warm start; on the oracle its bounds are set to the reference's (equal to
        # rounding, asserted above) so that both drivers search from identical numbers
        if exact:
            for j, n in enumerate(ws):
                n.lb = float(want[j])
        warm_start = ws
    return True


@pytest.mark.parametrize('name', CASES)
def test_driver_reproduces_the_reference_traces_on_the_oracle(name):
    assert _replay_case(load_fixture('bb_traces'), name, 'oracle', exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_driver_reproduces_the_reference_traces_on_the_gpu(name):
    assert _replay_case(load_fixture('bb_traces'), name, 'hip', exact=False)
