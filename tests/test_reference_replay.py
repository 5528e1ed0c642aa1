"""Replays the model errors of the reference's PUBLISHED closed-loop study through this repository's controller and
compares, step by step, with the node counts the reference recorded (tests/golden/reference_closed_loop.npz, made by
tests/golden/make_reference_data.py from notebooks/cart_pole_with_walls/data/{errors,nodes_cs,nodes_ws,nodes_len_ws}_sd_*.npy
and, for sd = 0.01 -- whose arrays are pickled and are not loaded --, from the text log data/solve_log_sd_0.010.log;
loop: notebooks/cart_pole_with_walls/statistical_analysis.py:93-196).  ALL published simulations are held: 100 x 50
steps at sd 0.001 and 0.003, the 109 started simulations of sd 0.01 (100 complete, 9 that leave the feasible set).

What is pinned -- at every step of every replayed simulation:
  * the size of the warm start (the shifted cover that survives the retain rule)  == published, including the
    excursions of sd 0.003 (77 ... 254 nodes);
  * the QP solves of the cold-started search                                       within 3 of published (the
    reference's own spread at equal states is 158..161: the order in which equal bounds are met depends on the last
    digits of the multipliers);
  * warm- and cold-started costs equal (np.isclose, the reference's own assertion, :171-173);
  * (sd 0.01) a simulation that left the feasible set in the published run leaves it here at the same step, with the
    same number of cold solves spent on proving it.
What differs and why: the warm-started search solves FEWER nodes here (mean 9.1 against 12.6 published).  A warm-started
step solves the dive through the binaries of the stage that enters the horizon (1 + 2 nub = 9 nodes) plus every leaf
whose infeasibility proof did not survive the shift (it is reopened with bound 0, controller.py:555-558).  Which Farkas
ray a solver returns is not unique; the rays of this solver carry no terminal-set multipliers (lazy terminal set) and
lose 0.2 proofs per step, Gurobi's lose about 3.6.  The tests assert exactly this decomposition, and that without the
lazy terminal set every proof is lost (about 75 reopened leaves per step)."""
import numpy as np
import pytest

from helpers import make_controller, load_fixture
from warm_start_hmpc_amd.batched import BatchedMPC

X0 = np.array([0., 0., 1., 0.])
# The ONE place in the 15 166 published steps where the published cover is not reproduced -- by oracle, kernel and fleet
# driver alike: at step 11 of simulation 94 (sd 0.003) the MIQP has a TIE -- two assignments of the last stages' binaries
# cost the same to 4e-10 relative (test_the_one_deviation_from_the_published_covers_is_a_tie below).  The dive meets the
# one that is worse in the tenth digit first, so the subtree of the other is opened as well: 3 more solves, 2 more
# leaves, carried for three steps until the shift drops them.  Gurobi met them in the other order (or saw them equal).
# Which of the two is met first is decided by the last digits of the bounds: oracle, shipped kernel, the kernels compiled for the
# problem (since round 5 the same code as the shipped ones) and the fleet driver all carry both extra leaves: 175
# (profiles/mc_r05/).  Pinned to that ONE value again (round 4: a range, because the compiled kernel of that round met 174 at
# the first of the three steps).
KNOWN_TIES = {('0003', 94): {11: (175, 175), 12: (175, 175), 13: (175, 175)}}


def _compare(st, tag, sims, steps, max_lost=1.0, warm_mean=True):
    """st: result of BatchedMPC.closed_loop / fleet.closed_loop_study on the published errors of `tag` for the
    simulations `sims` (indices into the fixture), at most `steps` steps."""
    ref = load_fixture('reference_closed_loop')
    sims = list(sims)
    pub_steps = ref['steps_' + tag][sims] if 'steps_' + tag in ref.files else np.full(len(sims), 50)
    assert st['cost_mismatches'] == []                                   # warm == cold cost at every step
    tot = dict(cs=[], pcs=[], ws=[], pws=[], reopened=[], calm=[])
    for j, i in enumerate(sims):
        n = min(int(pub_steps[j]), steps)                # steps with a solution (a warm start was built)
        lw, cs, ws = np.array(st['len_ws'][j]), np.array(st['nodes_cs'][j]), np.array(st['nodes_ws'][j])
        assert len(lw) == n, (tag, i, len(lw), n)                        # the simulation ends where the published one does
        want = ref['nodes_len_ws_' + tag][i, :n].copy()
        tie = KNOWN_TIES.get((tag, i), {})
        for t, (lo, hi) in tie.items():
            if t < n:
                assert lo <= lw[t] <= hi, (tag, i, t, lw[t])
                want[t] = lw[t]
        assert np.array_equal(lw, want), (tag, i)                        # cover sizes: exactly the published ones
        m = len(cs)                                                      # == n, or n + 1: the step that has no solution
        assert m == (n + 1 if n < min(steps, 50) else n), (tag, i, m, n)
        dcs = np.abs(cs - ref['nodes_cs_' + tag][i, :m])
        assert np.all(dcs[[t for t in range(m) if t not in tie]] <= 3), (tag, i)  # cold solves, also on the infeasible step
        assert np.all(dcs <= 6)
        tot['cs'].append(cs[:n]); tot['pcs'].append(ref['nodes_cs_' + tag][i, :n])
        if n > 1:
            tot['ws'].append(ws[1:n]); tot['pws'].append(ref['nodes_ws_' + tag][i, 1:n])
            tot['reopened'].append(np.array(st['reopened'][j])[:n - 1]); tot['calm'].append(lw[:n - 1] == 77)
    ws, pws, reopened, calm = (np.concatenate(tot[k]) for k in ('ws', 'pws', 'reopened', 'calm'))
    # warm solves = dive through the entering stage (9 at nub = 4; a little less when a bound prunes the dive early,
    # more when the search backtracks) + reopened leaves
    assert np.all(ws[calm] - reopened[calm] >= 5) and np.median(ws[calm] - reopened[calm]) == 9
    assert reopened[calm].mean() < max_lost
    if warm_mean:                                                        # (a statement about means: not for a handful of steps)
        assert ws.mean() <= pws.mean() and ws.mean() >= 8.0              # fewer lost proofs than Gurobi's rays, same dive
    assert np.all(pws[calm] >= 9)                                        # the published runs never beat the dive either
    cs, pcs = np.concatenate(tot['cs']), np.concatenate(tot['pcs'])
    return dict(cold_equal=float(np.mean(cs == pcs)), warm=float(ws.mean()), warm_published=float(pws.mean()),
                lost=float(reopened.mean()), cover_max=int(max(max(v) for v in st['len_ws'] if len(v))))


def _replay(ctrl, tag, sims, steps, max_lost=1.0, warm_mean=True):
    ref = load_fixture('reference_closed_loop')
    sims = list(range(sims)) if isinstance(sims, int) else list(sims)
    st = BatchedMPC(ctrl).closed_loop(X0, steps, seeds=tuple(sims), frontier_width=1, cold_too=True,
                                      errors=ref['errors_' + tag][sims])
    return _compare(st, tag, sims, steps, max_lost, warm_mean)


def test_replay_of_published_error_sequences_cpu():
    ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
    _replay(ctrl, '0001', sims=4, steps=10)
    _replay(ctrl, '0003', sims=3, steps=8, max_lost=4.0)                # larger disturbances break more proofs


def test_replay_of_the_published_large_disturbances_cpu():
    # sd = 0.01, parsed from the reference's text log: simulation 0 goes through covers of 110 and 149 nodes and a cold
    # search of 301 solves before its MIQP has no solution at step 8 (231 solves to prove it); 93 ends at step 6; 81 ends
    # at step 10 on an infeasible ROOT relaxation (1 solve).  Reproduced step by step, at the same steps.
    ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
    r = _replay(ctrl, '0010', sims=(0, 93, 81), steps=12, max_lost=6.0, warm_mean=False)
    assert r['cover_max'] == 149


def test_the_one_deviation_from_the_published_covers_is_a_tie():
    # simulation 94 of sd 0.003 up to step 11 (KNOWN_TIES): the cold search of that step meets a first incumbent and then
    # a second, fully fixed node whose cost is lower by less than 1e-9 relative -- an exact tie of the MIQP to the
    # accuracy of any solver; covers before it equal the published ones, the cover after it has the two extra leaves
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8)
    bm = BatchedMPC(ctrl)
    seen, marks, consume, many = [], [], bm._consume, bm.feedforward_many

    def spy_consume(tr, node, res, b, tol):
        seen.append((float(res['obj'][b]), int((tr.fix[node] >= 0).sum())))
        return consume(tr, node, res, b, tol)

    def spy_many(*a, **k):
        marks.append(len(seen))
        return many(*a, **k)
    bm._consume, bm.feedforward_many = spy_consume, spy_many
    st = bm.closed_loop(X0, 12, seeds=(94,), frontier_width=1, cold_too=True, errors=ref['errors_0003'][[94]])
    assert st['len_ws'][0][:11] == ref['nodes_len_ws_0003'][94, :11].tolist()
    assert st['len_ws'][0][11] == 175 and ref['nodes_len_ws_0003'][94, 11] == 173
    cold = seen[marks[22]:marks[23]]                      # calls alternate cold, warm: step 11's cold search is call 22
    full = sorted(o for o, depth in cold if depth == ctrl.T * ctrl.mld.nub and np.isfinite(o))
    assert len(full) >= 2 and 0 < (full[1] - full[0]) / full[0] < 1e-9


def test_without_the_lazy_terminal_set_every_proof_is_lost():
    # the other end of the bracket around the published 12.6: interior-point Farkas rays have maximal support and, when
    # they may use the terminal-set rows, none of them survives the shift
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='oracle', threads=8, lazy_terminal=False)
    st = BatchedMPC(ctrl).closed_loop(X0, 4, seeds=(0, 1), frontier_width=1, errors=ref['errors_0001'][:2])
    assert np.mean(st['reopened']) > 70 and np.mean([v[1:] for v in st['nodes_ws']]) > 75


@pytest.mark.gpu
def test_replay_of_the_whole_published_study_gpu():
    # Every simulation the reference published -- 100 x 50 steps at sd 0.001 and 0.003, the 109 started simulations of
    # sd 0.01 -- on the HIP path through the C++ fleet driver (hmpc_fleet_*): a cold- and a warm-started search per step
    # and simulation, frontier_width 1 (the reference's node order), one kernel launch per round shared by all trees.
    from warm_start_hmpc_amd.fleet import closed_loop_study
    ref = load_fixture('reference_closed_loop')
    ctrl = make_controller('cart_pole_with_walls', backend='hip')
    for tag, lost, cover in (('0001', 1.0, 77), ('0003', 4.0, 279), ('0010', 6.0, 423)):
        errors = ref['errors_' + tag]
        st = closed_loop_study(ctrl, X0, errors, frontier_width=1, cold_too=True)
        r = _compare(st, tag, range(errors.shape[0]), 50, max_lost=lost)
        # published covers reach 279 / 423 nodes on complete simulations (more on the ones that end early)
        assert r['cover_max'] >= cover
        assert st['steps'] == int(np.sum(ref['steps_' + tag])) if tag == '0010' else st['steps'] == 5000
        print('sd %s: %d steps, cold solves equal to published on %.0f %% of them, warm %.2f (published %.2f), proofs lost '
              'per shift %.2f, largest cover %d' % (tag, st['steps'], 100 * r['cold_equal'], r['warm'], r['warm_published'],
                                                    r['lost'], r['cover_max']))
