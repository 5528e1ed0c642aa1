"""Regenerates tests/golden/reference_closed_loop.npz from the reference's PUBLISHED experiment output.

Data only (no reference code): the model errors the reference applied in its closed-loop Monte-Carlo study and the
node counts it recorded per MPC step, written by notebooks/cart_pole_with_walls/statistical_analysis.py:93-207
(cs = QP solves of the cold-started branch and bound at that step, ws = of the warm-started one, len_ws = size of the
warm start built after the step).

  sd 0.001, 0.003 -- ALL 100 published simulations, from the plain arrays
      /root/reference/notebooks/cart_pole_with_walls/data/errors_sd_{0.001,0.003}.npy        (100 x 50 x 4 float64)
      /root/reference/notebooks/cart_pole_with_walls/data/nodes_{cs,ws,len_ws}_sd_{...}.npy  (100 x 50 int64)
  sd 0.010 -- the arrays of this level are pickled object arrays (ragged: simulations that left the feasible set are
      stored shorter) and are NOT loaded; the same numbers are parsed from the text log
      /root/reference/notebooks/cart_pole_with_walls/data/solve_log_sd_0.010.log  (from line 8 on; format written at
      statistical_analysis.py:96,111,128,145,161,190,195).  The log holds 109 started simulations: 100 complete ones
      and 9 that ended on a step where the cold-started search returned no solution (`grb: 0` nodes on 7 of them).  The log prints
      the model error with 8 digits; the full-precision value is `e_sd * randn(4) * x_max` from the stream of
      `np.random.seed(simulation index)` (statistical_analysis.py:73,176 -- nothing else draws from numpy's stream),
      regenerated here and checked against every printed vector.  (Regenerating sd 0.001 / 0.003 the same way
      reproduces the stored arrays but for 15 of 20 000 entries that are one unit in the last place off -- the
      reference machine's libm in the Gaussian draw; the regenerated sd 0.01 errors are that close to the applied ones.)

Keys: `errors_<tag>`, `nodes_cs_<tag>`, `nodes_ws_<tag>`, `nodes_len_ws_<tag>` for tag in 0001, 0003, 0010;
for 0010 additionally `steps_0010` (number of steps with a warm start built, 50 for a complete simulation; a
simulation with steps < 50 was infeasible at step `steps`, whose cs / ws counts are still recorded) and rows padded
with 0 beyond.  Runs in the build container only (the reference does not travel); tests replay these disturbances
through this repository's controller (tests/test_reference_replay.py).

    python tests/golden/make_reference_data.py
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = '/root/reference/notebooks/cart_pole_with_walls/data'
N_STEPS = 50

STEP = re.compile(r'Time step (\d+) \(cs: (\d+), [\d.]+\) \(ws: (\d+), [\d.]+\) \(grb: (\d+), [\d.]+\) '
                  r'\(grb_fair: (\d+), [\d.]+\) (?:\(ws info: (\d+)(?:, [\d.]+)+\) \(e: [\d.]+, \[([^\]]*)\]\))?')


def parse_log(path, e_sd, x_max):
    text = open(path).read().replace('\n ', ' ')            # numpy wraps long vectors
    parts = re.split(r'\n\nSimulation (\d+)\nSimulation success (\d+)\n', text)
    sims = []
    for k in range(1, len(parts), 3):
        index, body = int(parts[k]), parts[k + 2]
        assert 'Unseccessful' not in body                   # no solver ever broke, only infeasible MIQPs
        rng = np.random.RandomState(index)                  # == np.random.seed(index); np.random.randn
        cs, ws, lw, err, grb = [], [], [], [], []
        for m in STEP.finditer(body):
            assert int(m.group(1)) == len(cs)
            assert len(cs) == len(lw)                       # nothing follows an infeasible step
            cs.append(int(m.group(2))); ws.append(int(m.group(3))); grb.append(int(m.group(4)))
            if m.group(6) is None:
                continue                                    # infeasible step: no warm start, no error
            lw.append(int(m.group(6)))
            e = e_sd * np.multiply(rng.randn(4), x_max)         # the operation order of statistical_analysis.py:176
            printed = np.array([float(v) for v in m.group(7).split()])
            assert np.allclose(printed, e, rtol=0, atol=6e-9), (index, len(cs), printed, e)
            err.append(e)
        if len(lw) < N_STEPS:
            # ended on a step whose cold-started search found no solution (`solution_cs is None`, :165-166); Gurobi's
            # own MIQP agrees on 7 of the 9 (`grb: 0` nodes) and reports a solution on two borderline ones
            assert len(cs) == len(lw) + 1
        sims.append((index, cs, ws, lw, err))
    return sims


if __name__ == '__main__':
    out = {}
    for sd in ('0.001', '0.003'):
        tag = sd.replace('.', '')
        for key in ('errors', 'nodes_cs', 'nodes_ws', 'nodes_len_ws'):
            a = np.load(os.path.join(DATA, '%s_sd_%s.npy' % (key, sd)), allow_pickle=False)
            assert a.shape[:2] == (100, N_STEPS)
            out['%s_%s' % (key, tag)] = a
        cs, ws, lw = (out['nodes_%s_%s' % (k, tag)] for k in ('cs', 'ws', 'len_ws'))
        print('sd', sd, 'published over 100 simulations: cold %.1f (%d..%d), warm (steps >= 1) %.2f (%d..%d), cover %d..%d'
              % (cs.mean(), cs.min(), cs.max(), ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max(), lw.min(), lw.max()))
        out['summary_%s' % tag] = np.array([cs.mean(), cs.min(), cs.max(), ws[:, 1:].mean(), ws[:, 1:].min(), ws[:, 1:].max(), lw.min(), lw.max()])

    fx = np.load(os.path.join(HERE, 'cart_pole_with_walls.npz'))
    sims = parse_log(os.path.join(DATA, 'solve_log_sd_0.010.log'), 0.01, fx['x_max'])
    assert [s[0] for s in sims] == list(range(len(sims)))
    n = len(sims)
    steps = np.array([len(s[3]) for s in sims])
    cs, ws, lw = (np.zeros((n, N_STEPS), dtype=np.int64) for _ in range(3))
    err = np.zeros((n, N_STEPS, 4))
    for i, (_, c, w, l, e) in enumerate(sims):
        cs[i, :len(c)], ws[i, :len(w)], lw[i, :len(l)] = c, w, l
        if e:
            err[i, :len(e)] = e
    out.update(errors_0010=err, nodes_cs_0010=cs, nodes_ws_0010=ws, nodes_len_ws_0010=lw, steps_0010=steps)
    full = steps == N_STEPS
    print('sd 0.010 from the log: %d simulations started, %d complete, infeasible at step %s of simulations %s'
          % (n, full.sum(), steps[~full].tolist(), np.flatnonzero(~full).tolist()))
    print('   complete ones: cold %.1f (%d..%d), warm (steps >= 1) %.2f (%d..%d), cover %d..%d'
          % (cs[full].mean(), cs[full].min(), cs[full].max(), ws[full][:, 1:].mean(), ws[full][:, 1:].min(),
             ws[full][:, 1:].max(), lw[full].min(), lw[full].max()))
    np.savez_compressed(os.path.join(HERE, 'reference_closed_loop.npz'), **out)
