"""Diagnostic (GPU box): compiled kernels of many problems under two compiler recipes -- the ILP schedule (round 4's default) and the
compiler's default schedule --, each problem and recipe in a process of its own under a watchdog (a wrong binary may not come back):
statuses and the rays of infeasible nodes against the oracle, 1 / 2 / 4 waves per node, nets off.

    python tests/gpu_dev_variant_sweep.py            (driver)          DBG_ONE=<spec> DBG_RECIPE=<ilp|default> (one case, internal)
"""
import os
import subprocess
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CART = [('cart_pole_with_walls', T, term) for T in (10, 20, 40) for term in (True, False)] + [('cart_pole_one_wall', T, True) for T in (10, 20, 40)]
RANDOM = [(6, 2, 3, 12, 3), (6, 2, 3, 8, 3), (8, 3, 4, 10, 2), (8, 5, 2, 12, 55), (3, 3, 6, 12, 38), (4, 4, 7, 10, 61), (9, 3, 4, 6, 23), (8, 4, 4, 6, 23),
          (5, 2, 2, 9, 21), (7, 3, 3, 14, 22), (4, 1, 5, 16, 24), (11, 2, 2, 5, 35), (5, 5, 5, 3, 36), (8, 2, 2, 30, 40), (6, 2, 3, 40, 41), (4, 2, 1, 25, 42),
          (1, 1, 1, 4, 31), (3, 1, 1, 2, 33), (10, 2, 4, 6, 23), (9, 3, 3, 6, 23)]

if os.environ.get('DBG_ONE'):
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('DBG_WATCHDOG', 60)), exit=True)
    import conftest  # noqa
    import numpy as np
    from helpers import make_controller, random_mld, random_prefix_frontier, _NoBackend
    os.environ['HMPC_JIT_SELFCHECK'] = '0'
    spec = eval(os.environ['DBG_ONE'])
    if isinstance(spec[0], str):
        hip = make_controller(spec[0], T=spec[1], terminal=spec[2], backend='hip').qp
        orc = make_controller(spec[0], T=spec[1], terminal=spec[2], backend='oracle', threads=16).qp
        T, nub, x0 = spec[1], (4 if 'walls' in spec[0] else 2), np.array([0., 0., .5 if spec[1] == 10 and spec[2] else 1., 0.])
    else:
        from warm_start_hmpc_amd.controller import HybridModelPredictiveController
        from warm_start_hmpc_amd.qp_backend import HipBatchedQP
        from oracle.oracle_qp import OracleBatchedQP
        nx, nuc, nub, T, seed = spec
        mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
        data = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend()).problem_data()
        hip, orc = HipBatchedQP(data), OracleBatchedQP(data, threads=16)
    fix = random_prefix_frontier(T, nub, 192, p_one=0.2, seed0=4000)
    fix[0, :] = -1
    b = orc.solve_batch(x0, fix)
    inf = b['status'] == 1
    out = []
    for w in ('1', '2', '4'):
        os.environ['HMPC_WAVES'] = w
        a = hip.solve_batch(x0, fix)
        same = np.array_equal(a['status'], b['status'])
        rays = same and (not inf.any() or (np.nanmax(np.abs(a['dual'][inf] - b['dual'][inf])) < 1e-5 and not np.isnan(a['dual'][inf]).any()))
        fin = (a['status'] == 0) & (b['status'] == 0)
        objs = same and (not fin.any() or np.max(np.abs(a['obj'][fin] - b['obj'][fin]) / (1 + np.abs(b['obj'][fin]))) < 1e-6)
        out.append('w%s %s' % (w, 'ok' if same and rays and objs else 'WRONG(status %s rays %s obj %s)' % (same, bool(rays), bool(objs))))
    print('RESULT', hip.kernel_info(), '; '.join(out), flush=True)
    sys.exit(0)

bad = {'ilp': 0, 'default': 0}
for spec in CART + RANDOM:
    for recipe in ('ilp', 'default'):
        env = dict(os.environ, DBG_ONE=repr(spec), DBG_RECIPE=recipe)
        if recipe == 'default':
            env['HMPC_JIT_SCHED'] = 'default'
        tic = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=200)
            res = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
            line = res[0] if res else ('NO RESULT (exit %d): %s' % (p.returncode, (p.stderr.strip().splitlines() or ['?'])[0][:120]))
        except subprocess.TimeoutExpired:
            line = 'TIMEOUT'
        good = line.startswith('RESULT') and 'WRONG' not in line
        bad[recipe] += not good
        print('%-5s %-8s %-44s %s (%.0f s)' % ('ok' if good else 'FAIL', recipe, spec, line, time.time() - tic), flush=True)
print('SWEEP: failures', bad)
