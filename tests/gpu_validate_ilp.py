"""Validation of the kernels compiled with the compiler's ILP schedule (GPU box; csrc/hmpc_jit.h: sched_flags).

The ILP schedule is worth 8 - 15 % on these one-wave-per-SIMD kernels and produced most of the wrong binaries of rounds 4 and 5 --
one of which did not come back from a launch.  hmpc_create therefore uses it only for binaries listed in the cache's VALIDATED
manifest.  This script writes that manifest: for every problem of tests/jit_problems.py (the controllers of the suite and the
bench, the random MLDs) and every wave count, the ILP-scheduled binary -- cold and hand-down instantiation -- is run against the
oracle in a PROCESS OF ITS OWN under a watchdog: statuses, objectives (1e-8), rays of infeasible nodes (1e-6), state
trajectories of polished nodes (1e-6), a batch larger than the resident grid (the dynamic hand-out of nodes) and a small one.
Only binaries that pass everything are listed, by the name of their cache entry (which covers the problem's sizes, the kernel
sources, the flags, the architecture and the compiler's identity).

    HMPC_JIT_SCHED=iterative-ilp python tests/jit_problems.py        # (build container: compiles them into the in-tree cache)
    python tests/gpu_validate_ilp.py                                 # (GPU box) -> gpurun_out/VALIDATED ; copy to warm-start-hybrid-mpc_amd/jit_cache/
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

if os.environ.get('VAL_ONE'):
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('VAL_WATCHDOG', 150)), exit=True)
    import conftest  # noqa
    import numpy as np
    from helpers import make_controller, random_prefix_frontier, _NoBackend
    from jit_problems import problem
    from warm_start_hmpc_amd.qp_backend import HipBatchedQP, jit_prebuild
    from oracle.oracle_qp import OracleBatchedQP
    os.environ['HMPC_JIT_SELFCHECK'] = '0'
    os.environ['HMPC_JIT_SCHED'] = os.environ.get('VAL_SCHED', 'iterative-ilp')   # (VAL_SCHED=default: the same run over the default recipe, report only)
    spec = eval(os.environ['VAL_ONE'])
    if isinstance(spec[0], str):
        ctrl = make_controller(spec[0], T=spec[1], terminal=spec[2], backend=_NoBackend())
        data, T, nub, nx = ctrl.problem_data(), spec[1], ctrl.mld.nub, ctrl.mld.nx
        x0 = np.array([0., 0., .5 if T == 10 and spec[2] else 1., 0.])
    else:
        data, mld, objective, x0 = problem(*spec)
        T, nub, nx = spec[4], spec[2], spec[0]
    paths = jit_prebuild(data)                                            # (the ILP binaries: HMPC_JIT_SCHED above)
    hip, orc = HipBatchedQP(data), OracleBatchedQP(data, threads=16)
    kinds = hip.kernel_info()
    big = kinds[2] in (1, 5)                                              # (the streaming form: four waves per node whatever the batch)
    xs = (T + 1) * nx
    ok, why = {}, []
    for waves in (('4',) if big else ('1', '2', '4')):
        good = True
        for count in ((160, 1400) if not big else (96, 320)):
            fix = random_prefix_frontier(T, nub, count, p_one=0.15, seed0=6000 + count)
            fix[0, :] = -1
            b = orc.solve_batch(x0, fix)
            os.environ['HMPC_WAVES'] = waves
            a = hip.solve_batch(x0, fix)
            same = np.array_equal(a['status'], b['status']) and np.all(a['status'] <= 1)
            fin, inf = (b['status'] == 0), (b['status'] == 1)
            pol = fin & (a['polished'] > 0) & (b['polished'] > 0)
            checks = {'statuses': bool(same)}
            if same:
                checks['objectives'] = not fin.any() or np.max(np.abs(a['obj'][fin] - b['obj'][fin]) / (1 + np.abs(b['obj'][fin]))) < 1e-8
                checks['rays'] = not inf.any() or (not np.isnan(a['dual'][inf]).any() and np.max(np.abs(a['dual'][inf] - b['dual'][inf])) < 1e-6)
                checks['trajectories'] = not pol.any() or np.max(np.abs(a['primal'][pol][:, :xs] - b['primal'][pol][:, :xs])) < 1e-6 * max(1., np.max(np.abs(b['primal'][pol][:, :xs])))
                checks['polished as the oracle'] = int(pol.sum()) >= int((fin & (b['polished'] > 0)).sum()) - 1
                # the hand-down instantiation (a binary of its own): every optimal, polished node handed its own record
                idx = np.where(pol, np.arange(count), -1).astype(np.int32)
                w = hip.solve_batch(x0, fix, warm=(a['primal'], a['dual'], idx))
                checks['hand-down statuses'] = np.array_equal(w['status'], a['status'])
                checks['hand-down objectives'] = not fin.any() or np.max(np.abs(w['obj'][fin] - a['obj'][fin]) / (1 + np.abs(a['obj'][fin]))) < 1e-6
                checks['hand-down verifies'] = not pol.any() or (w['iters'][pol] & 0xffff == 0).mean() > 0.9
            del os.environ['HMPC_WAVES']
            good = good and all(checks.values())
            why += ['w%s, %d nodes: %s' % (waves, count, k) for k, g in checks.items() if not g]
        ok[waves] = bool(good)
    # a cache entry may serve two wave counts (a problem whose one-wave kernel does not exist): it is valid if every count it serves is
    names = {}
    slots = ('4',) if big else ('1', '2', '4')
    for p_ in paths:
        base = os.path.basename(p_)[:-3]
        served = [w for w in slots if ('_w%s_' % w) in base] or list(slots)
        names[base] = all(ok.get(w, False) for w in served)
    print('RESULT', repr((kinds, ok, names, why)), flush=True)
    sys.exit(0)

from jit_problems import CONTROLLERS, REGISTER_SHAPES, SIZED
valid, report = [], []
for spec in CONTROLLERS + REGISTER_SHAPES + SIZED:
    tic = time.time()
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, VAL_ONE=repr(spec)), capture_output=True, text=True, timeout=400)
        res = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
        if res:
            kinds, ok, names, why = eval(res[0][7:])
            valid += [n for n, g in names.items() if g]
            line = 'kinds %s waves %s%s' % (kinds, ok, (' FAILED: ' + '; '.join(why)) if why else '')
        else:
            line = 'NO RESULT (exit %d): %s' % (p.returncode, (p.stderr.strip().splitlines() or ['?'])[-1][:160])
    except subprocess.TimeoutExpired:
        line = 'TIMEOUT'
    report.append('%-44s %s (%.0f s)' % (spec, line, time.time() - tic))
    print(report[-1], flush=True)
out = os.path.join(os.path.dirname(HERE), 'gpurun_out')
os.makedirs(out, exist_ok=True)
if os.environ.get('VAL_SCHED', 'iterative-ilp') != 'iterative-ilp':
    with open(os.path.join(out, 'VALIDATED.%s.report.txt' % os.environ['VAL_SCHED']), 'w') as f:
        f.write('\n'.join(report) + '\n')
    print('schedule %s: %d binaries pass' % (os.environ['VAL_SCHED'], len(set(valid))))
    sys.exit(0)
with open(os.path.join(out, 'VALIDATED'), 'w') as f:
    f.write('# binaries compiled with the ILP schedule that tests/gpu_validate_ilp.py ran against the oracle (1 / 2 / 4 waves, cold and hand-down\n'
            '# instantiation, small and large batches, a process and a watchdog each); csrc/hmpc_jit.h uses that schedule for these only\n')
    for n in sorted(set(valid)):
        f.write(n + '\n')
with open(os.path.join(out, 'VALIDATED.report.txt'), 'w') as f:
    f.write('\n'.join(report) + '\n')
print('VALIDATED: %d binaries' % len(set(valid)))
