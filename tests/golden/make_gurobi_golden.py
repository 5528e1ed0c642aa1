"""Writes tests/golden/gurobi_golden.npz: records of the REFERENCE'S OWN SOLVER (Gurobi, through the model of
tests/gurobi_reference.py) for the node sets of qp_golden.npz.

Needs ``gurobipy`` with a licence -- neither exists in the build image nor on the GPU boxes, so this file has not been run
there ("parity unpinned" against Gurobi output, pinned through properties and independent solvers instead).
A maintainer with a licence runs, from the repository root,

    python tests/golden/make_gurobi_golden.py

and commits the npz; ``tests/test_gurobi_golden.py`` then holds the CPU oracle (``-m "not gpu"``) and the HIP kernel
(``-m gpu``) to Gurobi's records: statuses equal, objectives 1e-6, states and the inputs the cost determines within 1e-5
relative (BASELINE.json), Farkas proofs through their sign and the certificate identities (rays are not unique).
Per set NAME of qp_golden.npz (n20, n20dive, n20tree, n20x0, n10, n10free, n40, onewall) it stores
    NAME_T, NAME_x0, NAME_fix, NAME_terminal, NAME_fixture   the inputs (copied from qp_golden.npz)
    NAME_status, NAME_obj, NAME_dual_obj, NAME_primal, NAME_dual   Gurobi's records in the layout of include/hmpc.h
and the Gurobi version / parameters in ``meta``."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

from helpers import make_controller  # noqa: E402
import gurobi_reference  # noqa: E402

# Gurobi's defaults are 1e-6 feasibility / optimality; the fixture is made tighter than the product, like qp_golden.npz
PARAMS = {'FeasibilityTol': 1e-9, 'OptimalityTol': 1e-9, 'BarConvTol': 1e-12, 'Threads': 1}


def fixture_of(name):
    return 'cart_pole_one_wall' if name == 'onewall' else 'cart_pole_with_walls'


def main():
    if not gurobi_reference.available():
        raise SystemExit('gurobipy is not importable here: this script needs Gurobi and a licence')
    import gurobipy
    g = np.load(os.path.join(HERE, 'qp_golden.npz'))
    names = sorted({k[:-4] for k in g.files if k.endswith('_fix')})
    out = {'meta': np.array('gurobi %s, params %s' % ('.'.join(str(v) for v in gurobipy.gurobi.version()), PARAMS))}
    for name in names:
        T, terminal = int(g[name + '_T']), bool(g[name + '_terminal'])
        ctrl = make_controller(fixture_of(name), T=T, terminal=terminal, backend='oracle')
        qp = gurobi_reference.GurobiBatchedQP(ctrl.problem_data(), gurobi_params=PARAMS)
        res = qp.solve_batch(g[name + '_x0'], g[name + '_fix'])
        assert np.array_equal(res['status'], g[name + '_status']), (name, 'statuses differ from the oracle-made fixture')
        for k in ('T', 'x0', 'fix', 'terminal'):
            out['%s_%s' % (name, k)] = g['%s_%s' % (name, k)]
        out[name + '_fixture'] = np.array(fixture_of(name))
        for k in ('status', 'obj', 'dual_obj', 'primal', 'dual'):
            out['%s_%s' % (name, k)] = res[k]
        print('%-8s %4d nodes (%d optimal), Gurobi %.3f s' % (name, len(res['status']), int((res['status'] == 0).sum()), res['solver_time']))
    np.savez_compressed(os.path.join(HERE, 'gurobi_golden.npz'), **out)
    print('wrote gurobi_golden.npz')


if __name__ == '__main__':
    main()
