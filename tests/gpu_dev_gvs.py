"""Diagnostic (GPU box): the workload of bench.py's `generic_vs_specialised` key -- random MLD nx = 6, nu = 2 + 3, N = 12, 2048
random prefixes -- on the register kernel compiled with the problem's sizes, on the run-time-sized kernel and on the oracle:
which nodes end undecided / unpolished on which kernel, and the iteration trace of each of them."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import random_mld, random_prefix_frontier, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP

os.environ['HMPC_JIT_SELFCHECK'] = '0'   # (no first-use check, no second opinion: what the compiled kernels themselves return)
mld, obj, x0 = random_mld(nx=6, nuc=2, nub=3, seed=3)
T = 12
c = HybridModelPredictiveController(mld, T, obj, None, backend=_NoBackend())
f = random_prefix_frontier(T, 3, 2048, p_one=0.3)
f[0, :] = -1
data = c.problem_data()
orc = OracleBatchedQP(data, threads=16).solve_batch(x0, f)


def describe(tag, r):
    st = r['status']
    pol = (r['polished'] != 0)
    print(tag, 'optimal', int((st == 0).sum()), 'infeasible', int((st == 1).sum()), 'undecided', np.flatnonzero(st > 1).tolist(),
          'unpolished optimal', np.flatnonzero((st == 0) & ~pol).tolist(), flush=True)


describe('oracle', orc)
spec = HipBatchedQP(data)
print('kinds', spec.kernel_info())
res = {}
for w in ('1', '2', '4'):
    os.environ['HMPC_WAVES'] = w
    res[w] = spec.solve_batch(x0, f)
    del os.environ['HMPC_WAVES']
    describe('sized register kernel, %s waves' % w, res[w])
os.environ['HMPC_JIT_SIZED'] = '0'
shp = HipBatchedQP(data)
del os.environ['HMPC_JIT_SIZED']
print('kinds per shape', shp.kernel_info())
os.environ['HMPC_WAVES'] = '1'
rs = shp.solve_batch(x0, f)
describe('per-shape register kernel, 1 wave', rs)
os.environ['HMPC_JIT'] = '0'
gen = HipBatchedQP(data)
del os.environ['HMPC_JIT']
rg = gen.solve_batch(x0, f)
describe('run-time-sized kernel, 1 wave', rg)
del os.environ['HMPC_WAVES']
hard = sorted(set(np.flatnonzero(res['1']['status'] > 1).tolist() + np.flatnonzero((res['1']['status'] == 0) & (res['1']['polished'] == 0)).tolist()))
print('iterations of nodes 936 1386 1653 1979: sized w1', (res['1']['iters'][[936, 1386, 1653, 1979]] & 0xffff).tolist(), 'per shape', (rs['iters'][[936, 1386, 1653, 1979]] & 0xffff).tolist(),
      'generic', (rg['iters'][[936, 1386, 1653, 1979]] & 0xffff).tolist(), 'oracle', (orc['iters'][[936, 1386, 1653, 1979]] & 0xffff).tolist())
if os.environ.get('DBG_NO_TRACE'):
    sys.exit(0)
print('hard nodes', hard, 'iters sized', (res['1']['iters'][hard] & 0xffff).tolist(), 'generic', (rg['iters'][hard] & 0xffff).tolist(), 'oracle', (orc['iters'][hard] & 0xffff).tolist())
np.savez('gpurun_out/gvs_hard.npz', hard=np.array(hard), fix=f[hard], st_sized=res['1']['status'][hard], st_gen=rg['status'][hard])
# traces: each hard node alone (node 0 of its launch is the traced one), one wave per node
os.environ['HMPC_TRACE'] = '1'
os.environ['HMPC_WAVES'] = '1'
os.environ['HMPC_JIT_SELFCHECK'] = '0'
for b in hard[:3]:
    for tag, env in (('sized', {}), ('generic', {'HMPC_JIT': '0'})):
        os.environ.update(env)
        q = HipBatchedQP(data)
        for k in env:
            del os.environ[k]
        print('--- trace node', b, tag, flush=True)
        sys.stderr.flush()
        r = q.solve_batch(x0, f[b:b + 1])
        sys.stderr.flush()
        print('status', r['status'], 'iters', r['iters'] & 0xffff, 'polished', r['polished'], flush=True)
