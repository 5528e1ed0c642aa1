"""Diagnostic: variants of the kernels of a problem whose compiled kernels came out wrong (see gpu_dev_selfcheck_case.py; like it,
it reproduced only before the static row map was cut back to nx + nu <= 15: profiles/r04_nz16_register_kernels.txt; round 5:
the cause was the LDS carve of the register kernels at nz >= 16, fixed, nx + nu = 16 is admitted again).  DBG_ONLY="label;label"."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
from oracle.oracle_qp import OracleBatchedQP
os.environ['HMPC_JIT_SELFCHECK'] = '0'
SPECS = ((9, 3, 4, 6, 23), (9, 3, 4, 10, 23), (8, 4, 4, 6, 23), (10, 2, 4, 6, 23), (9, 3, 3, 6, 23))
if os.environ.get('DBG_SHAPES'):   # "nx,nuc,nub,T,seed;..."
    SPECS = tuple(tuple(int(v) for v in q.split(',')) for q in os.environ['DBG_SHAPES'].split(';'))
for spec in SPECS:
    nx, nuc, nub, T, seed = spec
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
    ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    orc = OracleBatchedQP(ctrl.problem_data(), threads=16)
    fix = np.full((6, T * nub), -1, np.int8)
    fix[1, :nub] = 0
    fix[2, :2 * nub] = 0
    fix[3, 0] = 1
    fix[4, :3] = 0
    fix[5, :T * nub // 2] = 0
    b = orc.solve_batch(x0, fix)
    for label, env in (('sized', {}), ('sized, readlane broadcasts', {'HMPC_JIT_FLAGS': '-DHMPC_DPP_FEW'}), ('sized, default schedule', {'HMPC_JIT_SCHED': 'default'}),
                       ('sized, -O1', {'HMPC_JIT_FLAGS': '-O1'}), ('per shape', {'HMPC_JIT_SIZED': '0'}),
                       ('sized, no VGPR -> AGPR spilling', {'HMPC_JIT_FLAGS': '-mllvm -amdgpu-spill-vgpr-to-agpr=0'}),
                       ('sized, no stack slot sharing', {'HMPC_JIT_FLAGS': '-mllvm -no-stack-slot-sharing'}),
                       ('sized, stack slot colouring off', {'HMPC_JIT_FLAGS': '-mllvm -disable-ssc'}))[(int(os.environ.get('DBG_FROM', 0))):]:
        if os.environ.get('DBG_ONLY') and label not in os.environ['DBG_ONLY'].split(';'):
            continue
        os.environ.update(env)
        hip = HipBatchedQP(ctrl.problem_data())
        for k in env:
            del os.environ[k]
        out = []
        for waves in ('1', '2', '4'):
            os.environ['HMPC_WAVES'] = waves
            a = hip.solve_batch(x0, fix)
            del os.environ['HMPC_WAVES']
            out.append('w%s %s' % (waves, 'ok' if np.array_equal(a['status'], b['status']) else 'WRONG %s' % a['status']))
        print(spec, 'nz', nx + nuc + nub, label, hip.kernel_info(), '; '.join(out), 'oracle', b['status'], flush=True)
