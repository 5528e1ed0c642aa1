"""Diagnostic (GPU box): does hmpc_create pick up the register kernel compiled for an arbitrary shape?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
os.environ['HMPC_JIT_VERBOSE'] = '1'
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP, jit_shapes
for (nx, nuc, nub, T, seed) in ((6, 2, 3, 8, 3), (8, 3, 4, 10, 2)):
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
    ctrl = HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend())
    print(jit_shapes(ctrl.problem_data()))
    hip = HipBatchedQP(ctrl.problem_data())
    print('kernel kinds', hip.kernel_info(), flush=True)
