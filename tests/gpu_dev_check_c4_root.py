"""Diagnostic: the root relaxation of configs[4] alone on the check build's sized streaming kernel, per-iteration scalars (HMPC_TRACE=1)."""
import os
import sys
os.environ['HMPC_LIBRARY_NAME'] = 'libhmpc_check.so'
os.environ['HMPC_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import numpy as np
from helpers import random_mld, _NoBackend
from warm_start_hmpc_amd.controller import HybridModelPredictiveController
from warm_start_hmpc_amd.qp_backend import HipBatchedQP
mld, objective, x0 = random_mld()
ctrl = HybridModelPredictiveController(mld, 30, objective, None, backend=_NoBackend())
hip4 = HipBatchedQP(ctrl.problem_data())
a = hip4.solve_batch(x0, np.full((1, 240), -1, np.int8))
print('status', a['status'], 'iters', a['iters'], 'polished', a['polished'])
