"""Timing of the warm-start shift kernel alone (diagnostic; HMPC_LIB selects a library variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import conftest  # noqa
import numpy as np
import torch
import warm_start_hmpc_amd.qp_backend as qb
if os.environ.get('HMPC_LIB'):
    qb.LIBRARY_PATH = qb.LIBRARY_PATH.replace('libhmpc.so', os.environ['HMPC_LIB'])
from helpers import make_controller
import bench
T = int(os.environ.get('SHIFT_T', '20'))          # (n_dual = 45 T + 110: T = 10 and 26 give rows that are whole 128-byte lines)
ctrl = make_controller('cart_pole_with_walls', T=T, backend='hip')
dev = torch.device('cuda', 0)
for leaves in (4096, 65536, 262144):
    r = bench.shift_bandwidth(ctrl, dev, leaves=leaves, reps=20)
    print(os.environ.get('HMPC_LIB', 'libhmpc.so'), 'T', T, 'row bytes', 8 * ctrl.qp.n_dual, leaves, '%.3f ms' % r['kernel_ms_avg'], '%.0f GB/s' % r['achieved_GBs'], flush=True)
