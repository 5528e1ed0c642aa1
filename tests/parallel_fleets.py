"""``closed_loop_parallel``: K closed loops as several fleets on library handles and host threads of their own.

DIAGNOSTIC, not product (moved out of warm_start_hmpc_amd/fleet.py in round 5).  Measured three times (rounds 3 and 5): no gain
-- one fleet's step is bound by the device --, and with handles created and destroyed while other host threads are inside the
library the HIP runtime of this image (the libamdhip64 bundled with torch 2.10) has crashed inside its own allocation map: one
call in ~30 - 240 with 8 fleets (profiles/r05_fleet_trace.txt: native backtraces through hipFree in hmpc_destroy and through
libhsa-runtime64).  The library is driven by one host thread per process.
"""
import copy
from concurrent.futures import ThreadPoolExecutor
from time import perf_counter

import numpy as np

from warm_start_hmpc_amd.fleet import FleetMPC


def closed_loop_parallel(controller, x0, n_steps, errors, parts=2, keep=None, **kwargs):
    """K closed loops as ``parts`` fleets of K / parts loops, each on a library handle and a host thread of its own.

    One fleet alternates between host bookkeeping (selection, prune / branch, staging) and a kernel launch, so at many
    loops the GPU idles while the host works and vice versa.  Independent loops need no lockstep between fleets: with
    several fleets in flight the kernel of one overlaps the bookkeeping of the others (the C calls release the
    interpreter lock; every handle has its own stream and workspaces, include/hmpc.h: one launch in flight per handle).
    Returns the dictionary of ``FleetMPC.closed_loop`` with the per-loop arrays in the order of ``errors``; ``wall`` is
    the time until the last fleet has finished."""
    from warm_start_hmpc_amd.qp_backend import HipBatchedQP
    errors = np.asarray(errors, dtype=np.float64)
    K = errors.shape[0]
    parts = max(1, min(int(parts), K))
    bounds = [(K * j) // parts for j in range(parts + 1)]
    fleets = [] if keep is None else keep.setdefault(('fleets', parts, K), [])      # (keep: a dict -- handles and fleets are made once and reused)
    for j in range(parts if not fleets else 0):
        c = controller if j == 0 else copy.copy(controller)
        if j:
            params = dict(controller.solver_params)
            params.setdefault('device', controller.qp.device)
            c.qp = HipBatchedQP(controller.problem_data(), **params)
        fleets.append(FleetMPC(c, bounds[j + 1] - bounds[j], handdown=kwargs.pop('handdown', True) if j == 0 else fleets[0].handdown))
    tic = perf_counter()
    with ThreadPoolExecutor(max_workers=parts) as pool:
        runs = list(pool.map(lambda j: fleets[j].closed_loop(x0, n_steps, errors[bounds[j]:bounds[j + 1]], **kwargs), range(parts)))
    wall = perf_counter() - tic
    st = {k: np.concatenate([r[k] for r in runs]) for k in ('costs', 'nodes_ws', 'len_ws', 'reopened')}
    steps = sum(r['steps'] for r in runs)
    stats = [f.stats() for f in fleets]
    st.update(wall=wall, steps=steps, steps_per_sec=steps / wall if wall > 0 else 0., parts=parts,
              rounds=sum(s['rounds'] for s in stats), launched=sum(s['launched'] for s in stats), handed=sum(s['handed'] for s in stats))
    return st
