"""The problems of the test suite and of bench.py: ``hmpc_create`` compiles the kernels of a problem with its sizes as
constants at first use (csrc/hmpc_jit.h) -- register kernels where the static row map holds the problem, the run-time-sized
kernel or its streaming form elsewhere.  ``prewarm()`` compiles them into the cache ahead of time, without a GPU
(``jit_prebuild``) -- ``__graft_entry__.build()`` calls it, so that a GPU box (which has the compiler too) meets cache hits:

    python tests/jit_problems.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (puts the package on the path)
from helpers import random_mld, _NoBackend

# (nx, nuc, nub, seed, T): random MLDs of helpers.random_mld
REGISTER_SHAPES = ((6, 2, 3, 3, 8), (6, 2, 3, 3, 12), (8, 3, 4, 2, 10), (8, 5, 2, 55, 12), (3, 3, 6, 38, 12),   # (the last two: the binaries round 4 saw come out wrong)
                   (9, 3, 4, 23, 6), (8, 4, 4, 23, 6))   # nx + nu = 16: the full row of 16 lanes (round 4: every node NUMERICAL -- the LDS carve, fixed in round 5)
SIZED = ((20, 6, 8, 0, 30),       # BASELINE configs[4]: beyond one CU's LDS, the streaming form
         (10, 4, 4, 5, 8),        # nx + nu = 18: beyond the static row map, fits LDS (1 / 2 / 4 waves per node)
         (12, 3, 4, 23, 6))       # nx + nu = 19, another one


# (fixture, T, terminal set): the controllers of tests/ and bench.py (helpers.make_controller)
CONTROLLERS = tuple((f, T, term) for f in ('cart_pole_with_walls', 'cart_pole_one_wall') for T in (10, 20, 40) for term in (True, False))


def problem(nx, nuc, nub, seed, T):
    from warm_start_hmpc_amd.controller import HybridModelPredictiveController
    mld, objective, x0 = random_mld(nx=nx, nuc=nuc, nub=nub, seed=seed)
    return HybridModelPredictiveController(mld, T, objective, None, backend=_NoBackend()).problem_data(), mld, objective, x0


def prewarm(verbose=False, prune=False):
    """prune: entries of the IN-TREE cache that none of these problems uses are deleted afterwards -- what an edit of the kernel
    sources leaves behind (the key holds a hash of the sources); the directory travels to the GPU boxes with the tree."""
    from warm_start_hmpc_amd.qp_backend import jit_prebuild, LIBRARY_PATH
    from helpers import make_controller
    paths = []
    for spec in CONTROLLERS + REGISTER_SHAPES + SIZED:
        data = make_controller(spec[0], T=spec[1], terminal=spec[2], backend=_NoBackend()).problem_data() if isinstance(spec[0], str) else problem(*spec)[0]
        got = jit_prebuild(data)
        if verbose:
            print(spec, [os.path.basename(p) for p in got], flush=True)
        paths += got
    cache = os.path.join(os.path.dirname(LIBRARY_PATH), 'jit_cache')
    if prune and os.path.isdir(cache) and paths and all(os.path.dirname(q) == cache for q in paths):
        keep = set(os.path.basename(q) for q in paths)
        for f in os.listdir(cache):
            if f not in keep and f.endswith('.so'):            # (the VALIDATED manifest stays)
                os.remove(os.path.join(cache, f))
    return paths


if __name__ == '__main__':
    prewarm(verbose=True, prune='--prune' in sys.argv)
