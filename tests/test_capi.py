"""The C-ABI shared library: it loads, exports every symbol of include/hmpc.h, and the product
path fails loudly (no CPU fallback) where there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import make_controller, _NoBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_exports_header_symbols():
    from warm_start_hmpc_amd.qp_backend import LIBRARY_PATH, EXPORTED_SYMBOLS
    if not os.path.exists(LIBRARY_PATH):
        import __graft_entry__
        __graft_entry__.build()
    header = open(os.path.join(ROOT, 'include', 'hmpc.h')).read()
    declared = set(re.findall(r'\b(hmpc_[a-z_]+)\s*\(', header))
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    lib.hmpc_last_error.restype = ctypes.c_char_p
    assert lib.hmpc_last_error() == b''


def test_invalid_arguments_are_rejected_without_touching_the_gpu():
    from warm_start_hmpc_amd.qp_backend import load_library
    lib = load_library()
    out = ctypes.c_void_p()
    assert lib.hmpc_create(None, None, ctypes.byref(out)) == -1      # HMPC_EINVAL
    assert b'null' in lib.hmpc_last_error()
    ctrl = make_controller('cart_pole_with_walls', T=10, backend=_NoBackend())
    bad = ctrl.problem_data()
    bad['B'] = np.zeros((4, 3))
    from warm_start_hmpc_amd.qp_backend import HipBatchedQP
    with pytest.raises(ValueError):
        HipBatchedQP(bad)


@pytest.mark.skipif(_has_gpu(), reason='only meaningful on a box without a GPU')
def test_product_path_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError):
        make_controller('cart_pole_with_walls', T=10, backend='hip')
    # and the controller's default backend is the HIP one
    from warm_start_hmpc_amd.mld_system import MLDSystem
    from warm_start_hmpc_amd.controller import HybridModelPredictiveController
    from helpers import load_fixture
    d = load_fixture('cart_pole_with_walls')
    mld = MLDSystem([d['A'], d['B']], [d['F'], d['G'], d['h']], int(d['nub']))
    with pytest.raises(RuntimeError):
        HybridModelPredictiveController(mld, 10, [d['Q'], d['R'], d['Q_T']], None)


def _build_c_example(tmp_path):
    import subprocess
    from warm_start_hmpc_amd.qp_backend import LIBRARY_PATH
    exe = str(tmp_path / 'c_abi_example')
    libdir = os.path.dirname(LIBRARY_PATH)
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_abi_example.c'), '-L', libdir, '-lhmpc', '-Wl,-rpath,' + libdir, '-lm', '-o', exe]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(_has_gpu(), reason='only meaningful on a box without a GPU')
def test_header_is_plain_c_and_a_c_caller_fails_loudly_without_gpu(tmp_path):
    # include/hmpc.h is the boundary for callers in any language: it must compile as strict C99, and a C program linked
    # against libhmpc.so must get an error code and a message -- not a crash, not a CPU answer -- where there is no GPU
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'hmpc_create' in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_caller_gets_the_known_answers(tmp_path):
    # examples/c_abi_example.c: create / solve three nodes / infeasible node / facet LPs / destroy from plain C
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'c_abi_example: ok' in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_headline_kernel_keeps_its_register_budget(tmp_path):
    # The one-wave-per-node kernel of the cart-pole shape lives at the edge of the register file (256 + 256 registers per
    # lane, ~120 B of scratch): three times this round an edit that only added a flag or a branch elsewhere moved its
    # register allocation and cost 6-8 % of every launch.  The scratch size of that instantiation, as the
    # compiler reports it, is held here (cross-compiles without a GPU, ~15 s).
    import re
    import subprocess
    csrc = os.path.join(ROOT, 'warm-start-hybrid-mpc_amd', 'csrc')
    src = tmp_path / 'probe.hip'
    src.write_text('#define HMPC_KERNEL_ONLY\n#include "hmpc_device.h"\n#include "hmpc_kernel.hip"\n'
                   'template __global__ void hmpc_qp_kernel<4, 7, 4, 10, 3, 2, 1>(const DevProb, const double *, int, const int8_t *, int, '
                   'const DevOut, double *, double *, const int32_t *, const DevWarm);\n')
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-Wno-pass-failed',
                        '-I', os.path.join(ROOT, 'include'), '-I', csrc, '-c', str(src), '-o', str(tmp_path / 'probe.o'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = [int(v) for v in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', r.stderr)]
    assert scratch and max(scratch) <= 128, scratch
    # The streaming form of the generic kernel (BASELINE configs[4]) must not spill at all: a second instantiation of its
    # tile code once cost the whole kernel 650 B of scratch per lane -- 134 KB per workgroup pushed through an L2 the
    # factor slabs already overflow.
    src.write_text('#define HMPC_KERNEL_ONLY\n#include "hmpc_device.h"\n#include "hmpc_kernel.hip"\n'
                   'template __global__ void hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 4>(const DevProb, const double *, int, const int8_t *, int, '
                   'const DevOut, double *, double *, const int32_t *, const DevWarm);\n')
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-Wno-pass-failed',
                        '-I', os.path.join(ROOT, 'include'), '-I', csrc, '-c', str(src), '-o', str(tmp_path / 'probe.o'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = [int(v) for v in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', r.stderr)]
    assert scratch and max(scratch) == 0, scratch


def test_register_kernel_is_compiled_for_an_arbitrary_shape(monkeypatch):
    # The reference takes any MLDSystem (warm_start_hmpc/controller.py:58-117); the fast kernel here is a compile-time
    # instantiation.  For a shape without a built-in one hmpc_create compiles it from the same source into an on-disk
    # cache (csrc/hmpc_jit.h); hmpc_jit_build does the same without a GPU.  A random MLD with nx = 6, nu = 2 + 3 -- three
    # binaries: their six bound rows do not tile a wavefront --: four shared objects (1 / 2 / 4 waves per node, the one-wave
    # kernel also built for two waves per SIMD), each
    # exporting the getter of its two kernels; a second call is a cache hit.
    import time
    from helpers import random_mld, _NoBackend
    from warm_start_hmpc_amd.controller import HybridModelPredictiveController
    from warm_start_hmpc_amd.qp_backend import jit_shapes, jit_prebuild
    monkeypatch.setenv('HMPC_JIT_SIZED', '0')      # (the per-shape kernels; the default -- per problem -- in the next test)
    mld, objective, x0 = random_mld(nx=6, nuc=2, nub=3, seed=3)
    ctrl = HybridModelPredictiveController(mld, 8, objective, None, backend=_NoBackend())
    shapes = jit_shapes(ctrl.problem_data())
    assert [s[:3] + s[6:] for s in shapes] == [(6, 5, 3, 1, 8), (6, 5, 3, 2, 8), (6, 5, 3, 4, 8)]
    paths = jit_prebuild(ctrl.problem_data())
    assert len(paths) == 4 and all(os.path.exists(p) for p in paths)      # (1 wave: also its build for two waves per SIMD)
    for p in paths:
        assert hasattr(ctypes.CDLL(p), 'hmpc_jit_kernels')
    tic = time.perf_counter()
    assert jit_prebuild(ctrl.problem_data()) == paths and time.perf_counter() - tic < 2.0      # (cache hits)
    # shapes the static row map does not hold, and the built-in ones, are not compiled
    big = HybridModelPredictiveController(random_mld()[0], 30, random_mld()[1], None, backend=_NoBackend())
    assert jit_shapes(big.problem_data()) == []


def test_run_time_sized_kernel_is_compiled_with_the_sizes_of_a_problem():
    # Problems the static row map does not hold run the run-time-sized kernel (or, beyond one CU's LDS, its streaming form)
    # COMPILED WITH THEIR SIZES as constants -- same source, same code paths, hmpc_jit.h; 1.4x on BASELINE configs[4].
    # hmpc_jit_build_problem runs the host side of hmpc_create without a GPU: configs[4] gets the streaming form for four
    # waves per node (the only wave count that form runs), a random MLD with nx + nu = 18 the kernel for 1 / 2 / 4 waves.
    import time
    from jit_problems import problem, SIZED
    from warm_start_hmpc_amd.qp_backend import jit_prebuild
    from jit_problems import REGISTER_SHAPES
    cases = list(zip(SIZED, (['hmpc_s_stream_w4'], ['hmpc_s_generic_w1', 'hmpc_s_generic_w2', 'hmpc_s_generic_w4'], ['hmpc_s_generic_w1', 'hmpc_s_generic_w2', 'hmpc_s_generic_w4'])))
    # ... and where the static row map holds the problem, the register kernel with the row slots its horizon needs
    cases.append((REGISTER_SHAPES[0], ['hmpc_s_reg_6_5_3_4_1_1_w1_kc8_o2', 'hmpc_s_reg_6_5_3_2_1_1_w2_kc8', 'hmpc_s_reg_6_5_3_1_1_1_w4_kc8']))
    for spec, names in cases:
        data = problem(*spec)[0]
        paths = jit_prebuild(data)
        assert [os.path.basename(p)[:len(n)] for p, n in zip(paths, names)] == names and len(paths) == len(names), paths
        for p in paths:
            lib = ctypes.CDLL(p)
            assert hasattr(lib, 'hmpc_jit_kernels')
            lib.hmpc_jit_sized.restype = ctypes.c_char_p
            fields = lib.hmpc_jit_sized().decode()
            nx, nuc, nub, _, T = spec
            assert ('p.nx=%d;p.nu=%d;p.nub=%d;' % (nx, nuc + nub, nub)) in fields and ('p.T=%d;' % T) in fields
        tic = time.perf_counter()
        assert jit_prebuild(data) == paths and time.perf_counter() - tic < 2.0                    # (cache hits)


def test_two_processes_compile_the_same_problem_into_an_empty_cache(tmp_path):
    # one process per GPU (bench.py --gpus N, torch.distributed): on a cold cache every rank compiles the same kernels at the
    # same time -- each into a file of its own, moved into place atomically; both end with the same, loadable shared objects
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import conftest; from helpers import make_controller, _NoBackend; "
            "from warm_start_hmpc_amd.qp_backend import jit_prebuild; "
            "print('\\n'.join(jit_prebuild(make_controller('cart_pole_one_wall', T=10, backend=_NoBackend()).problem_data())))"
            % os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HMPC_JIT_CACHE=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    paths = [o[0].split() for o in outs]
    assert paths[0] == paths[1] and len(paths[0]) == 2 and all(q.startswith(str(tmp_path)) for q in paths[0])   # (2 and 4 waves per node)
    for q in paths[0]:
        assert hasattr(ctypes.CDLL(q), 'hmpc_jit_kernels')
    left = sorted(f for f in os.listdir(tmp_path) if not f.endswith('.so'))
    assert left == [], left                                                                    # no temporary files stay behind
