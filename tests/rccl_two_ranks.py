"""One rank of the two-process RCCL check of the C ABI's incumbent exchange (tests/test_fleet.py spawns two of these, each
a FRESH process bound to its own GPU before anything touches the device):

    python tests/rccl_two_ranks.py RANK NRANKS ID_FILE

Rank 0 creates the communicator id (hmpc_comm_unique_id) and writes its 128 bytes to ID_FILE (the caller's transport);
every rank then creates its handle on device RANK and the communicator (hmpc_comm_create) and goes through
  1. hmpc_allreduce_incumbent   MIN semantics of (upper bound, open candidates)
  2. the abort convention       one rank contributes -inf, every rank reads -inf
  3. hmpc_publish_incumbent     owner = lowest rank that holds the best bound, its assignment on every rank; a tie; no incumbent;
                                a bound of -inf on one rank: HMPC_EINVAL on every rank
  4. hmpc_allreduce_incumbent_device   the exchange of 1. on device memory, enqueued on the caller's stream
and prints one JSON line with what it saw."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'warm-start-hybrid-mpc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def main():
    rank, nranks, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import numpy as np
    from helpers import make_controller
    ctrl = make_controller('cart_pole_with_walls', T=10, backend='hip', device=rank)
    qp = ctrl.qp
    lib = qp.lib
    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        qp._check(lib.hmpc_comm_unique_id(ident))
        with open(id_file + '.tmp', 'wb') as f:
            f.write(ident.raw)
        os.rename(id_file + '.tmp', id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise SystemExit('no communicator id after 120 s')
            time.sleep(0.05)
        with open(id_file, 'rb') as f:
            ident.raw = f.read()
    comm = ctypes.c_void_p()
    qp._check(lib.hmpc_comm_create(qp.handle, nranks, rank, ident, ctypes.byref(comm)))
    seen = {}

    def allreduce(ub, n_open):
        a, b = ctypes.c_double(ub), ctypes.c_int32(n_open)
        qp._check(lib.hmpc_allreduce_incumbent(comm, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def publish(ub, value):
        a, owner = ctypes.c_double(ub), ctypes.c_int32(-7)
        bits = np.full(40, value, dtype=np.int8)
        qp._check(lib.hmpc_publish_incumbent(comm, ctypes.byref(a), bits.ctypes.data, bits.size, ctypes.byref(owner)))
        return a.value, owner.value, bits.tolist()

    def allreduce_device(ub, n_open):
        # the same exchange on device memory and on the caller's stream (hmpc_allreduce_incumbent_device): nothing copied,
        # nobody waits until this test reads the pair back
        import torch
        dev = torch.device('cuda', rank)
        pair = torch.tensor([ub, -float(n_open)], dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev)
        qp._check(lib.hmpc_allreduce_incumbent_device(comm, ctypes.c_void_p(pair.data_ptr()), ctypes.c_void_p(stream.cuda_stream)))
        after = pair * 1.0                                               # (work enqueued on the same stream sees the global pair)
        stream.synchronize()
        return float(after[0].item()), int(round(-float(after[1].item())))

    def publish_error(ub):
        a, owner = ctypes.c_double(ub), ctypes.c_int32(-7)
        bits = np.zeros(40, dtype=np.int8)
        return int(lib.hmpc_publish_incumbent(comm, ctypes.byref(a), bits.ctypes.data, bits.size, ctypes.byref(owner)))

    seen['min'] = allreduce(3.0 - rank, 5 if rank == 0 else 0)             # ranks hold 3, 2, ...: the last rank wins
    seen['min_device'] = allreduce_device(3.0 - rank, 5 if rank == 0 else 0)
    seen['abort'] = allreduce(float('-inf') if rank == nranks - 1 else 1.0, 1)
    seen['publish'] = publish(2.5 - 0.5 * rank, rank)                      # the last rank owns the incumbent
    seen['tie'] = publish(1.0, 10 + rank)                                  # equal bounds: the lowest rank
    seen['none'] = publish(float('inf'), 20 + rank)                        # no incumbent anywhere
    seen['poisoned'] = publish_error(float('-inf') if rank == nranks - 1 else 1.0)   # every rank gets the error, none blocks
    # the solver still works beside the communicator
    r = qp.solve_batch(np.array([0., 0., .5, 0.]), np.full((1, 40), -1, dtype=np.int8))
    seen['root_status'] = int(r['status'][0])
    qp._check(lib.hmpc_comm_destroy(comm))
    print(json.dumps(seen, default=lambda v: str(v)))


if __name__ == '__main__':
    main()
