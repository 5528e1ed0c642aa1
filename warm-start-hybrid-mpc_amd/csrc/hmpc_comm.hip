// hmpc_comm.hip -- the one collective of the path behind the C ABI: all-reduce of the incumbent over RCCL.
//
// The reference is single-process; sharding a frontier over the GPUs of a node is new.  Nodes are
// independent, so the only exchange is, once per branch-and-bound round, MIN over two float64 per rank,
// (upper bound, -open candidates): every rank then prunes against the global best and all ranks stop in the same round
// (the Python form is warm_start_hmpc_amd/distributed.py::IncumbentExchange); once per search the owner of the global
// incumbent is determined and its binary assignment broadcast (hmpc_publish_incumbent).  8-byte messages: latency bound on xGMI,
// bandwidth is irrelevant.  RCCL is bound at run time (dlopen): a process that never creates a communicator does not
// need the library, and a process that already loaded RCCL (PyTorch) shares that copy.
#include <dlfcn.h>

struct hmpc_comm {
    hmpc_handle *h = nullptr;
    void *lib = nullptr;
    void *comm = nullptr; // ncclComm_t
    double *d_pair = nullptr, *h_pair = nullptr;
    int8_t *d_bytes = nullptr;
    size_t cap_bytes = 0;
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

namespace {
struct RcclId { char bytes[128]; }; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value to ncclCommInitRank
void *rccl_open()
{
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (void *l = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) return l;
    }
    return nullptr;
}
} // namespace

extern "C" int hmpc_comm_unique_id(void *id128)
{
    g_err.clear();
    if (!id128) return fail(HMPC_EINVAL, "comm: null id buffer");
    void *lib = rccl_open();
    if (!lib) return fail(HMPC_EDEVICE, std::string("comm: cannot load RCCL: ") + dlerror());
    auto get = (int (*)(RcclId *))dlsym(lib, "ncclGetUniqueId");
    if (!get) return fail(HMPC_EDEVICE, "comm: ncclGetUniqueId not found");
    const int rc = get((RcclId *)id128);
    return rc == 0 ? HMPC_OK : fail(HMPC_EDEVICE, "comm: ncclGetUniqueId failed");
}

extern "C" int hmpc_comm_create(hmpc_handle *h, int32_t nranks, int32_t rank, const void *id128, hmpc_comm **out)
{
    g_err.clear();
    if (!h || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(HMPC_EINVAL, "comm: bad argument");
    HIPCHK(hipSetDevice(h->device));
    hmpc_comm *c = new hmpc_comm();
    c->h = h;
    c->rank = rank;
    c->nranks = nranks;
    c->lib = rccl_open();
    if (!c->lib) { delete c; return fail(HMPC_EDEVICE, "comm: cannot load RCCL"); }
    auto init = (int (*)(void **, int, RcclId, int))dlsym(c->lib, "ncclCommInitRank");
    c->AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(c->lib, "ncclAllReduce");
    c->Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(c->lib, "ncclBroadcast");
    c->CommDestroy = (int (*)(void *))dlsym(c->lib, "ncclCommDestroy");
    c->GetErrorString = (const char *(*)(int))dlsym(c->lib, "ncclGetErrorString");
    if (!init || !c->AllReduce || !c->Broadcast || !c->CommDestroy) { delete c; return fail(HMPC_EDEVICE, "comm: RCCL symbols not found"); }
    RcclId id;
    std::memcpy(id.bytes, id128, sizeof id.bytes);
    const int rc = init(&c->comm, nranks, id, rank);
    if (rc != 0) {
        std::string msg = std::string("comm: ncclCommInitRank failed: ") + (c->GetErrorString ? c->GetErrorString(rc) : "?");
        delete c;
        return fail(HMPC_EDEVICE, msg);
    }
    // (everything a later collective needs is allocated here: an allocation that fails between two collectives would
    // leave the other ranks waiting in the next one)
    c->cap_bytes = (size_t)std::max(1, h->dp.T * h->dp.nub);
    if (hipMalloc((void **)&c->d_pair, 2 * sizeof(double)) != hipSuccess || hipHostMalloc((void **)&c->h_pair, 2 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->d_bytes, c->cap_bytes) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        hmpc_comm_destroy(c);
        return fail(HMPC_EDEVICE, "comm: cannot allocate");
    }
    *out = c;
    return HMPC_OK;
}

extern "C" int hmpc_allreduce_incumbent(hmpc_comm *c, double *ub, int32_t *open)
{
    g_err.clear();
    if (!c || !ub || !open) return fail(HMPC_EINVAL, "comm: null argument");
    HIPCHK(hipSetDevice(c->h->device));
    c->h_pair[0] = *ub;
    c->h_pair[1] = -(double)*open;
    HIPCHK(hipMemcpyAsync(c->d_pair, c->h_pair, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int rc = c->AllReduce(c->d_pair, c->d_pair, 2, /* ncclFloat64 */ 8, /* ncclMin */ 3, c->comm, c->stream);
    if (rc != 0) return fail(HMPC_EDEVICE, std::string("comm: ncclAllReduce failed: ") + (c->GetErrorString ? c->GetErrorString(rc) : "?"));
    HIPCHK(hipMemcpyAsync(c->h_pair, c->d_pair, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *ub = c->h_pair[0];
    *open = (int32_t)std::llround(-c->h_pair[1]);
    return HMPC_OK;
}

// The same exchange without the host in it: pair[0] = upper bound, pair[1] = -(open candidates), two float64 in DEVICE
// memory, reduced in place (MIN) by a collective enqueued on the caller's stream -- the stream the solves of the round were
// launched on --; nothing is copied and nobody waits: the next launch on that stream sees the global bound.  (The
// host-pointer form above costs a copy up, a copy down and a stream synchronisation per round: ~30 us of host time in
// which the GPU idles; at 128 nodes per GPU -- BASELINE configs[2] on 8 GPUs -- a round is ~2 ms.)
extern "C" int hmpc_allreduce_incumbent_device(hmpc_comm *c, double *pair_device, void *stream)
{
    g_err.clear();
    if (!c || !pair_device) return fail(HMPC_EINVAL, "comm: null argument");
    HIPCHK(hipSetDevice(c->h->device));
    const int rc = c->AllReduce(pair_device, pair_device, 2, /* ncclFloat64 */ 8, /* ncclMin */ 3, c->comm, (hipStream_t)stream);
    if (rc != 0) return fail(HMPC_EDEVICE, std::string("comm: ncclAllReduce failed: ") + (c->GetErrorString ? c->GetErrorString(rc) : "?"));
    return HMPC_OK;
}

// Who owns the global incumbent, and its binary assignment on every rank ("pack (ub, rank) ... or
// follow with ncclBroadcast of the <= T nub byte winning identifier from the owner").  One all-reduce (MIN) of
// (ub, rank if this rank's bound equals ... ) cannot be had in one step without knowing the minimum, so: MIN over the
// bounds, then MIN over the ranks that hold it, then one broadcast of the owner's bytes -- three 8 / 8 / T nub byte
// messages, once per search (not per round).
extern "C" int hmpc_publish_incumbent(hmpc_comm *c, double *ub, int8_t *assignment, int32_t nbytes, int32_t *owner)
{
    g_err.clear();
    if (!c || !ub || !owner || nbytes < 0 || (nbytes > 0 && !assignment)) return fail(HMPC_EINVAL, "comm: bad argument");
    HIPCHK(hipSetDevice(c->h->device));
    auto reduce_min = [&](double v, double &out) -> int {
        c->h_pair[0] = v;
        HIPCHK(hipMemcpyAsync(c->d_pair, c->h_pair, sizeof(double), hipMemcpyHostToDevice, c->stream));
        const int rc = c->AllReduce(c->d_pair, c->d_pair, 1, /* ncclFloat64 */ 8, /* ncclMin */ 3, c->comm, c->stream);
        if (rc != 0) return fail(HMPC_EDEVICE, std::string("comm: ncclAllReduce failed: ") + (c->GetErrorString ? c->GetErrorString(rc) : "?"));
        HIPCHK(hipMemcpyAsync(c->h_pair, c->d_pair, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        out = c->h_pair[0];
        return HMPC_OK;
    };
    // (a failure between two collectives aborts the communicator: the peers are already in the next one.  Nothing below
    // can fail for a reason of this rank alone -- the staging buffer is allocated at creation --, and a bound that is no
    // bound is seen by EVERY rank after the first reduction: all of them return the error together)
    if ((size_t)nbytes > c->cap_bytes) return fail(HMPC_EINVAL, "comm: assignment longer than T * nub of the handle the communicator was created on");
    const double mine = (*ub == *ub) ? *ub : -std::numeric_limits<double>::infinity(); // (NaN: no order under MIN)
    double best = 0, who = 0;
    int rc;
    if ((rc = reduce_min(mine, best))) return rc;
    if (best == -std::numeric_limits<double>::infinity())
        return fail(HMPC_EINVAL, "comm: a rank published -inf / NaN as its upper bound (the abort convention of hmpc_allreduce_incumbent): no incumbent is published");
    // the lowest rank among those that hold the best bound (ties between ranks are possible: equal optima)
    if ((rc = reduce_min((mine == best && std::isfinite(mine)) ? (double)c->rank : (double)c->nranks, who))) return rc;
    *ub = best;
    if (who >= (double)c->nranks) { *owner = -1; return HMPC_OK; } // no rank has an incumbent: the problem is infeasible
    *owner = (int32_t)who;
    if (nbytes == 0) return HMPC_OK;
    if (c->rank == *owner) HIPCHK(hipMemcpyAsync(c->d_bytes, assignment, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    rc = c->Broadcast(c->d_bytes, c->d_bytes, (size_t)nbytes, /* ncclInt8 */ 0, *owner, c->comm, c->stream);
    if (rc != 0) return fail(HMPC_EDEVICE, std::string("comm: ncclBroadcast failed: ") + (c->GetErrorString ? c->GetErrorString(rc) : "?"));
    HIPCHK(hipMemcpyAsync(assignment, c->d_bytes, (size_t)nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HMPC_OK;
}

extern "C" int hmpc_comm_destroy(hmpc_comm *c)
{
    if (!c) return HMPC_OK;
    (void)hipSetDevice(c->h->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    if (c->d_pair) (void)hipFree(c->d_pair);
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    if (c->h_pair) (void)hipHostFree(c->h_pair);
    delete c;
    return HMPC_OK;
}
