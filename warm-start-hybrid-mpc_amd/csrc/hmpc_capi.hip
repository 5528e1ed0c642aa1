// hmpc_capi.hip -- host side of the C ABI declared in include/hmpc.h.
//
// hmpc_create   : scales the stage constraints, builds the sparse row / column / Gram lists the
//                 kernel walks, uploads everything once (the role of controller.py:119-184, which
//                 builds the Gurobi model once per controller).
// hmpc_solve_*  : one kernel launch per batch of nodes (replaces B sequential calls of
//                 controller.py:229-271 + bounded_qp.py:200-228).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hmpc_device.h"

#include "hmpc_kernel.hip" // one translation unit: the kernels are launched from this file
#include "hmpc_jit.h"      // register kernels for shapes without a built-in instantiation, compiled at hmpc_create
#include "hmpc_shift.hip"

#define HMPC_CHECK_NODES 64 // (even) nodes of the first-use check of a kernel compiled at hmpc_create (hmpc_check_compiled)
static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(HMPC_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct hmpc_cfg { // the kernel used for 1 / 2 / 4 waves per node, its LDS carve and resident-node count
    hmpc_kernel_choice k{};
    size_t lds = 0;
    int max_grid = 0;
    int sized = 0; // k is the run-time-sized kernel compiled with this problem's sizes (hmpc_jit_prepare_sized)
    int ilp = 0;   // ... with the compiler's ILP schedule: a binary the cache's VALIDATED manifest lists (hmpc_jit.h: sched_flags)
    // FIRST-USE CHECK of a kernel compiled at hmpc_create: `ref` is the shipped kernel that would serve this wave count without
    // the run-time compiler; the first launch through this configuration solves its first few nodes with both and compares
    // statuses and objectives (hmpc_check_compiled).  A kernel that disagrees is dropped for the handle.
    hmpc_kernel_choice ref{};
    size_t ref_lds = 0;
    int ref_grid = 0;
    int checked = 0; // 0 not yet, 1 agreed, -1 disagreed (ref serves)
    int second_opinions = 0; // batches with MAXITER / NUMERICAL nodes that the shipped kernel solved again and ended the same way (hmpc_solve_batch_device)
};

struct hmpc_handle {
    int device = 0;
    bool dry = false; // hmpc_jit_build_problem: the host side of hmpc_create without a device (nothing uploaded, nothing launched)
    hmpc_cfg cfg[3];
    DevProb dp{};
    std::vector<void *> allocs;
    double *rows_ws = nullptr;
    int32_t *order = nullptr; // processing order of large frontiers (hmpc_order_kernel)
    int order_cap = 0;
    int32_t *pend = nullptr;  // two-launch form of the lazy terminal set: [0] how many nodes wait for their second solve, [1 ..] which
    int pend_cap = 0;
    void *d_shift = nullptr; // staging of the host-pointer shift
    size_t shift_staged = 0;
    double *shift_tv = nullptr; // per tree: what the shift needs of (x0, u0) only (hmpc_shift_tree_kernel)
    double *shift_MT2 = nullptr; // M_mu in pairs of columns, as hmpc_shift_row_kernel keeps it in LDS
    size_t shift_tv_cap = 0;
    double *trace = nullptr;
    size_t lds = 0;
    int max_grid = 0, last_grid = 0;
    // staging for the host-pointer entry point
    void *d_x0 = nullptr;    // one device block (inputs, then outputs: stage_layout)
    void *h_stage = nullptr; // its pinned host mirror
    int staged = 0;
    bool staged_warm = false; // the blocks have room for one handed-down parent record per node
    int last_cfg = -1;            // configuration (0, 1, 2: 1 / 2 / 4 waves per node) of the last launch
    // SECOND OPINION (hmpc_solve_batch_device): nodes a compiled kernel leaves undecided are listed on the device and solved again
    // by the shipped kernel in the same stream.  hard: [0] how many of them the shipped kernel leaves undecided too, [1] its work
    // counter, [2] how many the compiled kernel left, [3 ..] which.  The two counts of the last call travel to h_hard (pinned)
    // behind hard_done and are looked at when the next call comes, or when a caller that has synchronised asks (hmpc_second_opinion_review).
    int32_t *hard = nullptr;
    int hard_cap = 0;
    int32_t *h_hard = nullptr;
    hipEvent_t hard_done = nullptr;
    int hard_cfg = -1;            // configuration the counts in flight belong to (-1: none)
    int second_runs = 0;          // calls in which the shipped kernel was asked (for the tests)
    void *chk = nullptr;          // device block of the first-use check: 2 x HMPC_CHECK_NODES x (obj, dual_obj, status, iters)
    void *h_chk = nullptr;        //   its PINNED host mirror (objectives, dual objectives, statuses of the three runs, the hand-down index)
    int jit_rejected = 0;         //   compiled kernels dropped by it
    std::vector<void *> jit_libs; // shared objects of kernels compiled for this problem's shape (hmpc_jit.h); never unloaded
    int jit_kernels = 0;          //   how many of the three wave counts run on such a kernel (hmpc_kernel_info)
};

namespace {

struct StageHost {
    int m, mg;
    std::vector<double> C, h, scale; // dense m x nz, scaled
    std::vector<int> rptr, rcol, cptr, crow, gptr, grow;
    std::vector<double> rval, cval, gval;
};

// [F G] rows scaled to unit norm, then the bound rows of the binaries.
void build_stage(const hmpc_problem &q, const double *F, const double *G, const double *h, int nrow, StageHost &s)
{
    const int nx = q.nx, nu = q.nu, nz = nx + nu, nub = q.nub, nuc = nu - nub;
    s.mg = nrow;
    s.m = nrow + 2 * nub;
    s.C.assign((size_t)s.m * nz, 0.0);
    s.h.assign(s.m, 0.0);
    s.scale.assign(nrow > 0 ? nrow : 1, 1.0);
    for (int r = 0; r < nrow; r++) {
        double n2 = 0;
        for (int j = 0; j < nx; j++) n2 += F[r * nx + j] * F[r * nx + j];
        for (int j = 0; j < nu; j++) n2 += G[r * nu + j] * G[r * nu + j];
        const double sc = n2 > 0 ? 1.0 / std::sqrt(n2) : 1.0;
        s.scale[r] = sc;
        for (int j = 0; j < nx; j++) s.C[(size_t)r * nz + j] = sc * F[r * nx + j];
        for (int j = 0; j < nu; j++) s.C[(size_t)r * nz + nx + j] = sc * G[r * nu + j];
        s.h[r] = sc * h[r];
    }
    for (int i = 0; i < nub; i++) {
        s.C[(size_t)(nrow + i) * nz + nx + nuc + i] = -1.0; // -ub <= 0
        s.C[(size_t)(nrow + nub + i) * nz + nx + nuc + i] = 1.0; // ub <= 1
        s.h[nrow + nub + i] = 1.0;
    }
    // rows
    s.rptr.assign(1, 0);
    for (int r = 0; r < s.m; r++) {
        for (int j = 0; j < nz; j++)
            if (s.C[(size_t)r * nz + j] != 0.0) { s.rcol.push_back(j); s.rval.push_back(s.C[(size_t)r * nz + j]); }
        s.rptr.push_back((int)s.rcol.size());
    }
    // columns
    s.cptr.assign(1, 0);
    for (int j = 0; j < nz; j++) {
        for (int r = 0; r < s.m; r++)
            if (s.C[(size_t)r * nz + j] != 0.0) { s.crow.push_back(r); s.cval.push_back(s.C[(size_t)r * nz + j]); }
        s.cptr.push_back((int)s.crow.size());
    }
    // Gram lists over the lower triangle, entry e = i (i + 1) / 2 + j
    s.gptr.assign(1, 0);
    for (int i = 0; i < nz; i++)
        for (int j = 0; j <= i; j++) {
            for (int r = 0; r < s.m; r++) {
                const double v = s.C[(size_t)r * nz + i] * s.C[(size_t)r * nz + j];
                if (v != 0.0) { s.grow.push_back(r); s.gval.push_back(v); }
            }
            s.gptr.push_back((int)s.grow.size());
        }
}

template <class T>
int upload(hmpc_handle *h, const std::vector<T> &v, const T **out)
{
    void *d = nullptr;
    if (h->dry) { *out = nullptr; return HMPC_OK; }
    const size_t bytes = (v.size() ? v.size() : 1) * sizeof(T);
    HIPCHK(hipMalloc(&d, bytes));
    h->allocs.push_back(d);
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(d);
    return HMPC_OK;
}

int upload_stage(hmpc_handle *h, const StageHost &s, SparseStage &d)
{
    d.m = s.m;
    d.mg = s.mg;
    int rc;
    if ((rc = upload(h, s.rptr, &d.rptr))) return rc;
    if ((rc = upload(h, s.rcol, &d.rcol))) return rc;
    if ((rc = upload(h, s.rval, &d.rval))) return rc;
    if ((rc = upload(h, s.cptr, &d.cptr))) return rc;
    if ((rc = upload(h, s.crow, &d.crow))) return rc;
    if ((rc = upload(h, s.cval, &d.cval))) return rc;
    if ((rc = upload(h, s.gptr, &d.gptr))) return rc;
    if ((rc = upload(h, s.grow, &d.grow))) return rc;
    if ((rc = upload(h, s.gval, &d.gval))) return rc;
    if ((rc = upload(h, s.h, &d.h))) return rc;
    if ((rc = upload(h, s.scale, &d.scale))) return rc;
    return HMPC_OK;
}

// Wave counts for which this problem gets a register kernel compiled with its sizes (hmpc_jit_prepare_sized): placeholders
// (waves, kc) for hmpc_pick_kernel where the static row map holds the problem -- nx + nu <= 16, every [F G] row with
// at most two input coefficients, columns of at most HMPC_KC_STRIDE entries, at most 128 Gram entries with terms, at least one
// binary, at most 16 row slots per lane.  The two cart-pole shapes have built-in instantiations, which hmpc_pick_kernel finds
// itself; other problems leave jit empty and take the run-time-sized kernel.
void hmpc_jit_register_shapes(const DevProb &p, hmpc_kernel_choice (&jit)[3], size_t lds_cu)
{
    if (getenv("HMPC_FORCE_GENERIC") || getenv("HMPC_FORCE_BIG")) return;
    if ((p.nx == 4 && p.nu == 7 && p.nub == 4) || (p.nx == 4 && p.nu == 4 && p.nub == 2)) return; // (built in)
    if (!p.static_rows || p.nz > 16 || p.nub < 1) return;
    const int kc = std::max(2, (p.kcol + 1) / 2 * 2);
    if (kc > HMPC_KC_STRIDE || hmpc_lds_bytes(p, kc, 0) > lds_cu) return;
    for (int c = 0; c < 3; c++) {
        int kf = 0, kb = 0, kt = 0;
        if (!hmpc_static_slots(p, 1 << c, kf, kb, kt)) continue;
        if (kt < 1) kt = 1;                 // (the row map keeps a terminal slot; a problem without terminal set leaves it empty)
        if (kf + kb + kt > 16) continue;    // (row state in registers: 4 doubles per slot and lane)
        jit[c] = {(hmpc_kernel_t)(uintptr_t)1, (hmpc_kernel_t)(uintptr_t)1, 1 << c, kc, 0};
    }
}

// The integer sizes of a problem as assignments to the fields of DevProb: the body of HMPC_SIZED(p) of a kernel compiled for
// this problem alone (hmpc_jit.h), and with it the key of its cache entry.  Everything the kernels read as an int that is
// fixed once the problem is: dimensions, row counts and their splits, list lengths, the magic numbers of the divisions,
// the LDS choices of hmpc_create.  Tolerances, options and pointers stay arguments.
std::string hmpc_sized_fields(const DevProb &p)
{
    char b[1024];
    snprintf(b, sizeof b,
             "p.nx=%d;p.nu=%d;p.nub=%d;p.nuc=%d;p.nz=%d;p.T=%d;p.nc=%d;p.ncL=%d;p.nT=%d;p.mreg=%d;p.Toff=%d;p.M=%d;p.Mpad=%d;p.n=%d;p.ne=%d;"
             "p.nq=%d;p.nr=%d;p.nqT=%d;p.n_primal=%d;p.n_dual=%d;p.mreg_magic=%uu;p.nnz0=%d;p.nng0=%d;p.reg.m=%d;p.reg.mg=%d;p.kcol=%d;p.ngram=%d;"
             "p.ring=%d;p.nd=%d;p.ndp=%d;p.ns=%d;p.nd_magic=%uu;p.nn_magic=%uu;p.split_lds=%d;p.static_rows=%d;p.polish_l1=%d;",
             p.nx, p.nu, p.nub, p.nuc, p.nz, p.T, p.nc, p.ncL, p.nT, p.mreg, p.Toff, p.M, p.Mpad, p.n, p.ne, p.nq, p.nr, p.nqT, p.n_primal, p.n_dual,
             p.mreg_magic, p.nnz0, p.nng0, p.reg.m, p.reg.mg, p.kcol, p.ngram, p.ring, p.nd, p.ndp, p.ns, p.nd_magic, p.nn_magic, p.split_lds,
             p.static_rows, p.polish_l1);
    return b;
}

static bool hmpc_sized_enabled()
{
    if (getenv("HMPC_FORCE_GENERIC") || getenv("HMPC_FORCE_BIG")) return false; // (the run-time-sized kernels themselves are asked for)
    if (const char *e = getenv("HMPC_JIT")) { if (atoi(e) == 0) return false; }
    if (const char *e = getenv("HMPC_JIT_SIZED")) { if (atoi(e) == 0) return false; }
    return true;
}

// SIZED KERNELS (hmpc_jit.h): whatever kernel hmpc_pick_kernel chose for a wave count -- a register kernel of the problem's
// shape (built in, or a placeholder of hmpc_jit_prepare), the run-time-sized kernel or its streaming form -- is compiled with
// this problem's sizes as constants (or fetched from the cache) and replaces the choice in `cfg`.  Register kernels get
// exactly the row slots the horizon needs.  The streaming form runs four waves per node whatever the batch
// (hmpc_solve_batch_device), so only that one is built.  Returns false if a placeholder could not be replaced (the caller
// then falls back to the kernels without sizes).
bool hmpc_jit_prepare_sized(const DevProb &p, hmpc_cfg (&cfg)[3], std::vector<void *> &libs, int &count_out, std::vector<std::string> *built)
{
    const std::string fields = hmpc_sized_fields(p);
    hmpc_jit_shape shapes[3];
    int slot[3], count = 0;
    int only = -1;
    if (const char *e = getenv("HMPC_WAVES")) { const int nw = atoi(e); only = nw == 1 ? 0 : nw == 2 ? 1 : nw == 4 ? 2 : -1; }
    for (int c = 0; c < 3; c++) {
        const hmpc_kernel_choice &k = cfg[c].k;
        if (k.kc > 0) {                                                 // register kernel: the static row map with the slots this horizon needs
            int kf = 0, kb = 0, kt = 0;
            if (!hmpc_static_slots(p, k.waves, kf, kb, kt)) continue;
            if (kt < 1) kt = 1;
            shapes[count] = hmpc_jit_shape{p.nx, p.nu, p.nub, kf, kb, kt, k.waves, k.kc, fields};
            slot[count++] = c;
            continue;
        }
        if (k.big && c != (only >= 0 ? only : 2)) continue;
        // Row state (slack, multiplier, two steps per row) in registers instead of the global slab where a lane holds at most 16
        // rows (HMPC_JIT_SIZED_ROWS: another limit, 0: never): the list row map with Rows<Mpad / threads>.  configs[4], 12 rows
        // per lane: HBM traffic per launch 65 -> 45 GB, 137.0 -> 133.6 ms (profiles/r04_c4_rows_ab.txt); 164 B of scratch per lane.
        int rs = p.Mpad / (WAVE << c), rs_max = 16;
        if (const char *e = getenv("HMPC_JIT_SIZED_ROWS")) rs_max = atoi(e);
        if (rs > rs_max) rs = 0;
        shapes[count] = hmpc_jit_shape{k.big ? -1 : 0, -1, 0, rs, 0, 0, 1 << c, 0, fields};
        slot[count++] = c;
    }
    // the ILP schedule for the binaries the VALIDATED manifest lists (hmpc_jit.h: sched_flags), the default schedule for all others
    if (!getenv("HMPC_JIT_SCHED")) {
        const uint64_t hsh = hmpc_jit::source_hash();
        for (int i = 0; i < count; i++) {
            hmpc_jit_shape probe = shapes[i];
            probe.ilp = 1;
            if (hmpc_jit::validated(hmpc_jit::name_of(probe, hsh))) shapes[i].ilp = 1;
        }
    }
    std::vector<std::string> paths;
    std::string err;
    if (count) (void)hmpc_jit_build_all(shapes, count, paths, err);
    for (int i = 0; i < count; i++) {
        if (paths[i].empty()) continue;
        cfg[slot[i]].ilp = shapes[i].ilp || (getenv("HMPC_JIT_SCHED") && std::string(getenv("HMPC_JIT_SCHED")) != "default");
        if (built) {                                                     // (dry run: built, not loaded)
            bool seen = false;
            for (const std::string &b : *built) seen |= b == paths[i];
            if (!seen) built->push_back(paths[i]);
            cfg[slot[i]].sized = 1;
            continue;
        }
        void *lib = dlopen(paths[i].c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("dlopen: ") + dlerror(); continue; }
        auto get = (void (*)(void **, void **))dlsym(lib, "hmpc_jit_kernels");
        auto what = (const char *(*)(void))dlsym(lib, "hmpc_jit_sized");
        if (!get || !what || fields != what()) { err = "not the kernel of this problem: " + paths[i]; continue; } // (a collision of the key's hash)
        void *cold = nullptr, *warm = nullptr;
        get(&cold, &warm);
        libs.push_back(lib);
        cfg[slot[i]].k.fn = (hmpc_kernel_t)cold;
        cfg[slot[i]].k.fn_warm = (hmpc_kernel_t)warm;
        cfg[slot[i]].sized = 1;
        count_out++;
    }
    if (!err.empty() && getenv("HMPC_JIT_VERBOSE")) fprintf(stderr, "hmpc: sized kernel for this problem not available (%s): the kernel without sizes serves it\n", err.c_str());
    for (int c = 0; c < 3; c++)
        if (!cfg[c].sized && cfg[c].k.fn == (hmpc_kernel_t)(uintptr_t)1) return false;
    return true;
}

} // namespace

extern "C" const char *hmpc_last_error(void) { return g_err.c_str(); }

// Diagnostic (HMPC_BACKTRACE=1): the native frames of a fatal signal on stderr, then the default action.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void hmpc_fatal_signal(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "hmpc: fatal signal, native frames of the faulting thread:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static void hmpc_install_backtrace()
{
    static const bool once = [] {
        if (getenv("HMPC_BACKTRACE")) { signal(SIGSEGV, hmpc_fatal_signal); signal(SIGABRT, hmpc_fatal_signal); signal(SIGBUS, hmpc_fatal_signal); }
        return true;
    }();
    (void)once;
}

// hmpc_create; with `built` the DRY form behind hmpc_jit_build_problem: the same host code up to the choice of kernels --
// which compiles what this problem's kernels need into the cache -- without a device: nothing is uploaded, no handle returned.
static int create_impl(const hmpc_problem *q, const hmpc_options *opt, hmpc_handle **out, std::vector<std::string> *built)
{
    g_err.clear();
    const bool dry = built != nullptr;
    if (!q || (!out && !dry)) return fail(HMPC_EINVAL, "null problem or output pointer");
    if (q->nx < 1 || q->nu < 1 || q->nub < 0 || q->nub > q->nu || q->T < 2 || q->nc < 0 || q->ncT < q->nc ||
        q->nq < 0 || q->nr < 0 || q->nqT < 0)
        return fail(HMPC_EINVAL, "inconsistent sizes (need nx,nu >= 1, 0 <= nub <= nu, T >= 2, ncT >= nc)");
    if (!q->A || !q->B || !q->F || !q->G || !q->h || !q->F_Tm1 || !q->G_Tm1 || !q->h_Tm1 || !q->Q || !q->R || !q->Q_T)
        return fail(HMPC_EINVAL, "null matrix pointer");

    hmpc_handle *h = new hmpc_handle();
    h->dry = dry;
    int dev = opt ? opt->device : -1;
    if (!dry) {
        if (dev < 0) {
            if (hipGetDevice(&dev) != hipSuccess) { delete h; return fail(HMPC_EDEVICE, "no HIP device available"); }
        }
        h->device = dev;
        if (hipSetDevice(dev) != hipSuccess) { delete h; return fail(HMPC_EDEVICE, "hipSetDevice failed"); }
    }

    DevProb &p = h->dp;
    const int nx = q->nx, nu = q->nu, nz = q->nx + q->nu;
    p.nx = q->nx; p.nu = q->nu; p.nub = q->nub; p.nuc = q->nu - q->nub; p.nz = q->nx + q->nu; p.T = q->T;
    p.nc = q->nc; p.ncL = q->ncT; p.nT = q->ncT - q->nc; p.mreg = q->nc + 2 * q->nub;
    p.Toff = q->T * p.mreg; p.M = p.Toff + p.nT; p.Mpad = (p.M + 255) / 256 * 256;
    p.n = q->T * p.nz + q->nx; p.ne = p.nz * (p.nz + 1) / 2;
    p.nq = q->nq; p.nr = q->nr; p.nqT = q->nqT;
    p.n_primal = (q->T + 1) * q->nx + q->T * q->nu;
    p.n_dual = (q->T + 1) * q->nx + (q->T - 1) * q->nc + q->ncT + 2 * q->T * q->nub + q->T * q->nq + q->nqT + q->T * q->nr;
    p.tol = opt && opt->tol > 0 ? opt->tol : 1e-8;
    p.tol_inf = opt && opt->tol_inf > 0 ? opt->tol_inf : 1e-6;
    p.max_iter = opt && opt->max_iter > 0 ? opt->max_iter : 100;
    p.lazy = opt ? opt->lazy_terminal : 1;
    p.refine = opt ? opt->refine : 1;
    p.dbg = getenv("HMPC_DBG") ? atoi(getenv("HMPC_DBG")) : 0;
    p.polish = opt ? opt->polish : 1;
    p.ptol = opt && opt->polish_tol > 0 ? opt->polish_tol : 1e-4;

    StageHost reg;
    build_stage(*q, q->F, q->G, q->h, q->nc, reg);
    p.mreg_magic = (unsigned)((0x100000000ULL + p.mreg - 1) / p.mreg);
    p.nnz0 = (int)reg.rcol.size();
    // columns of the stage rows padded to a fixed stride (compile-time shapes: static column products)
    std::vector<double> ccv((size_t)nz * HMPC_KC_STRIDE, 0.0);
    std::vector<int> cci((size_t)nz * HMPC_KC_STRIDE, 0);
    p.kcol = 0;
    for (int j = 0; j < nz; j++) {
        const int len = reg.cptr[j + 1] - reg.cptr[j];
        if (len > p.kcol) p.kcol = len;
        for (int k = 0; k < len && k < HMPC_KC_STRIDE; k++) {
            ccv[(size_t)j * HMPC_KC_STRIDE + k] = reg.cval[reg.cptr[j] + k];
            cci[(size_t)j * HMPC_KC_STRIDE + k] = reg.crow[reg.cptr[j] + k];
        }
    }
    p.static_rows = (p.kcol <= HMPC_KC_STRIDE && p.mreg <= 255) ? 1 : 0;
    for (int r = 0; r < q->nc; r++) {
        int cnt = 0;
        for (int j = 0; j < nu; j++) cnt += reg.C[(size_t)r * nz + nx + j] != 0.0;
        if (cnt > 2) p.static_rows = 0;
    }
    p.nng0 = (int)reg.grow.size();
    // The first nc rows of [F_Tm1 G_Tm1 | h_Tm1] must be the stage rows [F G | h] (controller.py:85-87):
    // the last stage then shares the stage lists and only the terminal-set rows are kept apart.
    for (int r = 0; r < q->nc; r++) {
        bool same = q->h_Tm1[r] == q->h[r];
        for (int j = 0; j < nx && same; j++) same = q->F_Tm1[r * nx + j] == q->F[r * nx + j];
        for (int j = 0; j < nu && same; j++) same = q->G_Tm1[r * nu + j] == q->G[r * nu + j];
        if (!same) { hmpc_destroy(h); return fail(HMPC_EINVAL, "the first nc rows of F_Tm1, G_Tm1, h_Tm1 must equal F, G, h"); }
    }
    // padded to a whole number of 256-row tiles (zero rows): a lane of the last row slot that has no terminal row
    // still addresses memory of these arrays
    const size_t nTpad = ((size_t)p.nT + 255) / 256 * 256 + 256;
    std::vector<double> Ct(nTpad * nz, 0.0), ht(nTpad, 0.0), sct(nTpad, 1.0);
    for (int k = 0; k < p.nT; k++) {
        const int r = q->nc + k;
        double n2 = 0;
        for (int j = 0; j < nx; j++) n2 += q->F_Tm1[r * nx + j] * q->F_Tm1[r * nx + j];
        for (int j = 0; j < nu; j++) n2 += q->G_Tm1[r * nu + j] * q->G_Tm1[r * nu + j];
        const double sc = n2 > 0 ? 1.0 / std::sqrt(n2) : 1.0;
        sct[k] = sc;
        for (int j = 0; j < nx; j++) Ct[(size_t)k * nz + j] = sc * q->F_Tm1[r * nx + j];
        for (int j = 0; j < nu; j++) Ct[(size_t)k * nz + nx + j] = sc * q->G_Tm1[r * nu + j];
        ht[k] = sc * q->h_Tm1[r];
    }

    // cost Hessians, scaled so that their largest entry is one
    std::vector<double> P((size_t)nz * nz, 0.0), PT((size_t)nx * nx, 0.0);
    double big = 0;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < nx; j++) {
            double a = 0, b = 0;
            for (int k = 0; k < q->nq; k++) a += q->Q[k * nx + i] * q->Q[k * nx + j];
            for (int k = 0; k < q->nqT; k++) b += q->Q_T[k * nx + i] * q->Q_T[k * nx + j];
            P[(size_t)i * nz + j] = 2 * a;
            PT[(size_t)i * nx + j] = 2 * b;
            big = std::fmax(big, std::fmax(std::fabs(2 * a), std::fabs(2 * b)));
        }
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nu; j++) {
            double a = 0;
            for (int k = 0; k < q->nr; k++) a += q->R[k * nu + i] * q->R[k * nu + j];
            P[(size_t)(nx + i) * nz + nx + j] = 2 * a;
            big = std::fmax(big, std::fabs(2 * a));
        }
    p.cs = big > 0 ? 1.0 / big : 1.0;
    for (auto &v : P) v *= p.cs;
    for (auto &v : PT) v *= p.cs;
    {   // curvature of the (scaled) cost: smallest positive diagonal entry -- decides the penalty level of the polish (DevProb)
        double cmin = 1.0;
        for (int i = 0; i < nz; i++) if (P[(size_t)i * nz + i] > 0) cmin = std::fmin(cmin, P[(size_t)i * nz + i]);
        for (int i = 0; i < nx; i++) if (PT[(size_t)i * nx + i] > 0) cmin = std::fmin(cmin, PT[(size_t)i * nx + i]);
        p.polish_l1 = cmin >= 1e-2 ? 1 : 0;
    }

    std::vector<int> ei, ej;
    for (int i = 0; i < nz; i++)
        for (int j = 0; j <= i; j++) { ei.push_back(i); ej.push_back(j); }
    // Number the entries with a nonempty Gram list first (the register factorisation gives one lane
    // to each of them); the lists follow the same numbering.
    {
        std::vector<int> order;
        for (int pass = 0; pass < 2; pass++)
            for (int e = 0; e < p.ne; e++)
                if ((reg.gptr[e + 1] > reg.gptr[e]) == (pass == 0)) order.push_back(e);
        p.ngram = 0;
        for (int e = 0; e < p.ne; e++) p.ngram += reg.gptr[e + 1] > reg.gptr[e];
        std::vector<int> ei2, ej2, gptr2(1, 0), grow2;
        std::vector<double> gval2;
        for (int e : order) {
            ei2.push_back(ei[e]); ej2.push_back(ej[e]);
            for (int k = reg.gptr[e]; k < reg.gptr[e + 1]; k++) { grow2.push_back(reg.grow[k]); gval2.push_back(reg.gval[k]); }
            gptr2.push_back((int)grow2.size());
        }
        ei.swap(ei2); ej.swap(ej2); reg.gptr.swap(gptr2); reg.grow.swap(grow2); reg.gval.swap(gval2);
        if (p.ngram > 128) p.static_rows = 0; // (the shipped kernels take 64, hmpc_pick_kernel; kernels compiled for a shape two trips of 64)
    }

    auto vec = [](const double *a, size_t n) { return std::vector<double>(a, a + n); };
    int rc = HMPC_OK;
    do {
        if ((rc = upload_stage(h, reg, p.reg))) break;
        if ((rc = upload(h, reg.C, &p.Creg))) break;
        {
            // dense / singleton split of the stage rows (generic kernel)
            std::vector<int> drow, rinfo(p.mreg, 0), sptr(nz + 1, 0), srow;
            std::vector<double> sval(p.mreg, 0.0);
            for (int r = 0; r < p.mreg; r++) {
                int cnt = 0, col = 0;
                for (int j = 0; j < nz; j++)
                    if (reg.C[(size_t)r * nz + j] != 0.0) { cnt++; col = j; }
                if (cnt >= 2) { rinfo[r] = -((int)drow.size() + 1); drow.push_back(r); }
                else { rinfo[r] = col; sval[r] = reg.C[(size_t)r * nz + col]; }
            }
            p.nd = (int)drow.size();
            p.ndp = (p.nd + 3) / 4 * 4;
            std::vector<double> Cdn((size_t)p.ndp * nz, 0.0);
            for (int k = 0; k < p.nd; k++)
                for (int j = 0; j < nz; j++) Cdn[(size_t)k * nz + j] = reg.C[(size_t)drow[k] * nz + j];
            drow.resize(p.ndp, 0);
            for (int j = 0; j < nz; j++) {
                for (int r = 0; r < p.mreg; r++)
                    if (rinfo[r] == j && sval[r] != 0.0) srow.push_back(r);
                sptr[j + 1] = (int)srow.size();
            }
            p.ns = (int)srow.size();
            std::vector<int> nrow;
            for (int r = 0; r < p.mreg; r++)
                if (rinfo[r] >= 0) nrow.push_back(r);
            p.nd_magic = p.nd > 0 ? (unsigned)((0x100000000ULL + p.nd - 1) / p.nd) : 0u;
            p.nn_magic = !nrow.empty() ? (unsigned)((0x100000000ULL + nrow.size() - 1) / nrow.size()) : 0u;
            if (nrow.empty()) nrow.push_back(0);
            if ((rc = upload(h, nrow, &p.nrow))) break;
            if (Cdn.empty()) Cdn.push_back(0.0);
            if (drow.empty()) drow.push_back(0);
            if (srow.empty()) srow.push_back(0);
            if ((rc = upload(h, Cdn, &p.Cdn))) break;
            if ((rc = upload(h, drow, &p.drow))) break;
            if ((rc = upload(h, rinfo, &p.rinfo))) break;
            if ((rc = upload(h, sval, &p.sval))) break;
            if ((rc = upload(h, sptr, &p.sptr))) break;
            if ((rc = upload(h, srow, &p.srow))) break;
        }
        if ((rc = upload(h, ccv, &p.ccv))) break;
        if ((rc = upload(h, cci, &p.cci))) break;
        if ((rc = upload(h, vec(q->F, (size_t)q->nc * nx), &p.F_raw))) break;
        if ((rc = upload(h, vec(q->G, (size_t)q->nc * nu), &p.G_raw))) break;
        if ((rc = upload(h, vec(q->h, (size_t)q->nc), &p.h_raw))) break;
        if ((rc = upload(h, vec(q->h_Tm1, (size_t)q->ncT), &p.hT_raw))) break;
        if ((rc = upload(h, Ct, &p.Ct))) break;
        if ((rc = upload(h, ht, &p.ht))) break;
        if ((rc = upload(h, sct, &p.sct))) break;
        if ((rc = upload(h, vec(q->A, (size_t)nx * nx), &p.A))) break;
        if ((rc = upload(h, vec(q->B, (size_t)nx * nu), &p.B))) break;
        if ((rc = upload(h, P, &p.P))) break;
        if ((rc = upload(h, PT, &p.PT))) break;
        if ((rc = upload(h, vec(q->Q, (size_t)q->nq * nx), &p.Q))) break;
        if ((rc = upload(h, vec(q->R, (size_t)q->nr * nu), &p.R))) break;
        if ((rc = upload(h, vec(q->Q_T, (size_t)q->nqT * nx), &p.QT))) break;
        if ((rc = upload(h, ei, &p.ei))) break;
        if ((rc = upload(h, ej, &p.ej))) break;
    } while (0);
    if (rc) { hmpc_destroy(h); return rc; }

    // launch geometry: one 64-lane workgroup per node in flight, as many per CU as LDS admits
    int cus = 0, lds_max = 0;
    if (!dry) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    }
    const size_t lds_cu = 160 * 1024;
    if (p.M >= 65536 || p.mreg >= 65536) { hmpc_destroy(h); return fail(HMPC_ETOOBIG, "more than 65535 constraint rows per node"); }
    // generic kernel on the matrix cores (nz >= 16): the dense stage rows go to LDS if they fit beside everything else
    p.split_lds = 0;
    p.ring = 1;
    if (p.nz >= 16) {
        const bool big = hmpc_lds_bytes(p, 0, 0) > lds_cu || getenv("HMPC_FORCE_BIG");
        p.split_lds = 1;
        if (hmpc_lds_bytes(p, 0, big ? 1 : 0) > lds_cu) p.split_lds = 0;
    }
    // streaming form: as many stages per chunk of staged multipliers as LDS has room for
    for (int r = 2; r >= 1; r--) { // (two stages per chunk hide the slab latency: the barrier of a chunk costs 0.5 % of a solve)
        p.ring = r;
        if (hmpc_lds_bytes(p, 0, 1) <= lds_cu) break;
    }
    if (const char *e = getenv("HMPC_RING")) {
        const int r = atoi(e);
        if (r >= 1 && r <= 2) p.ring = r;
    }
    // shapes without a built-in instantiation: the register kernel is compiled now (or found in the cache), hmpc_jit.h
    hmpc_kernel_choice jit[3] = {};
    // one kernel per number of waves per node; each has its own LDS carve and resident-node count
    const char *env = getenv("HMPC_BLOCKS_PER_CU");
    for (int pass = 0; pass < 2; pass++) {
        // first the kernels compiled with this problem's sizes (hmpc_jit.h); without them (HMPC_JIT_SIZED=0 / HMPC_JIT=0, no
        // compiler at run time, a compilation that fails) the shipped kernels: the built-in register kernels of the two
        // cart-pole shapes, the run-time-sized kernel for every other system
        const bool sized = pass == 0 && hmpc_sized_enabled();
        if (pass == 0 && !sized) continue;
        for (int c = 0; c < 3; c++) { jit[c] = hmpc_kernel_choice{}; h->cfg[c] = hmpc_cfg{}; }
        h->jit_kernels = 0; // (kernels of a first pass that did not complete are not in use)
        if (sized) hmpc_jit_register_shapes(p, jit, lds_cu);
        for (int c = 0; c < 3; c++) {
            hmpc_cfg &cf = h->cfg[c];
            cf.k = hmpc_pick_kernel(p, 1 << c, jit);
            cf.lds = hmpc_lds_bytes(p, cf.k.kc, cf.k.big);
            if (cf.lds > lds_cu || (lds_max > 0 && cf.lds > (size_t)lds_max)) {
                char msg[256];
                snprintf(msg, sizeof msg, "problem needs %zu bytes of LDS per node, more than one CU has (%d)", cf.lds, lds_max > 0 ? lds_max : (int)lds_cu);
                hmpc_destroy(h);
                return fail(HMPC_ETOOBIG, msg);
            }
        }
        if (!sized || hmpc_jit_prepare_sized(p, h->cfg, h->jit_libs, h->jit_kernels, built)) break;
    }
    if (dry) { delete h; return HMPC_OK; }
    for (int c = 0; c < 3; c++) {
        hmpc_cfg &cf = h->cfg[c];
        auto reserve = [](const hmpc_cfg &f) {
            return hipFuncSetAttribute((const void *)f.k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds) == hipSuccess &&
                   hipFuncSetAttribute((const void *)f.k.fn_warm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds) == hipSuccess;
        };
        if (!reserve(cf)) {
            // a kernel from the cache that this device does not take (a stale or foreign object: another architecture, another
            // runtime): the shipped kernel of the wave count serves instead, as after a failed first-use check
            const hmpc_kernel_choice ship = hmpc_pick_kernel(p, 1 << c, nullptr);
            bool ok = false;
            if (ship.fn != cf.k.fn) {
                (void)hipGetLastError();
                if (getenv("HMPC_JIT_VERBOSE")) fprintf(stderr, "hmpc: the kernel compiled for this problem (%d waves per node) cannot be set up on this device: the shipped kernel serves\n", cf.k.waves);
                cf.k = ship;
                cf.lds = hmpc_lds_bytes(p, ship.kc, ship.big);
                cf.sized = 0;
                cf.ilp = 0;
                h->jit_rejected++;
                ok = cf.lds <= lds_cu && (lds_max <= 0 || cf.lds <= (size_t)lds_max) && reserve(cf);
            }
            if (!ok) {
                hmpc_destroy(h);
                return fail(HMPC_EDEVICE, "cannot reserve dynamic LDS for the kernel");
            }
        }
        int per_cu = (int)(lds_cu / cf.lds);
        if (per_cu > 8) per_cu = 8;
        if (env && atoi(env) > 0) per_cu = atoi(env);
        cf.max_grid = (cus > 0 ? cus : 256) * per_cu;
        if (cf.max_grid > h->max_grid) h->max_grid = cf.max_grid;
    }
    h->lds = h->cfg[0].lds;
    // kernels compiled at hmpc_create are checked against the shipped kernel of the same wave count at their first launch
    bool ref_big = false;
    {
        const char *e = getenv("HMPC_JIT_SELFCHECK");
        const bool on = !(e && atoi(e) == 0);
        for (int c = 0; c < 3 && on; c++) {
            hmpc_cfg &cf = h->cfg[c];
            const hmpc_kernel_choice ref = hmpc_pick_kernel(p, 1 << c, nullptr);
            if (ref.fn == cf.k.fn) continue;                            // (a shipped kernel serves: nothing was compiled)
            const size_t lds = hmpc_lds_bytes(p, ref.kc, ref.big);
            if (lds > lds_cu || (lds_max > 0 && lds > (size_t)lds_max)) continue;
            if (hipFuncSetAttribute((const void *)ref.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) continue;
            if (ref.fn_warm && hipFuncSetAttribute((const void *)ref.fn_warm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError(); // (its list mode is the second opinion's launch: without it that net is off for this wave count)
                continue;
            }
            cf.ref = ref;
            cf.ref_lds = lds;
            int per_cu = (int)(lds_cu / lds);
            if (per_cu > 8) per_cu = 8;
            cf.ref_grid = (cus > 0 ? cus : 256) * per_cu;
            if (cf.ref_grid > h->max_grid) h->max_grid = cf.ref_grid;   // (the workspaces below must hold its launches too)
            ref_big = ref_big || ref.big;
        }
        // ((obj, dual_obj) x 3 runs, (status, iters) x 3 runs, the hand-down index, the records of the compiled kernel's cold run)
        // ... and the check's own nodes: initial states and fixing vectors (hmpc_check_set_kernel)
        if (hipMalloc(&h->chk, 3 * HMPC_CHECK_NODES * 2 * sizeof(double) + 3 * HMPC_CHECK_NODES * 2 * sizeof(int32_t) + HMPC_CHECK_NODES * sizeof(int32_t) +
                                   (size_t)HMPC_CHECK_NODES * (p.n_primal + p.n_dual + p.nx) * sizeof(double) + (size_t)HMPC_CHECK_NODES * p.T * p.nub + 64) != hipSuccess) {
            hmpc_destroy(h);
            return fail(HMPC_EDEVICE, "cannot allocate the check block");
        }
        // (pinned: an asynchronous copy to or from pageable memory -- the stack arrays this check used until round 5 -- has the
        // runtime register the pages for its duration, and that bookkeeping did not survive eight host threads checking their
        // handles at once: heap corruption inside the runtime, one crash in ~30 calls of fleet.closed_loop_parallel with 8 fleets,
        // none in 180 with the check off; profiles/r05_fleet_trace.txt)
        if (hipHostMalloc(&h->h_chk, 3 * HMPC_CHECK_NODES * 2 * sizeof(double) + 4 * HMPC_CHECK_NODES * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
            h->h_chk = nullptr;
            hmpc_destroy(h);
            return fail(HMPC_EDEVICE, "cannot allocate the check block's host mirror");
        }
        // second opinion of hmpc_solve_batch_device: its counts travel to two pinned words behind an event
        bool any_ref = false;
        for (int c = 0; c < 3; c++) any_ref = any_ref || h->cfg[c].ref.fn != nullptr;
        if (any_ref && on) {
            if (hipHostMalloc((void **)&h->h_hard, 2 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&h->hard_done, hipEventDisableTiming) != hipSuccess) {
                hmpc_destroy(h);
                return fail(HMPC_EDEVICE, "cannot allocate the second-opinion block");
            }
        }
    }
    p.fac_ws = nullptr;
    p.fac_stride = 0;
    if (h->cfg[0].k.big || h->cfg[1].k.big || h->cfg[2].k.big || ref_big) {
        p.fac_stride = p.T * (p.nx * p.nu + p.nu * (p.nu - 1) / 2) + (p.T + 1) * (p.nx * (p.nx + 1) / 2);
        if (hipMalloc((void **)&p.fac_ws, (size_t)h->max_grid * p.fac_stride * sizeof(double)) != hipSuccess) {
            hmpc_destroy(h);
            return fail(HMPC_EDEVICE, "cannot allocate the factor workspace");
        }
        h->allocs.push_back(p.fac_ws);
    }
    if (hipMalloc((void **)&p.work_counter, 2 * sizeof(int)) != hipSuccess) {
        hmpc_destroy(h);
        return fail(HMPC_EDEVICE, "cannot allocate the work counter");
    }
    h->allocs.push_back(p.work_counter);
    (void)hipMemset(p.work_counter, 0, 2 * sizeof(int));
    p.check_flag = (unsigned *)(p.work_counter + 1);
    if (hipMalloc((void **)&h->rows_ws, (size_t)h->max_grid * 4 * p.Mpad * sizeof(double)) != hipSuccess) {
        hmpc_destroy(h);
        return fail(HMPC_EDEVICE, "cannot allocate the row workspace");
    }
    if (getenv("HMPC_TRACE")) {
        (void)hipMalloc((void **)&h->trace, (2 * 64 * 8 + 32) * sizeof(double));
        (void)hipMemset(h->trace, 0, (2 * 64 * 8 + 32) * sizeof(double));
    }
    *out = h;
    return HMPC_OK;
}

extern "C" int hmpc_create(const hmpc_problem *q, const hmpc_options *opt, hmpc_handle **out) {
    hmpc_install_backtrace(); return create_impl(q, opt, out, nullptr); }

// What hmpc_create would compile for this problem -- the register kernels of its shape, or the run-time-sized kernel with
// its sizes (hmpc_jit.h) --, compiled into the cache WITHOUT a GPU: packaging, or warming the cache of a machine without a
// compiler from one that has it.  paths (may be NULL): the shared objects, newline separated.
extern "C" int hmpc_jit_build_problem(const hmpc_problem *q, const hmpc_options *opt, char *paths, int32_t paths_len)
{
    std::vector<std::string> built;
    const int rc = create_impl(q, opt, nullptr, &built);
    if (rc != HMPC_OK) return rc;
    if (paths && paths_len > 0) {
        std::string all;
        for (const std::string &b : built) all += b + "\n";
        snprintf(paths, (size_t)paths_len, "%s", all.c_str());
    }
    return HMPC_OK;
}

extern "C" int hmpc_destroy(hmpc_handle *h)
{
    if (!h) return HMPC_OK;
    if (h->dry) { delete h; return HMPC_OK; }
    (void)hipSetDevice(h->device);
    // (work this handle issued on a caller's stream may still be in flight -- the counts of a second opinion travel to pinned
    // memory behind an event nobody has waited for: everything on the device ends before anything is freed)
    (void)hipDeviceSynchronize();
    for (void *d : h->allocs) (void)hipFree(d);
    if (h->rows_ws) (void)hipFree(h->rows_ws);
    if (h->order) (void)hipFree(h->order);
    if (h->pend) (void)hipFree(h->pend);
    if (h->d_shift) (void)hipFree(h->d_shift);
    if (h->shift_tv) (void)hipFree(h->shift_tv);
    if (h->chk) (void)hipFree(h->chk);
    if (h->h_chk) (void)hipHostFree(h->h_chk);
    if (h->hard) (void)hipFree(h->hard);
    if (h->h_hard) (void)hipHostFree(h->h_hard);
    if (h->hard_done) (void)hipEventDestroy(h->hard_done);
    if (h->trace) (void)hipFree(h->trace);
    if (h->d_x0) (void)hipFree(h->d_x0);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    delete h;
    return HMPC_OK;
}

// Which kind of kernel serves this problem, per waves per node (1, 2, 4): 0 run-time-sized, 1 its streaming form,
// 2 built-in register kernel, 4 / 5 / 6 the run-time-sized kernel / its streaming form / the register kernel compiled with
// this problem's sizes at hmpc_create (3, the register kernel compiled per SHAPE of round 4, no longer exists).
extern "C" int hmpc_kernel_info(const hmpc_handle *h, int32_t *kind3)
{
    if (!h || !kind3) return fail(HMPC_EINVAL, "null argument");
    for (int c = 0; c < 3; c++) {
        const hmpc_kernel_choice &k = h->cfg[c].k;
        kind3[c] = k.kc > 0 ? (h->cfg[c].sized ? 6 : 2) : (k.big ? 1 : 0) + (h->cfg[c].sized ? 4 : 0);
    }
    return HMPC_OK;
}

// Which compiled kernels of this handle (1 / 2 / 4 waves per node) were built with the compiler's ILP schedule -- binaries listed in the
// cache's VALIDATED manifest (csrc/hmpc_jit.h) --; 0: the default schedule, or a shipped kernel.
extern "C" int hmpc_kernel_recipe(const hmpc_handle *h, int32_t *ilp3)
{
    if (!h || !ilp3) return fail(HMPC_EINVAL, "null argument");
    for (int c = 0; c < 3; c++) ilp3[c] = h->cfg[c].sized ? h->cfg[c].ilp : 0;
    return HMPC_OK;
}

// Compiled kernels of this handle: how many were dropped by the first-use check or the second opinion, in how many solve calls
// the shipped kernel was asked for a second opinion, and on how many of those batches it ended like the compiled one.
extern "C" int hmpc_jit_stats(const hmpc_handle *h, int32_t *dropped, int32_t *second_runs, int32_t *second_agreed)
{
    if (!h) return fail(HMPC_EINVAL, "null handle");
    if (dropped) *dropped = h->jit_rejected;
    if (second_runs) *second_runs = h->second_runs;
    if (second_agreed) *second_agreed = h->cfg[0].second_opinions + h->cfg[1].second_opinions + h->cfg[2].second_opinions;
    return HMPC_OK;
}

extern "C" int hmpc_record_sizes(const hmpc_handle *h, int32_t *n_primal, int32_t *n_dual)
{
    if (!h) return fail(HMPC_EINVAL, "null handle");
    if (n_primal) *n_primal = h->dp.n_primal;
    if (n_dual) *n_dual = h->dp.n_dual;
    return HMPC_OK;
}

extern "C" int hmpc_set_shift_maps(hmpc_handle *h, const hmpc_shift_maps *m)
{
    g_err.clear();
    if (!h || !m || !m->M_mu || !m->M_rho || !m->V) return fail(HMPC_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->device));
    DevProb &p = h->dp;
    // the retain rule of the shift kernel reads one binary per lane of a wavefront
    if (p.nub > 64) return fail(HMPC_EINVAL, "the node shift supports at most 64 binaries per stage");
    // a second call replaces the maps (the previous device copies are released)
    for (const double *old : {p.shift_Mmu, p.shift_Mrho, p.shift_V})
        if (old) {
            for (auto it = h->allocs.begin(); it != h->allocs.end(); ++it)
                if (*it == (void *)old) { h->allocs.erase(it); break; }
            (void)hipFree((void *)old);
        }
    p.shift_Mmu = p.shift_Mrho = p.shift_V = nullptr;
    int rc;
    auto vec = [](const double *a, size_t n) { return std::vector<double>(a, a + n); };
    if ((rc = upload(h, vec(m->M_mu, (size_t)p.nc * p.ncL), &p.shift_Mmu))) return rc;
    if ((rc = upload(h, vec(m->M_rho, (size_t)p.nq * p.nqT), &p.shift_Mrho))) return rc;
    if ((rc = upload(h, vec(m->V, (size_t)p.nub * p.nu), &p.shift_V))) return rc;
    {   // [pair of columns][row] -> (column 2k, column 2k + 1), an odd last column paired with zeros
        const size_t ncL2 = ((size_t)p.ncL + 1) / 2;
        std::vector<double> mt(2 * ncL2 * p.nc, 0.0);
        for (int r = 0; r < p.nc; r++)
            for (int k = 0; k < p.ncL; k++) mt[((size_t)(k / 2) * p.nc + r) * 2 + (k & 1)] = m->M_mu[(size_t)r * p.ncL + k];
        if (h->shift_MT2) {
            for (auto it = h->allocs.begin(); it != h->allocs.end(); ++it)
                if (*it == (void *)h->shift_MT2) { h->allocs.erase(it); break; }
            (void)hipFree(h->shift_MT2);
            h->shift_MT2 = nullptr;
        }
        const double *dev = nullptr;
        if ((rc = upload(h, mt, &dev))) return rc;
        h->shift_MT2 = (double *)dev;
    }
    return HMPC_OK;
}

static int hmpc_launch_shift(hmpc_handle *h, const ShiftArgs &a, void *stream);

extern "C" int hmpc_shift_batch_device(hmpc_handle *h, int32_t B, int32_t K, const int32_t *d_owner, const double *d_x0,
                                       const double *d_u0, const double *d_e0, const int8_t *d_fix, const double *d_lb,
                                       const double *d_dual, const double *d_dual_obj, int8_t *d_fix_out, double *d_lb_out,
                                       double *d_dual_out, double *d_dual_obj_out, uint8_t *d_flags, void *stream)
{
    g_err.clear();
    if (!h) return fail(HMPC_EINVAL, "null handle");
    if (!h->dp.shift_Mmu) return fail(HMPC_EINVAL, "hmpc_set_shift_maps has not been called");
    if (B < 0 || K < 1) return fail(HMPC_EINVAL, "bad leaf or tree count");
    if (B == 0) return HMPC_OK;
    if (!d_owner || !d_x0 || !d_u0 || !d_e0 || !d_fix || !d_lb || !d_dual || !d_dual_obj || !d_fix_out || !d_lb_out ||
        !d_dual_out || !d_dual_obj_out || !d_flags)
        return fail(HMPC_EINVAL, "null argument");
    if (d_dual == d_dual_out || d_fix == d_fix_out) return fail(HMPC_EINVAL, "the shift is not in place");
    HIPCHK(hipSetDevice(h->device));
    ShiftArgs a{B, K, d_owner, d_x0, d_u0, d_e0, d_fix, d_lb, d_dual, d_dual_obj, nullptr, d_fix_out, d_lb_out, d_dual_out, d_dual_obj_out, d_flags};
    return hmpc_launch_shift(h, a, stream);
}

// (shared with the fleet driver, which passes a row indirection)
static int hmpc_launch_shift(hmpc_handle *h, const ShiftArgs &a, void *stream)
{
    const int B = a.B;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    {   // rows staged in LDS by the memory pipeline (hmpc_shift.hip, second kernel): one workgroup per CU, a row buffer per wave
        const char *rows_env = getenv("HMPC_SHIFT_ROWS");   // (read per launch: the tests run both kernels in one process)
        const bool off = rows_env && atoi(rows_env) == 0;
        static const int cap = getenv("HMPC_SHIFT_ROW_WAVES") ? atoi(getenv("HMPC_SHIFT_ROW_WAVES")) : 0;   // (diagnostic)
        const size_t fixed = hmpc_shift_row_fixed_doubles(h->dp), per = hmpc_shift_row_wave_doubles(h->dp), room = 160 * 1024 / sizeof(double);
        int waves = fixed < room ? (int)((room - fixed) / per) : 0;
        if (waves > 16) waves = 16;
        if (cap > 0 && cap < waves) waves = cap;
        const DevProb &q = h->dp;
        if (!off && h->shift_MT2 && waves >= 4 && q.n_dual >= 2 && q.nub >= 1 && q.nc >= 1 && q.ncL >= 1 && q.nq >= 1 && q.nr >= 1 && q.nx >= 1) {
            const size_t lds = (fixed + (size_t)waves * per) * sizeof(double), need_tv = (size_t)a.K * hmpc_shift_tree_doubles(q);
            if (need_tv > h->shift_tv_cap) {   // (grows with the number of trees: the stream's earlier launches still read the old block)
                HIPCHK(hipStreamSynchronize((hipStream_t)stream));
                if (h->shift_tv) (void)hipFree(h->shift_tv);
                h->shift_tv = nullptr;
                h->shift_tv_cap = 0;
                HIPCHK(hipMalloc((void **)&h->shift_tv, need_tv * sizeof(double)));
                h->shift_tv_cap = need_tv;
            }
            int grid = cus > 0 ? cus : 256;
            const int need = (B + waves - 1) / waves;
            if (grid > need) grid = need;
            if (hipFuncSetAttribute((const void *)hmpc_shift_row_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
                hipLaunchKernelGGL(hmpc_shift_tree_kernel, dim3(a.K), dim3(64), 0, (hipStream_t)stream, h->dp, a.K, a.x0, a.u0, h->shift_tv);
                hipLaunchKernelGGL(hmpc_shift_row_kernel, dim3(grid), dim3(64 * waves), lds, (hipStream_t)stream, h->dp, a, (const double *)h->shift_tv, (const double2 *)h->shift_MT2);
                HIPCHK(hipGetLastError());
                return HMPC_OK;
            }
            (void)hipGetLastError();
        }
    }
    // persistent workgroups: enough to fill the device, each wave walks leaves with stride grid * SHIFT_WAVES
    const bool staged = hmpc_shift_lds_doubles(h->dp, true) * sizeof(double) <= 64 * 1024;
    const size_t lds = hmpc_shift_lds_doubles(h->dp, staged) * sizeof(double);
    int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) return fail(HMPC_ETOOBIG, "the shift's last-stage vectors exceed one CU's LDS");
    int grid = (cus > 0 ? cus : 256) * per_cu;
    const int need = (B + SHIFT_WAVES - 1) / SHIFT_WAVES;
    if (grid > need) grid = need;
    if (staged) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)hmpc_shift_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(hmpc_shift_kernel<true>, dim3(grid), dim3(64 * SHIFT_WAVES), lds, (hipStream_t)stream, h->dp, a);
    } else {
        hipLaunchKernelGGL(hmpc_shift_kernel<false>, dim3(grid), dim3(64 * SHIFT_WAVES), lds, (hipStream_t)stream, h->dp, a);
    }
    HIPCHK(hipGetLastError());
    return HMPC_OK;
}

extern "C" int hmpc_shift_batch(hmpc_handle *h, int32_t B, int32_t K, const int32_t *owner, const double *x0, const double *u0,
                                const double *e0, const int8_t *fix, const double *lb, const double *dual, const double *dual_obj,
                                int8_t *fix_out, double *lb_out, double *dual_out, double *dual_obj_out, uint8_t *flags)
{
    g_err.clear();
    if (!h) return fail(HMPC_EINVAL, "null handle");
    if (B < 0 || K < 1) return fail(HMPC_EINVAL, "bad leaf or tree count");
    if (B == 0) return HMPC_OK;
    if (!owner || !x0 || !u0 || !e0 || !fix || !lb || !dual || !dual_obj || !fix_out || !lb_out || !dual_out || !dual_obj_out || !flags)
        return fail(HMPC_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->device));
    const DevProb &p = h->dp;
    const size_t nf = (size_t)B * p.T * p.nub, nd = (size_t)B * p.n_dual * sizeof(double), nb = (size_t)B * sizeof(double);
    // one staging block: inputs then outputs
    struct Part { size_t bytes; const void *src; void *dst; size_t off; };
    Part parts[] = {{(size_t)B * 4, owner, nullptr, 0}, {(size_t)K * p.nx * 8, x0, nullptr, 0}, {(size_t)K * p.nu * 8, u0, nullptr, 0},
                    {(size_t)K * p.nx * 8, e0, nullptr, 0}, {nf, fix, nullptr, 0}, {nb, lb, nullptr, 0}, {nd, dual, nullptr, 0},
                    {nb, dual_obj, nullptr, 0}, {nf, nullptr, fix_out, 0}, {nb, nullptr, lb_out, 0}, {nd, nullptr, dual_out, 0},
                    {nb, nullptr, dual_obj_out, 0}, {(size_t)B, nullptr, flags, 0}};
    size_t total = 0;
    for (Part &q : parts) { q.off = total; total += (q.bytes + 255) / 256 * 256; }
    if (total > h->shift_staged) {
        if (h->d_shift) (void)hipFree(h->d_shift);
        h->d_shift = nullptr;
        h->shift_staged = 0;
        HIPCHK(hipMalloc(&h->d_shift, total));
        h->shift_staged = total;
    }
    char *base = (char *)h->d_shift;
    for (const Part &q : parts)
        if (q.src) HIPCHK(hipMemcpyAsync(base + q.off, q.src, q.bytes, hipMemcpyHostToDevice, 0));
    const int rc = hmpc_shift_batch_device(h, B, K, (const int32_t *)(base + parts[0].off), (const double *)(base + parts[1].off),
                                           (const double *)(base + parts[2].off), (const double *)(base + parts[3].off),
                                           (const int8_t *)(base + parts[4].off), (const double *)(base + parts[5].off),
                                           (const double *)(base + parts[6].off), (const double *)(base + parts[7].off),
                                           (int8_t *)(base + parts[8].off), (double *)(base + parts[9].off),
                                           (double *)(base + parts[10].off), (double *)(base + parts[11].off),
                                           (uint8_t *)(base + parts[12].off), nullptr);
    if (rc != HMPC_OK) return rc;
    for (const Part &q : parts)
        if (q.dst) HIPCHK(hipMemcpyAsync(q.dst, base + q.off, q.bytes, hipMemcpyDeviceToHost, 0));
    HIPCHK(hipStreamSynchronize(0));
    return HMPC_OK;
}

extern "C" int hmpc_launch_info(const hmpc_handle *h, int32_t *grid, int32_t *lds_bytes)
{
    if (!h) return fail(HMPC_EINVAL, "null handle");
    if (grid) *grid = h->last_grid;
    if (lds_bytes) *lds_bytes = (int32_t)h->lds;
    return HMPC_OK;
}

// FIRST-USE CHECK of a kernel compiled at hmpc_create (hmpc_cfg::ref).  The run-time compiler produces code nobody has run
// before for a problem nobody has seen -- and this kernel lives at the edge of the register file (round 5 traced the wrong
// binaries of round 4 to the compiler's stack-slot colouring of spilled scalars: DESIGN.md 4.8).  So the first launch through a
// configuration solves HMPC_CHECK_NODES nodes -- spread over ITS batch, plus the root relaxation and the deepest node of the
// batch's first initial state (hmpc_check_set_kernel) -- with the compiled kernel and with the shipped kernel of the same wave
// count, and compares statuses and objectives (1e-6 relative: the two are the same algorithm).  Agreement: the compiled
// kernel serves from then on.  Disagreement: it is dropped for this handle, loudly.  One stream synchronisation, once per
// configuration (hmpc_validate_kernels runs it ahead of time: a caller that captures its stream, or must not block in a solve call).
static int hmpc_check_compiled(hmpc_handle *h, hmpc_cfg &cf, const double *d_x0, int x0_stride, const int8_t *d_fix, int B, hipStream_t stream)
{
    cf.checked = 1;
    if (!cf.ref.fn || !h->chk || h->trace) return HMPC_OK;
    if (getenv("HMPC_JIT_SELFCHECK_SKIP_FIRST")) return HMPC_OK; // (test hook: leaves a wrong kernel to the second opinion of hmpc_solve_batch_device)
    constexpr int N = HMPC_CHECK_NODES;
    const DevProb &p = h->dp;
    const int nfix = p.T * p.nub;
    double *obj = (double *)h->chk, *dobj = obj + 3 * N;
    int32_t *st = (int32_t *)(dobj + 3 * N), *it = st + 3 * N, *idx = it + 3 * N;
    double *prim = (double *)(idx + N + (N & 1)), *dual = prim + (size_t)N * p.n_primal; // (N even: the records stay 8-byte aligned)
    double *x0c = dual + (size_t)N * p.n_dual;
    int8_t *fixc = (int8_t *)(x0c + (size_t)N * p.nx);
    hipLaunchKernelGGL(hmpc_check_set_kernel, dim3(N), dim3(256), 0, stream, d_x0, x0_stride, d_fix, B, nfix, p.nx, N, x0c, fixc);
    HIPCHK(hipGetLastError());
    const DevWarm w{nullptr, nullptr, nullptr, nullptr, 0};
    // runs 0 / 1: the shipped and the compiled kernel, cold (the compiled one keeps its records for run 2)
    for (int which = 0; which < 2; which++) {
        const hmpc_kernel_choice &k = which ? cf.k : cf.ref;
        const size_t lds = which ? cf.lds : cf.ref_lds;
        const int grid = N < (which ? cf.max_grid : cf.ref_grid) ? N : (which ? cf.max_grid : cf.ref_grid);
        const DevOut o{obj + which * N, dobj + which * N, st + which * N, it + which * N, which ? prim : nullptr, which ? dual : nullptr};
        HIPCHK(hipMemsetAsync(h->dp.work_counter, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k.fn, dim3(grid), dim3(64 * k.waves), lds, stream, h->dp, x0c, p.nx, fixc, N, o, h->rows_ws, (double *)nullptr,
                           (const int32_t *)nullptr, w);
        HIPCHK(hipGetLastError());
    }
    if (!h->h_chk) return HMPC_OK;
    double *hobj = (double *)h->h_chk, *hdob = hobj + 3 * N;   // (pinned, see hmpc_create)
    int32_t *hst = (int32_t *)(hdob + 3 * N), *hidx = hst + 3 * N;
    HIPCHK(hipMemcpyAsync(hobj, obj, 2 * N * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(hdob, dobj, 2 * N * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(hst, st, 2 * N * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    bool same = true;
    for (int b = 0; b < N; b++) {
        const int sa = hst[b], sb = hst[N + b];
        // (a node the SHIPPED kernel leaves undecided decides nothing about the compiled one)
        same = same && (sa == sb || sa >= HMPC_MAXITER);
        if (sa == HMPC_OPTIMAL && sb == HMPC_OPTIMAL) {
            const double a = hobj[b], c = hobj[N + b];
            same = same && std::fabs(a - c) <= 1e-6 * (1.0 + std::fabs(a));
        }
        // (an infeasible node's ray is normalised to a unit largest entry: its dual objective is a scalar of the WHOLE ray -- the
        // one wrong binary round 5 met that was not loud had right statuses and rays scaled by 1e-43, dual objectives off by 1e-2)
        if (sa == HMPC_INFEASIBLE && sb == HMPC_INFEASIBLE) {
            const double a = hdob[b], c = hdob[N + b];
            same = same && (std::fabs(a - c) <= 1e-4 * std::fabs(a) + 1e-9) && c == c;
        }
    }
    // run 2: the HAND-DOWN instantiation of the compiled kernel (its own binary), every optimal node handed its own record:
    // same statuses, same objectives, and a polished node's active set verifies without an interior-point iteration
    if (same && cf.k.fn_warm) {
        for (int b = 0; b < N; b++) hidx[b] = hst[N + b] == HMPC_OPTIMAL ? b : -1;
        HIPCHK(hipMemcpyAsync(idx, hidx, N * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        const DevWarm ww{prim, dual, idx, nullptr, 0};
        const DevOut o{obj + 2 * N, dobj + 2 * N, st + 2 * N, it + 2 * N, nullptr, nullptr};
        HIPCHK(hipMemsetAsync(h->dp.work_counter, 0, sizeof(int), stream));
        hipLaunchKernelGGL(cf.k.fn_warm, dim3(N < cf.max_grid ? N : cf.max_grid), dim3(64 * cf.k.waves), cf.lds, stream, h->dp, x0c, p.nx, fixc, N, o, h->rows_ws,
                           (double *)nullptr, (const int32_t *)nullptr, ww);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hobj + 2 * N, obj + 2 * N, N * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(hst + 2 * N, st + 2 * N, N * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int b = 0; b < N; b++) {
            same = same && (hst[2 * N + b] == hst[N + b] || hst[N + b] >= HMPC_MAXITER);
            if (hst[N + b] == HMPC_OPTIMAL && hst[2 * N + b] == HMPC_OPTIMAL)
                same = same && std::fabs(hobj[2 * N + b] - hobj[N + b]) <= 1e-6 * (1.0 + std::fabs(hobj[N + b]));
        }
    }
    if (getenv("HMPC_JIT_SELFCHECK_FAIL")) same = false; // (test hook: the path a disagreement takes)
    if (!same) {
        fprintf(stderr, "hmpc: the kernel compiled for this problem (%d waves per node) disagrees with the shipped kernel on the %d nodes of its first-use check: "
                        "dropped, the shipped kernel serves this handle (please report; HMPC_JIT_SIZED=0 / HMPC_JIT=0 avoid the compilation)\n", cf.k.waves, N);
        cf.k = cf.ref;
        cf.lds = cf.ref_lds;
        cf.max_grid = cf.ref_grid;
        cf.sized = 0;
        cf.ilp = 0;
        cf.checked = -1;
        h->jit_rejected++;
    }
    return HMPC_OK;
}

// The counts of the last second opinion, if they have arrived (never blocks unless `wait`): a compiled kernel that left nodes
// undecided of which the shipped kernel decides some is dropped for the handle; three batches on which both end the same way
// and the compiled kernel is trusted with its hard nodes (no further second launches for that configuration).
static void hmpc_second_opinion_review_impl(hmpc_handle *h, bool wait)
{
    if (h->hard_cfg < 0 || !h->hard_done) return;
    if (wait) { if (hipEventSynchronize(h->hard_done) != hipSuccess) return; }
    else if (hipEventQuery(h->hard_done) != hipSuccess) { (void)hipGetLastError(); return; }
    hmpc_cfg &cc = h->cfg[h->hard_cfg];
    h->hard_cfg = -1;
    const int after = h->h_hard[0], first = h->h_hard[1];
    if (first <= 0 || !cc.ref.fn || cc.k.fn == cc.ref.fn) return;
    if (after < first) {
        fprintf(stderr, "hmpc: the kernel compiled for this problem (%d waves per node) left %d nodes of a batch undecided of which the shipped kernel decides %d: "
                        "dropped, the shipped kernel serves this handle (please report; HMPC_JIT_SIZED=0 / HMPC_JIT=0 avoid the compilation)\n",
                cc.k.waves, first, first - after);
        cc.k = cc.ref;
        cc.lds = cc.ref_lds;
        cc.max_grid = cc.ref_grid;
        cc.sized = 0;
        cc.ilp = 0;
        cc.checked = -1;
        h->jit_rejected++;
    } else {
        cc.second_opinions++;
    }
}

extern "C" int hmpc_second_opinion_review(hmpc_handle *h)
{
    if (!h) return fail(HMPC_EINVAL, "null handle");
    hmpc_second_opinion_review_impl(h, true);
    return HMPC_OK;
}

// Runs the first-use checks of all three configurations now (one synchronisation each) instead of inside the first solve call
// through each: for callers that capture their stream in a graph or must not block there.  x0 / fix: any batch of the problem.
extern "C" int hmpc_validate_kernels(hmpc_handle *h, const double *d_x0, int32_t x0_stride, const int8_t *d_fix, int32_t B, void *stream)
{
    g_err.clear();
    if (!h || !d_x0 || !d_fix || B < 1) return fail(HMPC_EINVAL, "null argument or empty batch");
    HIPCHK(hipSetDevice(h->device));
    for (int c = 0; c < 3; c++)
        if (!h->cfg[c].checked) {
            const int rc = hmpc_check_compiled(h, h->cfg[c], d_x0, x0_stride, d_fix, B, (hipStream_t)stream);
            if (rc != HMPC_OK) return rc;
        }
    return HMPC_OK;
}

extern "C" int hmpc_solve_batch_device(hmpc_handle *h, const double *d_x0, int32_t x0_stride, const int8_t *d_fix,
                                       int32_t B, const hmpc_warm *d_warm, const hmpc_result *d_out, void *stream)
{
    g_err.clear();
    if (!h || !d_x0 || !d_fix || !d_out) return fail(HMPC_EINVAL, "null argument");
    if (B < 0 || (x0_stride != 0 && x0_stride < h->dp.nx)) return fail(HMPC_EINVAL, "bad batch size or x0 stride");
    if (B == 0) return HMPC_OK;
    HIPCHK(hipSetDevice(h->device));
    hmpc_second_opinion_review_impl(h, false);
    DevOut o{d_out->obj, d_out->dual_obj, d_out->status, d_out->iters, d_out->primal, d_out->dual};
    DevWarm w{nullptr, nullptr, nullptr, nullptr, 0};
    if (d_warm && d_warm->index) {
        if (!d_warm->primal || !d_warm->dual) return fail(HMPC_EINVAL, "hmpc_warm: index without record rows");
        w = DevWarm{d_warm->primal, d_warm->dual, d_warm->index, nullptr, 0};
    }
    int nw = hmpc_waves_for(B, h->cfg[0].max_grid);
    // the streaming form holds one node per CU whatever the number of waves: always spread it over all four SIMDs
    if (h->cfg[2].k.big && !getenv("HMPC_WAVES")) nw = 4;
    hmpc_cfg &cfm = h->cfg[nw == 1 ? 0 : nw == 2 ? 1 : 2];
    h->last_cfg = nw == 1 ? 0 : nw == 2 ? 1 : 2;
    if (!cfm.checked) {
        const int rc = hmpc_check_compiled(h, cfm, d_x0, x0_stride, d_fix, B, (hipStream_t)stream);
        if (rc != HMPC_OK) return rc;
    }
    const hmpc_cfg &cf = cfm;
    const hmpc_kernel_choice &k = cf.k;
    const int grid = B < cf.max_grid ? B : cf.max_grid;
    h->last_grid = grid;
    h->lds = cf.lds;
    // (every launch: also a launch with B <= grid reads the counter once per workgroup, and what it reads must be >= 0)
    HIPCHK(hipMemsetAsync(h->dp.work_counter, 0, sizeof(int), (hipStream_t)stream));
    // more nodes than resident workgroups: hand them out shallow first (hmpc_order_kernel)
    int32_t *order = nullptr;
    if (B > grid + grid / 8 && !getenv("HMPC_NO_ORDER")) {
        if (B > h->order_cap) {
            HIPCHK(hipStreamSynchronize((hipStream_t)stream));
            if (h->order) (void)hipFree(h->order);
            h->order = nullptr;
            h->order_cap = 0;
            HIPCHK(hipMalloc((void **)&h->order, (size_t)(B + B / 2) * sizeof(int32_t)));
            h->order_cap = B + B / 2;
        }
        order = h->order;
        hipLaunchKernelGGL(hmpc_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_fix, B, h->dp.T * h->dp.nub, order, w,
                           (h->dp.T + 1) * h->dp.nx + (h->dp.T - 1) * h->dp.nc + h->dp.nc, h->dp.nT, h->dp.n_dual);
    }
    // TWO-LAUNCH FORM of the lazy terminal set (DevWarm; round 4; OPT-IN: HMPC_SPLIT=1): a large cold batch of a register
    // kernel with one wave per node.  A node that needs the terminal-set rows takes 24+ iterations where the others take
    // 7 - 12, and at one wave per node it is the tail of the launch wherever it starts (6 % of the nodes of real closed-loop
    // trees: 12.4 ms per 4096 nodes against 9.0 without them).  The first launch leaves those nodes to a second one: four
    // waves per node, each from its own first record -- the active set of the masked solve plus the terminal rows it
    // violates: 232 of 238 such nodes verify that way in a few rounds (11.5 instead of 13.6 iterations per optimal node).
    // Same statuses, same vertices (1e-9).  MEASURED SLOWER, which is why it is not the default: 13.05 ms against 12.44 --
    // the six nodes whose own set does not verify (infeasible with the terminal set: they owe a certificate) run a full
    // second solve, 24 iterations = 3 ms even at four waves, ALONE on the device: the tail has moved into a launch of its
    // own.  What would pay is knowing those nodes before their first solve (profiles/r04_split.txt).
    const hmpc_cfg &c4 = h->cfg[2];
    const bool split = !w.index && nw == 1 && k.kc > 0 && c4.k.kc > 0 && h->dp.nT > 0 && h->dp.lazy && h->dp.polish && o.primal && o.dual && o.iters &&
                       o.status && getenv("HMPC_SPLIT") && !h->trace;
    if (split) {
        if (B + 1 > h->pend_cap) {
            HIPCHK(hipStreamSynchronize((hipStream_t)stream));
            if (h->pend) (void)hipFree(h->pend);
            h->pend = nullptr;
            h->pend_cap = 0;
            HIPCHK(hipMalloc((void **)&h->pend, (size_t)(B + B / 2 + 1) * sizeof(int32_t)));
            h->pend_cap = B + B / 2 + 1;
        }
        HIPCHK(hipMemsetAsync(h->pend, 0, sizeof(int32_t), (hipStream_t)stream));
        w.pend = h->pend;
    }
    hipLaunchKernelGGL(w.index ? k.fn_warm : k.fn, dim3(grid), dim3(64 * k.waves), cf.lds, (hipStream_t)stream, h->dp, d_x0, x0_stride,
                       d_fix, B, o, h->rows_ws, h->trace, (const int32_t *)order, w);
    HIPCHK(hipGetLastError());
    if (split) {
        // (how many nodes wait is known on the device only: enough workgroups for all of them, those without a node leave at once)
        const int grid2 = B < c4.max_grid ? B : c4.max_grid;
        HIPCHK(hipMemsetAsync(h->dp.work_counter, 0, sizeof(int), (hipStream_t)stream));
        DevWarm w2{o.primal, o.dual, nullptr, h->pend, 1};
        hipLaunchKernelGGL(c4.k.fn_warm, dim3(grid2), dim3(64 * c4.k.waves), c4.lds, (hipStream_t)stream, h->dp, d_x0, x0_stride, d_fix, B, o, h->rows_ws,
                           h->trace, (const int32_t *)nullptr, w2);
        HIPCHK(hipGetLastError());
    }
    // SECOND OPINION on a kernel compiled at hmpc_create (binaries of this kernel have come out wrong from the
    // compiler, always loudly -- nodes ending NUMERICAL -- and the reference never hands back an undecided node,
    // bounded_qp.py:216-228): the nodes such a kernel leaves MAXITER / NUMERICAL are listed on the device (hmpc_hard_kernel)
    // and solved again, in the same stream, by the SHIPPED kernel of the wave count (its hand-down instantiation in list mode,
    // DevWarm::second == 2; a node is handed what the first launch handed it); its records replace theirs.  Nothing is
    // synchronised: with no such node -- every call so far of every default kernel -- the two launches end at once (~10 us).
    // The counts travel to the host behind an event and are looked at by the next call (hmpc_second_opinion_review_impl).
    // Every entry that solves goes through here: hmpc_solve_batch, hmpc_fleet_solve, callers with device pointers.
    if (cfm.ref.fn && k.fn != cfm.ref.fn && cfm.ref.fn_warm && o.status && !h->trace && !split && cfm.second_opinions < 3 && h->hard_done) {
        if (B + 4 > h->hard_cap) {
            HIPCHK(hipStreamSynchronize((hipStream_t)stream));
            if (h->hard) (void)hipFree(h->hard);
            h->hard = nullptr;
            h->hard_cap = 0;
            HIPCHK(hipMalloc((void **)&h->hard, (size_t)(B + B / 2 + 4) * sizeof(int32_t)));
            h->hard_cap = B + B / 2 + 4;
        }
        HIPCHK(hipMemsetAsync(h->hard, 0, 3 * sizeof(int32_t), (hipStream_t)stream));
        hipLaunchKernelGGL(hmpc_hard_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const int32_t *)o.status, B, h->hard + 2);
        HIPCHK(hipGetLastError());
        DevProb p2 = h->dp;
        p2.work_counter = h->hard + 1;
        const int g2 = B < 64 ? B : 64; // (hard nodes are few; more of them than workgroups are handed out through the counter)
        const DevWarm w3{w.primal, w.dual, w.index, h->hard + 2, 2};
        hipLaunchKernelGGL(cfm.ref.fn_warm, dim3(g2 < cfm.ref_grid ? g2 : cfm.ref_grid), dim3(64 * cfm.ref.waves), cfm.ref_lds, (hipStream_t)stream, p2, d_x0, x0_stride,
                           d_fix, B, o, h->rows_ws, (double *)nullptr, (const int32_t *)nullptr, w3);
        HIPCHK(hipGetLastError());
        h->second_runs++;
        if (h->hard_cfg < 0) { // (the counts of an earlier call still in flight: this call's are not looked at)
            HIPCHK(hipMemcpyAsync(h->h_hard, h->hard, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
            HIPCHK(hipMemcpyAsync(h->h_hard + 1, h->hard + 2, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
            HIPCHK(hipEventRecord(h->hard_done, (hipStream_t)stream));
            h->hard_cfg = h->last_cfg;
        }
    }
    return HMPC_OK;
}

// Staging of the host-pointer entry point: ONE device block and ONE pinned host block, inputs first, then the outputs in
// the order obj | dual_obj | status | iters | primal | dual -- one copy up, one copy down per call (round 1: two pageable
// copies up, six down, each its own synchronisation: ~100 us of a 1.4 ms branch-and-bound round).
struct StageLayout {
    size_t x0, fix, widx, wprim, wdual, obj, dobj, status, iters, primal, dual, in_bytes, total;
};
// nw: parent records handed down with the batch (gathered: one row per node that has one)
static StageLayout stage_layout(const DevProb &p, size_t B, size_t nw = 0)
{
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    StageLayout L;
    L.x0 = 0;
    L.fix = up(B * p.nx * sizeof(double));
    L.widx = L.fix + up(B * (size_t)p.T * p.nub + 1);
    L.wprim = L.widx + (nw ? up(B * sizeof(int32_t)) : 0);
    L.wdual = L.wprim + up(nw * (size_t)p.n_primal * sizeof(double));
    L.in_bytes = L.wdual + up(nw * (size_t)p.n_dual * sizeof(double));
    L.obj = L.in_bytes;
    L.dobj = L.obj + up(B * sizeof(double));
    L.status = L.dobj + up(B * sizeof(double));
    L.iters = L.status + up(B * sizeof(int32_t));
    L.primal = L.iters + up(B * sizeof(int32_t));
    L.dual = L.primal + up(B * (size_t)p.n_primal * sizeof(double));
    L.total = L.dual + up(B * (size_t)p.n_dual * sizeof(double));
    return L;
}

static int ensure_staging(hmpc_handle *h, int B, bool with_warm)
{
    if (B <= h->staged && (!with_warm || h->staged_warm)) return HMPC_OK;
    if (h->d_x0) { (void)hipFree(h->d_x0); h->d_x0 = nullptr; }
    if (h->h_stage) { (void)hipHostFree(h->h_stage); h->h_stage = nullptr; }
    const int want = B > h->staged ? B : h->staged;
    h->staged = 0;
    const int cap = want < 64 ? 64 : want + want / 4;
    with_warm = with_warm || h->staged_warm;
    const StageLayout L = stage_layout(h->dp, (size_t)cap, with_warm ? (size_t)cap : 0);
    HIPCHK(hipMalloc(&h->d_x0, L.total));
    HIPCHK(hipHostMalloc(&h->h_stage, L.total, hipHostMallocDefault));
    h->staged = cap;
    h->staged_warm = with_warm;
    return HMPC_OK;
}

extern "C" int hmpc_solve_batch(hmpc_handle *h, const double *x0, int32_t x0_stride, const int8_t *fix, int32_t B,
                                const hmpc_warm *warm, const hmpc_result *out)
{
    g_err.clear();
    if (!h || !x0 || !fix || !out) return fail(HMPC_EINVAL, "null argument");
    if (B < 0 || (x0_stride != 0 && x0_stride < h->dp.nx)) return fail(HMPC_EINVAL, "bad batch size or x0 stride");
    if (B == 0) return HMPC_OK;
    HIPCHK(hipSetDevice(h->device));
    const DevProb &p = h->dp;
    // parent records handed down: gathered, one row per node that has one (the index is rewritten to the gathered rows)
    size_t nwarm = 0;
    if (warm && warm->index) {
        if (!warm->primal || !warm->dual) return fail(HMPC_EINVAL, "hmpc_warm: index without record rows");
        for (int b = 0; b < B; b++) {
            if (warm->index[b] >= warm->rows) return fail(HMPC_EINVAL, "hmpc_warm: index beyond the rows handed in");
            nwarm += warm->index[b] >= 0;
        }
    }
    int rc = ensure_staging(h, B, nwarm > 0);
    if (rc) return rc;
    // offsets of THIS batch (they always fit the capacity the blocks were allocated for): with the capacity's offsets
    // a small branch-and-bound round after one large call dragged the whole capacity-sized primal region along
    const StageLayout L = stage_layout(p, (size_t)B, nwarm);
    char *hs = (char *)h->h_stage, *ds = (char *)h->d_x0;
    const size_t nfix = (size_t)p.T * p.nub;
    if (x0_stride == 0) std::memcpy(hs + L.x0, x0, p.nx * sizeof(double));
    else
        for (int b = 0; b < B; b++) std::memcpy(hs + L.x0 + (size_t)b * p.nx * sizeof(double), x0 + (size_t)b * x0_stride, p.nx * sizeof(double));
    std::memcpy(hs + L.fix, fix, (size_t)B * nfix);
    hmpc_warm dw{nullptr, nullptr, nullptr, 0};
    if (nwarm) {
        int32_t *idx = (int32_t *)(hs + L.widx);
        size_t q = 0;
        for (int b = 0; b < B; b++) {
            const int32_t r = warm->index[b];
            idx[b] = r >= 0 ? (int32_t)q : -1;
            if (r < 0) continue;
            std::memcpy(hs + L.wprim + q * p.n_primal * sizeof(double), warm->primal + (size_t)r * p.n_primal, p.n_primal * sizeof(double));
            std::memcpy(hs + L.wdual + q * p.n_dual * sizeof(double), warm->dual + (size_t)r * p.n_dual, p.n_dual * sizeof(double));
            q++;
        }
        dw = hmpc_warm{(const double *)(ds + L.wprim), (const double *)(ds + L.wdual), (const int32_t *)(ds + L.widx), (int32_t)nwarm};
    }
    HIPCHK(hipMemcpyAsync(ds, hs, L.in_bytes, hipMemcpyHostToDevice, nullptr));
    hmpc_result d{(double *)(ds + L.obj), (double *)(ds + L.dobj), (int32_t *)(ds + L.status), (int32_t *)(ds + L.iters),
                  out->primal ? (double *)(ds + L.primal) : nullptr, out->dual ? (double *)(ds + L.dual) : nullptr};
    rc = hmpc_solve_batch_device(h, (const double *)(ds + L.x0), x0_stride == 0 ? 0 : p.nx, (const int8_t *)(ds + L.fix), B,
                                 nwarm ? &dw : nullptr, &d, nullptr);
    if (rc) return rc;
    // small outputs in one copy through the pinned block; large primal / dual blocks straight into the caller's arrays
    // (a pageable copy is pipelined by the runtime, a detour through the staging block would not be)
    const size_t pbytes = (size_t)B * p.n_primal * sizeof(double), dbytes = (size_t)B * p.n_dual * sizeof(double);
    const bool big = pbytes + dbytes > (size_t)4 << 20;
    const size_t small_end = big ? L.primal : (out->dual ? L.dual + dbytes : out->primal ? L.primal + pbytes : L.primal);
    HIPCHK(hipMemcpyAsync(hs + L.obj, ds + L.obj, small_end - L.obj, hipMemcpyDeviceToHost, nullptr));
    if (big) {
        if (out->primal) HIPCHK(hipMemcpyAsync(out->primal, ds + L.primal, pbytes, hipMemcpyDeviceToHost, nullptr));
        if (out->dual) HIPCHK(hipMemcpyAsync(out->dual, ds + L.dual, dbytes, hipMemcpyDeviceToHost, nullptr));
    }
    HIPCHK(hipStreamSynchronize(nullptr));
#ifdef HMPC_CHECK
    {
        unsigned flag = 0;
        HIPCHK(hipMemcpy(&flag, h->dp.check_flag, sizeof flag, hipMemcpyDeviceToHost));
        if (flag) {
            char msg[128];
            snprintf(msg, sizeof msg, "HMPC_CHECK: in-kernel checks failed, flag bits 0x%x", flag);
            (void)hipMemset(h->dp.check_flag, 0, sizeof flag);
            return fail(HMPC_EDEVICE, msg);
        }
    }
#endif
    hmpc_second_opinion_review_impl(h, false); // (the stream is idle: the counts of this call's second opinion have arrived)
    if (out->obj) std::memcpy(out->obj, hs + L.obj, (size_t)B * sizeof(double));
    if (out->dual_obj) std::memcpy(out->dual_obj, hs + L.dobj, (size_t)B * sizeof(double));
    if (out->status) std::memcpy(out->status, hs + L.status, (size_t)B * sizeof(int32_t));
    if (out->iters) std::memcpy(out->iters, hs + L.iters, (size_t)B * sizeof(int32_t));
    if (!big) {
        if (out->primal) std::memcpy(out->primal, hs + L.primal, pbytes);
        if (out->dual) std::memcpy(out->dual, hs + L.dual, dbytes);
    }
    if (h->trace) {
        std::vector<double> tr(2 * 64 * 8 + 32);
        (void)hipMemcpy(tr.data(), h->trace, tr.size() * sizeof(double), hipMemcpyDeviceToHost);
        for (int ph = 0; ph < 2; ph++)
            for (int it = 0; it < 64; it++) {
                const double *t = &tr[(ph * 64 + it) * 8];
                if (t[0] == 0.0) break;
                fprintf(stderr, "hip ph %d it %3d tau %.3e kap %.3e mu %.3e rp %.3e rd %.3e gap %.3e eta %.3e cert %.3e\n", ph, it,
                        t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
            }
        const char *names[8] = {"residuals", "rows->D/e", "factor", "kkt_solve", "rhs aff/corr", "refine residual", "combine/step/update", "loop top"};
        for (int ph = 0; ph < 2; ph++) {
            double tot = 0;
            for (int k = 0; k < 8; k++) tot += tr[2 * 64 * 8 + ph * 8 + k];
            if (tot > 0)
                for (int k = 0; k < 8; k++)
                    fprintf(stderr, "hip stamps ph %d %-22s %12.0f cycles %5.1f%%\n", ph, names[k], tr[2 * 64 * 8 + ph * 8 + k], 100 * tr[2 * 64 * 8 + ph * 8 + k] / tot);
        }
        // (the generic kernel's solve stamps its phases as: 6 g = C'e, 7 both sweeps, 8 lam, 9 dz, 10 nu of the fixed binaries)
        const char *fn[16] = {"F gram+PA", "F assemble col", "F prescribe", "F eliminate", "F writeback+sync", "F count", "S g=C'e", "S backward", "S Minv*mu", "S forward", "S lam,dz,dnuf",
                              "W prepass", "W first chunk", "W wave 0 recursion", "W chunk barrier", "S count"};
        for (int k = 0; k < 16; k++)
            if (tr[2 * 64 * 8 + 16 + k] > 0) fprintf(stderr, "hip fine   %-22s %12.0f cycles\n", fn[k], tr[2 * 64 * 8 + 16 + k]);
        (void)hipMemset(h->trace, 0, tr.size() * sizeof(double));
    }
    return HMPC_OK;
}

#include "hmpc_fleet.hip" // closed loops in lockstep (same translation unit: uses the launchers above)
#include "hmpc_comm.hip"  // incumbent all-reduce over RCCL
#include "hmpc_lp.hip"    // batched dense LPs of the offline terminal ingredients
