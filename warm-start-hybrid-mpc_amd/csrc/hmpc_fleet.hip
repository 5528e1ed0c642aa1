// hmpc_fleet.hip -- K closed loops advanced in lockstep behind the C ABI (include/hmpc.h, hmpc_fleet_*).
//
// What it replaces: the loop body of the reference's closed-loop study (notebooks/cart_pole_with_walls/
// statistical_analysis.py:93-196; plot_trajectory.py:19-45) -- per MPC step one warm-started branch and bound
// (warm_start_hmpc/branch_and_bound.py:408-499 with the brancher of controller.py:395-429) and one construction
// of the next warm start (controller.py:431-564) -- for K independent loops at once.
//
// Why here and not in Python: at a few hundred trees the per-node bookkeeping of an interpreted driver costs more
// than the kernels (round 1: 2 k steps/s against 500 k QP/s of kernel capacity).  The split is
//   host (this file): tree topology and bounds -- identifiers, lower bounds, which dual row a node carries; candidate
//     selection, prune / incumbent / branch; a few kilobytes per tree;
//   device: every multiplier.  A solved node's dual row is written by the QP kernel straight into a row pool and never
//     leaves HBM; children reference their parent's row by index; the node shift (hmpc_shift.hip) reads the leaves'
//     rows through that index and writes the next step's pool.  Per round the host uploads the candidates'
//     identifiers and initial states (112 B per node) and downloads objective, status and the multipliers of the
//     binaries' bounds (1.3 KB per node) through pinned staging, on one stream.
// Semantics per tree are those of warm_start_hmpc_amd/batched.py (feedforward_many / construct_warm_start_many),
// against which tests/test_fleet.py checks it step by step.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "hmpc_tree.h" // FleetResult, FleetTree and the tree bookkeeping (host only, testable under sanitizers)

struct hmpc_fleet {
    hmpc_handle *h = nullptr;
    int K = 0;
    std::vector<FleetTree> trees;
    hipStream_t stream = nullptr;
    double *pool[2] = {nullptr, nullptr}, *dobj[2] = {nullptr, nullptr};
    double *ppool = nullptr; // primal rows of the current step's solved nodes (same row index as pool[cur])
    size_t cap_rows = 0, used = 0;
    int cur = 0;
    // per-round device buffers and their pinned host mirrors
    size_t cap_b = 0;
    int8_t *d_fix = nullptr, *h_fix = nullptr, *d_fix_out = nullptr;
    double *d_x0 = nullptr, *h_x0 = nullptr, *d_obj = nullptr, *h_obj = nullptr, *d_primal = nullptr, *h_nu = nullptr;
    int32_t *d_status = nullptr, *h_status = nullptr, *d_iters = nullptr, *h_iters = nullptr;
    int32_t *d_owner = nullptr, *h_owner = nullptr, *d_src = nullptr, *h_src = nullptr, *d_widx = nullptr, *h_widx = nullptr;
    int handdown = 1; // parent -> child hand-down of active sets (hmpc_fleet_options)
    double *d_lb = nullptr, *h_lb = nullptr, *d_lb_out = nullptr;
    uint8_t *d_flags = nullptr, *h_flags = nullptr;
    double *d_kx0 = nullptr, *d_ku0 = nullptr, *d_ke0 = nullptr, *h_k = nullptr; // K x nx, K x nu, K x nx (pinned: 3 blocks)
    double *h_bits = nullptr;                                                  // dive prediction: primal rows of a round's nodes (pinned)
    size_t cap_bits = 0;
    double *h_prow = nullptr, *d_prow = nullptr;                               // K primal rows: pinned / device (incumbents of a step)
    int32_t *h_inc = nullptr, *d_inc = nullptr;                                // K: pool row of each loop's incumbent
    long long rounds = 0, launched = 0, handed = 0;
    long long uncertified = 0, resting = 0; // nodes pruned without a certificate; searches whose optimum rests on such a prune (hmpc_fleet_uncertified)
    double t_select = 0, t_stage = 0, t_device = 0, t_consume = 0, t_shift = 0; // host wall time by phase (hmpc_fleet_timing)
    bool broken = false; // a call failed midway: the trees are half updated until hmpc_fleet_reset(f, -1)
};

// Rows `rows[k]` of a pool into a dense block (the incumbents' primal rows at the end of a step: one launch and one copy
// instead of one synchronous copy per new incumbent -- 1024 loops x ~17 us were a quarter of a step's wall time).
__global__ void hmpc_gather_rows(const double *__restrict__ pool, const int32_t *__restrict__ rows, int width, double *__restrict__ out)
{
    const int k = blockIdx.x, r = rows[k];
    if (r < 0) return;
    for (int j = threadIdx.x; j < width; j += blockDim.x) out[(size_t)k * width + j] = pool[(size_t)r * width + j];
}

namespace {

// An error in the middle of a step leaves some trees advanced and others not: the fleet refuses further steps until the
// caller has reset it (hmpc_fleet_reset(f, -1)).
int fleet_fail(hmpc_fleet *f, int code, const std::string &msg)
{
    f->broken = true;
    return fail(code, msg + " -- the fleet must be reset (hmpc_fleet_reset(f, -1)) before it is used again");
}

template <class T> int dev_alloc(T **p, size_t n) { return hipMalloc((void **)p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
template <class T> int pin_alloc(T **p, size_t n) { return hipHostMalloc((void **)p, (n ? n : 1) * sizeof(T), hipHostMallocDefault) == hipSuccess ? 0 : -1; }

void fleet_free_round(hmpc_fleet *f)
{
    for (void *d : {(void *)f->d_fix, (void *)f->d_fix_out, (void *)f->d_x0, (void *)f->d_obj, (void *)f->d_primal, (void *)f->d_status, (void *)f->d_iters,
                    (void *)f->d_owner, (void *)f->d_src, (void *)f->d_widx, (void *)f->d_lb, (void *)f->d_lb_out, (void *)f->d_flags})
        if (d) (void)hipFree(d);
    for (void *d : {(void *)f->h_fix, (void *)f->h_x0, (void *)f->h_obj, (void *)f->h_nu, (void *)f->h_status, (void *)f->h_iters, (void *)f->h_owner,
                    (void *)f->h_src, (void *)f->h_widx, (void *)f->h_lb, (void *)f->h_flags})
        if (d) (void)hipHostFree(d);
    f->d_fix = f->h_fix = f->d_fix_out = nullptr;
    f->d_x0 = f->h_x0 = f->d_obj = f->h_obj = f->d_primal = f->h_nu = nullptr;
    f->d_status = f->h_status = f->d_iters = f->h_iters = nullptr;
    f->d_owner = f->h_owner = f->d_src = f->h_src = f->d_widx = f->h_widx = nullptr;
    f->d_lb = f->h_lb = f->d_lb_out = nullptr;
    f->d_flags = f->h_flags = nullptr;
    f->cap_b = 0;
}

int fleet_ensure_round(hmpc_fleet *f, size_t B)
{
    if (B <= f->cap_b) return HMPC_OK;
    HIPCHK(hipStreamSynchronize(f->stream));
    fleet_free_round(f);
    const DevProb &p = f->h->dp;
    const size_t cap = std::max<size_t>(2 * B, 1024), nfix = (size_t)p.T * p.nub;
    int bad = 0;
    bad |= dev_alloc(&f->d_fix, cap * nfix) | pin_alloc(&f->h_fix, cap * nfix) | dev_alloc(&f->d_fix_out, cap * nfix);
    bad |= dev_alloc(&f->d_x0, cap * p.nx) | pin_alloc(&f->h_x0, cap * p.nx);
    bad |= dev_alloc(&f->d_obj, cap) | pin_alloc(&f->h_obj, cap);
    bad |= pin_alloc(&f->h_nu, cap * 2 * nfix);
    bad |= dev_alloc(&f->d_status, cap) | pin_alloc(&f->h_status, cap) | dev_alloc(&f->d_iters, cap) | pin_alloc(&f->h_iters, cap);
    bad |= dev_alloc(&f->d_owner, cap) | pin_alloc(&f->h_owner, cap) | dev_alloc(&f->d_src, cap) | pin_alloc(&f->h_src, cap);
    bad |= dev_alloc(&f->d_widx, cap) | pin_alloc(&f->h_widx, cap);
    bad |= dev_alloc(&f->d_lb, cap) | pin_alloc(&f->h_lb, cap) | dev_alloc(&f->d_lb_out, cap);
    bad |= dev_alloc(&f->d_flags, cap) | pin_alloc(&f->h_flags, cap);
    if (bad) return fail(HMPC_EDEVICE, "fleet: cannot allocate the round buffers");
    f->cap_b = cap;
    return HMPC_OK;
}

// Room for `rows` rows in both pools; the rows in use of the current pool are kept.
int fleet_ensure_rows(hmpc_fleet *f, size_t rows)
{
    if (rows <= f->cap_rows) return HMPC_OK;
    HIPCHK(hipStreamSynchronize(f->stream));
    const DevProb &p = f->h->dp;
    const size_t cap = std::max<size_t>(rows + rows / 2, 4096);
    for (int s = 0; s < 2; s++) {
        double *np_ = nullptr, *nd = nullptr;
        if (dev_alloc(&np_, cap * p.n_dual) || dev_alloc(&nd, cap)) {
            if (np_) (void)hipFree(np_);
            if (nd) (void)hipFree(nd);
            return fail(HMPC_EDEVICE, "fleet: cannot grow the row pools");
        }
        if (s == f->cur && f->used) {
            if (hipMemcpy(np_, f->pool[s], f->used * p.n_dual * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess ||
                hipMemcpy(nd, f->dobj[s], f->used * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess) {
                (void)hipFree(np_);
                (void)hipFree(nd);
                return fail(HMPC_EDEVICE, "fleet: cannot copy the row pools");
            }
        }
        if (f->pool[s]) (void)hipFree(f->pool[s]);
        if (f->dobj[s]) (void)hipFree(f->dobj[s]);
        f->pool[s] = np_;
        f->dobj[s] = nd;
    }
    {
        double *pp = nullptr;
        if (dev_alloc(&pp, cap * p.n_primal)) return fail(HMPC_EDEVICE, "fleet: cannot grow the row pools");
        if (f->ppool && f->used && hipMemcpy(pp, f->ppool, f->used * p.n_primal * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipFree(pp);
            return fail(HMPC_EDEVICE, "fleet: cannot copy the row pools");
        }
        if (f->ppool) (void)hipFree(f->ppool);
        f->ppool = pp;
    }
    f->cap_rows = cap;
    return HMPC_OK;
}

} // namespace

extern "C" int hmpc_fleet_create(hmpc_handle *h, int32_t K, hmpc_fleet **out)
{
    g_err.clear();
    if (!h || !out || K < 1) return fail(HMPC_EINVAL, "fleet: null handle or K < 1");
    if (!h->dp.shift_Mmu) return fail(HMPC_EINVAL, "fleet: hmpc_set_shift_maps has not been called");
    HIPCHK(hipSetDevice(h->device));
    hmpc_fleet *f = new hmpc_fleet();
    f->h = h;
    f->K = K;
    const DevProb &p = h->dp;
    f->trees.resize(K);
    for (auto &t : f->trees) {
        tree_reset_cold(t, p.T * p.nub);
        t.x0.assign(p.nx, 0.0);
    }
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) { delete f; return fail(HMPC_EDEVICE, "fleet: cannot create a stream"); }
    int bad = dev_alloc(&f->d_kx0, (size_t)K * p.nx) | dev_alloc(&f->d_ku0, (size_t)K * p.nu) | dev_alloc(&f->d_ke0, (size_t)K * p.nx);
    bad |= pin_alloc(&f->h_k, (size_t)K * (2 * p.nx + p.nu)) | pin_alloc(&f->h_prow, (size_t)K * p.n_primal) | dev_alloc(&f->d_prow, (size_t)K * p.n_primal);
    bad |= pin_alloc(&f->h_inc, (size_t)K) | dev_alloc(&f->d_inc, (size_t)K);
    if (bad) { hmpc_fleet_destroy(f); return fail(HMPC_EDEVICE, "fleet: cannot allocate"); }
    *out = f;
    return HMPC_OK;
}

extern "C" int hmpc_fleet_destroy(hmpc_fleet *f)
{
    if (!f) return HMPC_OK;
    (void)hipSetDevice(f->h->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    fleet_free_round(f);
    for (int s = 0; s < 2; s++) {
        if (f->pool[s]) (void)hipFree(f->pool[s]);
        if (f->dobj[s]) (void)hipFree(f->dobj[s]);
    }
    for (void *d : {(void *)f->d_kx0, (void *)f->d_ku0, (void *)f->d_ke0, (void *)f->ppool, (void *)f->d_prow, (void *)f->d_inc})
        if (d) (void)hipFree(d);
    if (f->h_k) (void)hipHostFree(f->h_k);
    if (f->h_prow) (void)hipHostFree(f->h_prow);
    if (f->h_inc) (void)hipHostFree(f->h_inc);
    if (f->h_bits) (void)hipHostFree(f->h_bits);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
    return HMPC_OK;
}

extern "C" int hmpc_fleet_reset(hmpc_fleet *f, int32_t k)
{
    if (!f || k < -1 || k >= f->K) return fail(HMPC_EINVAL, "fleet: bad loop index");
    const int nfix = f->h->dp.T * f->h->dp.nub;
    for (int i = (k < 0 ? 0 : k); i < (k < 0 ? f->K : k + 1); i++) tree_reset_cold(f->trees[i], nfix);
    if (k < 0) f->broken = false;
    return HMPC_OK;
}

// Loop k has ended (its MIQP had no solution, or its caller has no further use for it): its tree is dropped and the loop
// takes no part in further steps -- no launches, no rows held in the pools -- until hmpc_fleet_reset makes it cold again.
extern "C" int hmpc_fleet_stop(hmpc_fleet *f, int32_t k)
{
    if (!f || k < 0 || k >= f->K) return fail(HMPC_EINVAL, "fleet: bad loop index");
    tree_reset_cold(f->trees[k], f->h->dp.T * f->h->dp.nub);
    f->trees[k].running = false;
    return HMPC_OK;
}

// Rows of the pools in use.  (For tests and reports: a fleet that is reset and solved at every step -- the cold searches
// of a closed-loop study -- must not grow with the number of steps.)
extern "C" int hmpc_fleet_rows(const hmpc_fleet *f, int64_t *used, int64_t *capacity)
{
    if (!f) return fail(HMPC_EINVAL, "fleet: null argument");
    if (used) *used = (int64_t)f->used;
    if (capacity) *capacity = (int64_t)f->cap_rows;
    return HMPC_OK;
}

// One MPC step of every running loop: branch and bound from the loop's current tree (the root for a cold loop).
// speculation = k > 0: with every candidate that has to be solved, its descendants through the next k binaries (2 + 4 +
// ... + 2^k nodes) ride in the same launch; their results wait in a per-tree cache and are consumed -- unchanged -- if and
// when the search selects them, so incumbent, leaves and solve counts are those of the search without speculation and only
// the number of launches drops (a warm-started step: from five to one or two).  Worth it for few loops (latency), a waste
// for many (throughput).
extern "C" int hmpc_fleet_solve(hmpc_fleet *f, const double *x0, int32_t width, int32_t speculation, double tol, double *cost,
                                double *u0, double *x1, int32_t *solves, int32_t *n_leaves)
{
    g_err.clear();
    if (!f || !x0) return fail(HMPC_EINVAL, "fleet: null argument");
    if (f->broken) return fail(HMPC_EINVAL, "fleet: an earlier call failed midway; reset the fleet (hmpc_fleet_reset(f, -1)) first");
    struct Guard { hmpc_fleet *f; bool ok = false; ~Guard() { if (!ok) f->broken = true; } } guard{f};
    if (width < 1) width = 1;
    // speculation < 0: DIVE PREDICTION (for few loops: it fetches the primal rows of every round).  A branch-and-bound dive
    // follows the relaxation: where a parent's relaxed binaries round to, its descendants' mostly stay.  With a picked node
    // whose parent's record is at hand, the whole predicted rest of the dive -- the node extended by the parent's rounded
    // binaries, one more at a time -- and the sibling of every step ride in the same launch: 2 (T nub - depth) nodes, linear
    // in the depth where the subtree expansion (speculation > 0) is exponential.  Results wait in the cache and are consumed
    // only if and when the search selects those nodes: incumbent, leaves and solve counts are those of the search without it.
    // A cold start is then the root, one launch with the predicted dive, and what the prediction missed.
    const bool dive = speculation < 0;
    if (speculation < 0) speculation = 0;
    hmpc_handle *h = f->h;
    HIPCHK(hipSetDevice(h->device));
    const DevProb &p = h->dp;
    const int K = f->K, nfix = p.T * p.nub, nx = p.nx, nu = p.nu;
    const int o_lb = (p.T + 1) * nx + (p.T - 1) * p.nc + p.ncL; // nu_lb then nu_ub, contiguous in the dual row
    const double inf = std::numeric_limits<double>::infinity();
    for (int k = 0; k < K; k++) tree_begin_step(f->trees[k], x0 + (size_t)k * nx, nx);
    {
        // Rows nobody references are reclaimed: when every tree is cold (no node carries a row of its own or of its parent)
        // the pools start from zero again.  Without this a fleet that is reset and solved at every step, never shifted --
        // the cold searches of fleet.closed_loop_study -- kept every row ever written: 880 k rows of ~8 KB over the
        // published sd = .01 study, 25-30 GB with the spare pool and the regrow copies (only hmpc_fleet_shift compacted).
        bool cold = true;
        for (int k = 0; k < K && cold; k++) {
            const FleetTree &t = f->trees[k];
            for (int i = 0; i < t.n && cold; i++) cold = t.row[i] < 0 && t.wrow[i] < 0;
        }
        if (cold) f->used = 0;
    }
    std::vector<std::vector<int>> picks(K);
    std::vector<int> order;
    struct Launch { int k, depth; };
    std::vector<Launch> launch;
    std::vector<int8_t> level, next; // identifiers of one level of a speculative expansion
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        // candidates of every tree: alive, bound below the incumbent; the `width` smallest bounds, first wins ties
        double t0 = now();
        size_t npick = 0;
        for (int k = 0; k < K; k++) {
            tree_select(f->trees[k], width, tol, picks[k]);
            npick += picks[k].size();
        }
        f->t_select += now() - t0;
        if (npick == 0) break;
        t0 = now();
        // what has to be launched: picked nodes without a cached result, and their speculative descendants
        launch.clear();
        size_t B = 0;
        int rc;
        bool any_warm = false;
        for (int pass = 0; pass < 2; pass++) { // pass 0 counts, pass 1 fills the staging buffers
            if (pass == 1) {
                if (B == 0) break;
                if ((rc = fleet_ensure_round(f, B))) return rc;
                if ((rc = fleet_ensure_rows(f, f->used + B))) return rc;
            }
            size_t b = 0;
            for (int k = 0; k < K; k++) {
                FleetTree &t = f->trees[k];
                for (int i : picks[k])
                    tree_expand(t, i, nfix, speculation, dive, f->handdown != 0, level, next, [&](const int8_t *row, int depth, int32_t widx) {
                        if (pass == 1) {
                            std::memcpy(f->h_fix + b * nfix, row, nfix);
                            std::memcpy(f->h_x0 + b * nx, t.x0.data(), nx * sizeof(double));
                            f->h_widx[b] = widx;
                            any_warm |= widx >= 0;
                            launch.push_back({k, depth});
                        }
                        b++;
                    });
            }
            B = b;
        }
        f->t_stage += now() - t0;
        t0 = now();
        if (B > 0) {
            HIPCHK(hipMemcpyAsync(f->d_fix, f->h_fix, B * nfix, hipMemcpyHostToDevice, f->stream));
            HIPCHK(hipMemcpyAsync(f->d_x0, f->h_x0, B * nx * sizeof(double), hipMemcpyHostToDevice, f->stream));
            double *rows = f->pool[f->cur] + f->used * p.n_dual;
            hmpc_result r{f->d_obj, f->dobj[f->cur] + f->used, f->d_status, f->d_iters, f->ppool + f->used * p.n_primal, rows};
            // (the parents' rows lie below f->used, this launch writes from f->used on)
            hmpc_warm hw{f->ppool, f->pool[f->cur], f->d_widx, (int32_t)f->used};
            if (any_warm) HIPCHK(hipMemcpyAsync(f->d_widx, f->h_widx, B * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
            if ((rc = hmpc_solve_batch_device(h, f->d_x0, nx, f->d_fix, (int32_t)B, any_warm ? &hw : nullptr, &r, f->stream))) return rc;
            HIPCHK(hipMemcpyAsync(f->h_obj, f->d_obj, B * sizeof(double), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipMemcpyAsync(f->h_status, f->d_status, B * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipMemcpyAsync(f->h_iters, f->d_iters, B * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
            if (dive) { // the round's primal rows: the rounded binaries of its vertex nodes predict their descendants' dives
                if (B > f->cap_bits) {
                    HIPCHK(hipStreamSynchronize(f->stream));
                    if (f->h_bits) (void)hipHostFree(f->h_bits);
                    f->h_bits = nullptr;
                    f->cap_bits = 0;
                    if (pin_alloc(&f->h_bits, 2 * B * p.n_primal)) return fail(HMPC_EDEVICE, "fleet: cannot allocate the prediction buffer");
                    f->cap_bits = 2 * B;
                }
                HIPCHK(hipMemcpyAsync(f->h_bits, f->ppool + f->used * p.n_primal, B * p.n_primal * sizeof(double), hipMemcpyDeviceToHost, f->stream));
            }
            HIPCHK(hipMemcpy2DAsync(f->h_nu, 2 * nfix * sizeof(double), rows + o_lb, p.n_dual * sizeof(double), 2 * nfix * sizeof(double), B,
                                    hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipStreamSynchronize(f->stream));
            f->t_device += now() - t0;
            t0 = now();
            f->rounds++;
            f->launched += (long long)B;
            for (size_t q = 0; q < B; q++) {
                if (f->h_iters[q] & HMPC_ITERS_WEAK) {
                    // infeasible, but the ray is no proof to tolerance: it prunes this node at this step only.  With a
                    // dual objective of -inf the shift reopens the leaf whatever the model error (controller.py:555-558).
                    const double ninf = -inf;
                    HIPCHK(hipMemcpy(f->dobj[f->cur] + f->used + q, &ninf, sizeof(double), hipMemcpyHostToDevice));
                }
                const int d = launch[q].depth;
                const double *nu_ = f->h_nu + q * 2 * nfix;
                FleetResult e{f->h_obj[q], d < nfix ? nu_[d] : 0.0, d < nfix ? nu_[nfix + d] : 0.0, (int32_t)(f->used + q),
                              f->h_status[q] == HMPC_OPTIMAL && (f->h_iters[q] & HMPC_ITERS_POLISHED) != 0, f->h_status[q] > 1,
                              (f->h_iters[q] & HMPC_ITERS_UNCERTIFIED) != 0};
                f->handed += (f->h_iters[q] & HMPC_ITERS_HANDED) != 0;
                if (dive && e.vertex && d < nfix) {
                    std::vector<int8_t> bits(nfix);
                    const double *u = f->h_bits + q * p.n_primal + (size_t)(p.T + 1) * nx;
                    for (int j = 0; j < nfix; j++) bits[j] = u[(j / p.nub) * nu + p.nuc + (j % p.nub)] > 0.5 ? 1 : 0;
                    f->trees[launch[q].k].rounded.emplace(e.row, std::move(bits));
                }
                f->trees[launch[q].k].cache.emplace(tree_key(f->h_fix + q * nfix, d), e);
            }
            f->used += B;
        }
        // prune / incumbent / branch, node by node in selection order (branch_and_bound.py:476-489)
        struct Tick { double &acc; double t0; ~Tick() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; } } tick{f->t_consume, t0};
        for (int k = 0; k < K; k++) {
            const int bad = tree_consume(f->trees[k], picks[k], nfix, tol);
            if (bad == 1) return fail(HMPC_EDEVICE, "fleet: a selected node has no result");
            if (bad == 2) return fleet_fail(f, HMPC_EDEVICE, "fleet: the QP solver did not converge on a node (status MAXITER / NUMERICAL)");
        }
    }
    for (int k = 0; k < K; k++) { // prunes without a certificate (HMPC_ITERS_UNCERTIFIED): counted; said aloud where a search's result rests on one
        FleetTree &t = f->trees[k];
        if (!t.uncertified) continue;
        f->uncertified += t.uncertified;
        if (t.unc_lb < t.ub) {
            if (!f->resting)
                fprintf(stderr, "hmpc: a branch-and-bound search pruned a node on the collapse of tau alone (no infeasibility certificate, HMPC_ITERS_UNCERTIFIED) "
                                "whose bound %.6g lay below the final incumbent %.6g: the returned optimum rests on that prune (hmpc_fleet_uncertified counts further ones)\n",
                        t.unc_lb, t.ub);
            f->resting++;
        }
        t.uncertified = 0;
    }
    {   // the incumbents' primal rows: one gather launch, one copy
        bool any = false;
        for (int k = 0; k < K; k++) {
            f->h_inc[k] = (f->trees[k].running && f->trees[k].inc >= 0) ? f->trees[k].inc_row : -1;
            any |= f->h_inc[k] >= 0;
        }
        if (any) {
            HIPCHK(hipMemcpyAsync(f->d_inc, f->h_inc, (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
            hipLaunchKernelGGL(hmpc_gather_rows, dim3(K), dim3(256), 0, f->stream, (const double *)f->ppool, (const int32_t *)f->d_inc, p.n_primal, f->d_prow);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(f->h_prow, f->d_prow, (size_t)K * p.n_primal * sizeof(double), hipMemcpyDeviceToHost, f->stream));
            HIPCHK(hipStreamSynchronize(f->stream));
            for (int k = 0; k < K; k++)
                if (f->h_inc[k] >= 0) f->trees[k].primal.assign(f->h_prow + (size_t)k * p.n_primal, f->h_prow + (size_t)(k + 1) * p.n_primal);
        }
    }
    for (int k = 0; k < K; k++) {
        FleetTree &t = f->trees[k];
        t.cache.clear();
        if (cost) cost[k] = t.running ? t.ub : inf;
        if (solves) solves[k] = t.solves;
        if (n_leaves) n_leaves[k] = t.running ? tree_leaves(t) : 0;
        const bool ok = t.running && t.inc >= 0;
        for (int j = 0; j < nu && u0; j++) u0[(size_t)k * nu + j] = ok ? t.primal[(size_t)(p.T + 1) * nx + j] : NAN;
        for (int j = 0; j < nx && x1; j++) x1[(size_t)k * nx + j] = ok ? t.primal[nx + j] : NAN;
        if (t.running && t.inc < 0) t.running = false; // infeasible MIQP: the loop ends here (statistical_analysis.py:99-108)
    }
    guard.ok = true;
    return HMPC_OK;
}

// The next step's warm start of every running loop (controller.py:431-564): retain rule on the host (it only needs the
// identifiers and the applied binaries), everything that touches multipliers in one launch of the shift kernel.
extern "C" int hmpc_fleet_shift(hmpc_fleet *f, const double *e0, int32_t *cover, int32_t *reopened)
{
    g_err.clear();
    if (!f || !e0) return fail(HMPC_EINVAL, "fleet: null argument");
    if (f->broken) return fail(HMPC_EINVAL, "fleet: an earlier call failed midway; reset the fleet (hmpc_fleet_reset(f, -1)) first");
    struct Guard { hmpc_fleet *f; bool ok = false; ~Guard() { if (!ok) f->broken = true; } } guard{f};
    struct Tick { double &acc; double t0; ~Tick() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; } }
        tick{f->t_shift, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()};
    hmpc_handle *h = f->h;
    HIPCHK(hipSetDevice(h->device));
    const DevProb &p = h->dp;
    const int K = f->K, nfix = p.T * p.nub, nx = p.nx, nu = p.nu, nub = p.nub, nuc = p.nuc;
    // kept leaves of all trees
    std::vector<std::vector<int>> keep(K);
    size_t B = 0;
    for (int k = 0; k < K; k++) {
        FleetTree &t = f->trees[k];
        if (cover) cover[k] = 0;
        if (reopened) reopened[k] = 0;
        if (!t.running || t.inc < 0) continue;
        tree_retain(t, t.primal.data() + (size_t)(p.T + 1) * nx, nuc, nub, nfix, keep[k]);
        B += keep[k].size();
    }
    if (B == 0) { guard.ok = true; return HMPC_OK; }
    int rc = fleet_ensure_round(f, B);
    if (rc) return rc;
    if ((rc = fleet_ensure_rows(f, std::max(f->used, B)))) return rc;
    double *hx = f->h_k, *hu = hx + (size_t)K * nx, *he = hu + (size_t)K * nu;
    size_t b = 0;
    for (int k = 0; k < K; k++) {
        FleetTree &t = f->trees[k];
        std::memcpy(hx + (size_t)k * nx, t.x0.data(), nx * sizeof(double));
        std::memcpy(he + (size_t)k * nx, e0 + (size_t)k * nx, nx * sizeof(double));
        for (int j = 0; j < nu; j++) hu[(size_t)k * nu + j] = keep[k].empty() ? 0.0 : t.primal[(size_t)(p.T + 1) * nx + j];
        for (int i : keep[k]) {
            if (t.row[i] < 0) return fail(HMPC_EINVAL, "fleet: a leaf carries no multipliers (unsolved root?)");
            std::memcpy(f->h_fix + b * nfix, t.fix.data() + (size_t)i * nfix, nfix);
            f->h_owner[b] = k;
            f->h_src[b] = t.row[i];
            f->h_lb[b] = t.lb[i];
            b++;
        }
    }
    HIPCHK(hipMemcpyAsync(f->d_kx0, hx, (size_t)K * nx * sizeof(double), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_ku0, hu, (size_t)K * nu * sizeof(double), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_ke0, he, (size_t)K * nx * sizeof(double), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_fix, f->h_fix, B * nfix, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_owner, f->h_owner, B * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_src, f->h_src, B * sizeof(int32_t), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->d_lb, f->h_lb, B * sizeof(double), hipMemcpyHostToDevice, f->stream));
    const int nxt = f->cur ^ 1;
    ShiftArgs a{(int)B, K, f->d_owner, f->d_kx0, f->d_ku0, f->d_ke0, f->d_fix, f->d_lb, f->pool[f->cur], f->dobj[f->cur], f->d_src,
                f->d_fix_out, f->d_lb_out, f->pool[nxt], f->dobj[nxt], f->d_flags};
    if ((rc = hmpc_launch_shift(h, a, f->stream))) return rc;
    HIPCHK(hipMemcpyAsync(f->h_lb, f->d_lb_out, B * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(f->h_flags, f->d_flags, B, hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    // the shifted leaves are the next tree: identifiers move one stage towards the present
    b = 0;
    for (int k = 0; k < K; k++) {
        FleetTree &t = f->trees[k];
        if (keep[k].empty()) continue;
        const size_t n = keep[k].size();
        for (size_t j = 0; j < n; j++)
            if (!(f->h_flags[b + j] & 1)) return fail(HMPC_EDEVICE, "fleet: host and device disagree on the retain rule");
        const int reop = tree_adopt_shifted(t, keep[k], f->h_lb + b, f->h_flags + b, (int32_t)b, nub, nfix);
        b += n;
        if (cover) cover[k] = (int32_t)n;
        if (reopened) reopened[k] = reop;
    }
    f->cur = nxt;
    f->used = B;
    guard.ok = true;
    return HMPC_OK;
}

extern "C" int hmpc_fleet_uncertified(const hmpc_fleet *f, int64_t *pruned, int64_t *searches_resting_on_one)
{
    if (!f) return fail(HMPC_EINVAL, "fleet: null");
    if (pruned) *pruned = f->uncertified;
    if (searches_resting_on_one) *searches_resting_on_one = f->resting;
    return HMPC_OK;
}

extern "C" int hmpc_fleet_stats(const hmpc_fleet *f, int64_t *rounds, int64_t *launched)
{
    if (!f) return fail(HMPC_EINVAL, "fleet: null");
    if (rounds) *rounds = f->rounds;
    if (launched) *launched = f->launched;
    return HMPC_OK;
}

// Host wall time of the fleet's calls by phase since creation, seconds: candidate selection, staging of a round's nodes,
// device (copies, kernel, synchronisation), consumption of the results (prune / incumbent / branch), node shifts.
extern "C" int hmpc_fleet_timing(const hmpc_fleet *f, double *seconds5)
{
    if (!f || !seconds5) return fail(HMPC_EINVAL, "fleet: null");
    seconds5[0] = f->t_select; seconds5[1] = f->t_stage; seconds5[2] = f->t_device; seconds5[3] = f->t_consume; seconds5[4] = f->t_shift;
    return HMPC_OK;
}

extern "C" int hmpc_fleet_handdown(hmpc_fleet *f, int32_t enable, int64_t *verified)
{
    if (!f) return fail(HMPC_EINVAL, "fleet: null");
    if (enable >= 0) f->handdown = enable != 0;
    if (verified) *verified = f->handed;
    return HMPC_OK;
}
