// tree_driver.cpp -- host-only driver of the fleet's tree bookkeeping (csrc/hmpc_tree.h) for AddressSanitizer / UBSan.
//
// TEST INFRASTRUCTURE (built and run by tests/test_sanitizers.py with g++ -fsanitize=address,undefined; the GPU pool has
// no device sanitizer).  It walks the same sequence as hmpc_fleet_solve / hmpc_fleet_shift -- select, expand
// (count pass, fill pass), consume, retain, adopt -- on K trees, with the QP relaxations of every round solved by the CPU
// oracle (liboracle_qp.so, loaded at run time) in place of the kernel launch, and prints per step and tree the cost, the
// number of solves and of leaves as one JSON line; the Python test compares them with the Python branch and bound.
// The node shift itself is a device kernel and not part of this driver: the leaves a step retains are adopted with the
// bound -inf (every leaf reopened -- a valid warm start whatever the model error), which drives tree_adopt_shifted and a
// warm-started search of the next step through the same code.
//
//   tree_driver PROBLEM.bin LIBORACLE K STEPS WIDTH SPECULATION DIVE HANDDOWN
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

#include "hmpc_tree.h"

typedef int (*oracle_fn)(int, int, int, int, int, int, int, int, int, const double *, const double *, const double *, const double *, const double *,
                         const double *, const double *, const double *, const double *, const double *, const double *, const double *, int, int,
                         const int8_t *, double, double, int, int, int, int, int, double, const double *, const double *, const int32_t *, double *,
                         double *, int *, int *, double *, double *, int *);

static std::vector<double> read_block(FILE *f, size_t n)
{
    std::vector<double> v(n);
    if (n && fread(v.data(), sizeof(double), n, f) != n) { fprintf(stderr, "short problem file\n"); exit(2); }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: tree_driver PROBLEM.bin LIBORACLE K STEPS WIDTH SPECULATION DIVE HANDDOWN\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("problem file"); return 2; }
    int32_t dims[9]; // nx nu nub T nc ncL nq nr nqT
    if (fread(dims, sizeof(int32_t), 9, f) != 9) return 2;
    const int nx = dims[0], nu = dims[1], nub = dims[2], T = dims[3], nc = dims[4], ncL = dims[5], nq = dims[6], nr = dims[7], nqT = dims[8];
    const int nuc = nu - nub, nfix = T * nub;
    auto A = read_block(f, (size_t)nx * nx), Bm = read_block(f, (size_t)nx * nu), F = read_block(f, (size_t)nc * nx), G = read_block(f, (size_t)nc * nu),
         h = read_block(f, nc), FT = read_block(f, (size_t)ncL * nx), GT = read_block(f, (size_t)ncL * nu), hT = read_block(f, ncL),
         Q = read_block(f, (size_t)nq * nx), R = read_block(f, (size_t)nr * nu), QT = read_block(f, (size_t)nqT * nx);
    const int K = atoi(argv[3]), steps = atoi(argv[4]), width = atoi(argv[5]);
    int speculation = atoi(argv[6]);
    const bool dive = atoi(argv[7]) != 0, handdown = atoi(argv[8]) != 0;
    auto x0s = read_block(f, (size_t)K * nx);
    fclose(f);
    void *lib = dlopen(argv[2], RTLD_NOW);
    if (!lib) { fprintf(stderr, "cannot load %s: %s\n", argv[2], dlerror()); return 2; }
    oracle_fn solve = (oracle_fn)dlsym(lib, "oracle_solve_batch");
    if (!solve) { fprintf(stderr, "oracle_solve_batch not found\n"); return 2; }
    const int n_primal = (T + 1) * nx + T * nu, n_dual = (T + 1) * nx + (T - 1) * nc + ncL + 2 * T * nub + T * nq + nqT + T * nr;
    const int o_lb = (T + 1) * nx + (T - 1) * nc + ncL;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<FleetTree> trees(K);
    for (auto &t : trees) tree_reset_cold(t, nfix);
    std::vector<double> pool_dual, pool_primal; // the row pools of a step (host copies of what lives in HBM)
    std::vector<std::vector<int>> picks(K), keep(K);
    std::vector<int8_t> level, next, h_fix;
    std::vector<double> h_x0;
    std::vector<int32_t> h_widx;
    struct Launch { int k, depth; };
    std::vector<Launch> launch;
    printf("[");
    for (int step = 0; step < steps; step++) {
        size_t used = 0;
        pool_dual.clear();
        pool_primal.clear();
        // (rows a warm-started tree carries refer to the previous step's pool: reopened leaves are re-solved before
        // anything reads them, which is why -inf bounds make every carried row dead)
        for (int k = 0; k < K; k++) tree_begin_step(trees[k], x0s.data() + (size_t)k * nx, nx);
        long rounds = 0;
        for (;;) {
            size_t npick = 0;
            for (int k = 0; k < K; k++) { tree_select(trees[k], width, 0.0, picks[k]); npick += picks[k].size(); }
            if (npick == 0) break;
            launch.clear();
            size_t B = 0;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) {
                    if (B == 0) break;
                    h_fix.assign(B * nfix, 0);
                    h_x0.assign(B * nx, 0.0);
                    h_widx.assign(B, -1);
                }
                size_t b = 0;
                for (int k = 0; k < K; k++) {
                    FleetTree &t = trees[k];
                    for (int i : picks[k])
                        tree_expand(t, i, nfix, speculation, dive, handdown, level, next, [&](const int8_t *row, int depth, int32_t widx) {
                            if (pass == 1) {
                                std::memcpy(h_fix.data() + b * nfix, row, nfix);
                                std::memcpy(h_x0.data() + b * nx, t.x0.data(), nx * sizeof(double));
                                h_widx[b] = widx;
                                launch.push_back({k, depth});
                            }
                            b++;
                        });
                }
                B = b;
            }
            if (B > 0) {
                std::vector<double> obj(B), dobj(B), primal(B * n_primal), dual(B * n_dual);
                std::vector<int> status(B), iters(B), polished(B);
                bool any_warm = false;
                for (size_t q = 0; q < B; q++) any_warm |= h_widx[q] >= 0;
                const int rc = solve(nx, nu, nub, T, nc, ncL, nq, nr, nqT, A.data(), Bm.data(), F.data(), G.data(), h.data(), FT.data(), GT.data(), hT.data(),
                                     Q.data(), R.data(), QT.data(), h_x0.data(), nx, (int)B, h_fix.data(), 1e-8, 1e-6, 100, 4, 1, 1, 1, 1e-4,
                                     any_warm ? pool_primal.data() : nullptr, any_warm ? pool_dual.data() : nullptr, any_warm ? h_widx.data() : nullptr,
                                     obj.data(), dobj.data(), status.data(), iters.data(), primal.data(), dual.data(), polished.data());
                if (rc != 0) { fprintf(stderr, "oracle failed: %d\n", rc); return 3; }
                rounds++;
                for (size_t q = 0; q < B; q++) {
                    const int d = launch[q].depth;
                    const double *nu_ = dual.data() + q * n_dual + o_lb;
                    FleetResult e{obj[q], d < nfix ? nu_[d] : 0.0, d < nfix ? nu_[nfix + d] : 0.0, (int32_t)(used + q),
                                  status[q] == 0 && (polished[q] & 0xff) != 0, status[q] > 1};
                    if (dive && e.vertex && d < nfix) {
                        std::vector<int8_t> bits(nfix);
                        const double *u = primal.data() + q * n_primal + (size_t)(T + 1) * nx;
                        for (int j = 0; j < nfix; j++) bits[j] = u[(j / nub) * nu + nuc + (j % nub)] > 0.5 ? 1 : 0;
                        trees[launch[q].k].rounded.emplace(e.row, std::move(bits));
                    }
                    trees[launch[q].k].cache.emplace(tree_key(h_fix.data() + q * nfix, d), e);
                }
                pool_dual.insert(pool_dual.end(), dual.begin(), dual.end());
                pool_primal.insert(pool_primal.end(), primal.begin(), primal.end());
                used += B;
            }
            for (int k = 0; k < K; k++) {
                const int bad = tree_consume(trees[k], picks[k], nfix, 0.0);
                if (bad) { fprintf(stderr, "tree_consume: %d\n", bad); return 4; }
            }
        }
        printf("%s[", step ? "," : "");
        for (int k = 0; k < K; k++) {
            FleetTree &t = trees[k];
            printf("%s{\"cost\": %.17g, \"solves\": %d, \"leaves\": %d, \"rounds\": %ld}", k ? "," : "", t.running && t.inc >= 0 ? t.ub : 1e300, t.solves, tree_leaves(t), rounds);
            if (t.running && t.inc >= 0) t.primal.assign(pool_primal.begin() + (size_t)t.inc_row * n_primal, pool_primal.begin() + (size_t)(t.inc_row + 1) * n_primal);
            else t.running = false;
            t.cache.clear();
        }
        printf("]");
        // retain / adopt: the next state is the model's (no error); every retained leaf is reopened (bound -inf)
        for (int k = 0; k < K; k++) {
            FleetTree &t = trees[k];
            if (!t.running) continue;
            tree_retain(t, t.primal.data() + (size_t)(T + 1) * nx, nuc, nub, nfix, keep[k]);
            std::vector<double> lb(keep[k].size(), -inf);
            std::vector<uint8_t> flags(keep[k].size(), 3);
            tree_adopt_shifted(t, keep[k], lb.data(), flags.data(), 0, nub, nfix);
            std::memcpy(x0s.data() + (size_t)k * nx, t.primal.data() + nx, nx * sizeof(double)); // x_1 of the incumbent
        }
    }
    printf("]\n");
    {   // a node pruned WITHOUT a certificate (HMPC_ITERS_UNCERTIFIED) is counted, with the bound it carried before its solve:
        // what hmpc_fleet_uncertified reports and what decides whether a search's optimum rests on such a prune
        FleetTree t;
        tree_reset_cold(t, nfix);
        tree_begin_step(t, x0s.data(), nx);
        t.lb[0] = 0.25;
        t.cache.emplace(tree_key(t.fix.data(), 0), FleetResult{inf, 0.0, 0.0, 0, false, false, true});
        const std::vector<int> pk{0};
        if (tree_consume(t, pk, nfix, 0.0) != 0 || t.uncertified != 1 || t.unc_lb != 0.25 || t.lb[0] != inf || t.solves != 1) {
            fprintf(stderr, "uncertified prune: not accounted for\n");
            return 5;
        }
        tree_begin_step(t, x0s.data(), nx);
        if (t.uncertified != 0 || t.unc_lb != inf) { fprintf(stderr, "uncertified prune: a step does not start clean\n"); return 5; }
    }
    // (no dlclose: the OpenMP runtime the oracle brought in keeps worker threads alive)
    return 0;
}
