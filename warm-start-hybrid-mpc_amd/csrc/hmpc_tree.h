// hmpc_tree.h -- the host-side bookkeeping of one branch-and-bound tree of the fleet driver (hmpc_fleet.hip), without
// anything of HIP in it: topology and bounds (identifiers, lower bounds, which pool row a node carries), candidate
// selection, what rides in a launch (picked nodes, speculative descendants, predicted dives), prune / incumbent / branch
// on the results, and the retain / adopt steps either side of the node shift.  Semantics of the reference:
//   selection            warm_start_hmpc/branch_and_bound.py:462-470, 541-563 (best first, first wins ties)
//   prune / incumbent / branch   branch_and_bound.py:476-489; children in the order [0-branch, 1-branch], their bounds
//                        parent bound + multiplier of the tightened bound (controller.py:13-44, 395-429)
//   retain rule, shift   controller.py:431-564 (a leaf survives if its first-stage binaries agree with the applied input)
// Kept apart so that it can be driven by recorded or CPU-computed QP results under AddressSanitizer / UBSan
// (tests/host/tree_driver.cpp, tests/test_sanitizers.py): the GPU pool has no device sanitizer.
#ifndef HMPC_TREE_H
#define HMPC_TREE_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

struct FleetResult { // a solved node waiting to be consumed by the search (speculative expansion)
    double obj, nu_lb, nu_ub; // objective; multipliers of the two bounds of the next binary in time
    int32_t row;              // its row in the pools
    bool vertex;              // optimal and polished: its record may be handed down to its children (hmpc_warm)
    bool failed;              // the solver did not converge on it (MAXITER / NUMERICAL): an error IF the search consumes it
    bool uncertified = false; // infeasible on the collapse of tau alone, no ray even loosely verified (HMPC_ITERS_UNCERTIFIED)
};

struct FleetTree {
    std::vector<int8_t> fix;   // n x nfix, -1 free / 0 / 1 (chronological prefixes)
    std::vector<double> lb;    // lower bound, +inf: proved infeasible
    std::vector<int32_t> row;  // dual row of the current pool the node carries (own if solved, else parent's), -1: none
    std::vector<int32_t> wrow; // row of the parent's record (primal and dual pools of THIS step) to hand down, -1: none
    std::vector<uint8_t> alive;
    std::vector<int16_t> depth; // fixed binaries
    int n = 0;
    double ub = std::numeric_limits<double>::infinity();
    int inc = -1;               // incumbent node
    int32_t inc_row = -1;       // its row in the primal pool
    std::vector<double> primal; // its primal row
    int solves = 0;
    int uncertified = 0;        // nodes of this step pruned without a certificate ...
    double unc_lb = std::numeric_limits<double>::infinity(); // ... and the smallest bound such a node carried before its solve
    bool running = true;        // false once the MIQP of a step was infeasible (the loop has ended)
    std::vector<double> x0;     // state of the last solve
    std::unordered_map<std::string, FleetResult> cache; // key: the fixed prefix of the identifier
    std::unordered_map<int32_t, std::vector<int8_t>> rounded; // dive prediction: a solved vertex node's relaxed binaries, rounded, by pool row
};

inline void tree_reset_cold(FleetTree &t, int nfix)
{
    t.fix.assign(nfix, (int8_t)-1);
    t.lb.assign(1, -std::numeric_limits<double>::infinity());
    t.row.assign(1, -1);
    t.wrow.assign(1, -1);
    t.alive.assign(1, 1);
    t.depth.assign(1, 0);
    t.n = 1;
    t.running = true;
}

// a step begins: no incumbent, no results waiting
inline void tree_begin_step(FleetTree &t, const double *x0, int nx)
{
    t.ub = std::numeric_limits<double>::infinity();
    t.inc = -1;
    t.inc_row = -1;
    t.solves = 0;
    t.uncertified = 0;
    t.unc_lb = std::numeric_limits<double>::infinity();
    t.cache.clear();
    t.rounded.clear();
    t.x0.assign(x0, x0 + nx);
}

inline std::string tree_key(const int8_t *fx, int depth) { return std::string((const char *)fx, (size_t)depth); }

// candidates: alive, bound below the incumbent; the `width` smallest bounds, first wins ties
inline void tree_select(const FleetTree &t, int width, double tol, std::vector<int> &picks)
{
    picks.clear();
    if (!t.running) return;
    for (int i = 0; i < t.n; i++)
        if (t.alive[i] && t.lb[i] < t.ub - tol) picks.push_back(i);
    std::stable_sort(picks.begin(), picks.end(), [&](int a, int b) { return t.lb[a] < t.lb[b]; });
    if ((int)picks.size() > width) picks.resize(width);
}

// What has to be launched for picked node i: the node itself unless its result is waiting, its descendants through the next
// `speculation` binaries, and -- dive prediction -- the rest of the dive its parent's rounded relaxed binaries predict, with
// the sibling of every step.  emit(identifier row, depth, row of the record to hand down or -1) is called once per node.
template <class Emit>
inline void tree_expand(const FleetTree &t, int i, int nfix, int speculation, bool dive, bool handdown, std::vector<int8_t> &level,
                        std::vector<int8_t> &next, Emit emit)
{
    const int8_t *fx = t.fix.data() + (size_t)i * nfix;
    if (t.cache.count(tree_key(fx, t.depth[i]))) return;
    level.assign(fx, fx + nfix);
    int depth = t.depth[i];
    for (int s = 0; s <= speculation; s++) {
        const size_t cnt = level.size() / nfix;
        next.clear();
        for (size_t q = 0; q < cnt; q++) {
            const int8_t *row = level.data() + q * nfix;
            // hand-down: the picked node receives its parent's record (solved in an earlier round of this step); a
            // speculative descendant's parent rides in this very launch
            if (s == 0 || !t.cache.count(tree_key(row, depth))) emit(row, depth, (s == 0 && handdown) ? t.wrow[i] : -1);
            if (s < speculation && depth < nfix)
                for (int v = 0; v < 2; v++) {
                    next.insert(next.end(), row, row + nfix);
                    next[next.size() - nfix + depth] = (int8_t)v;
                }
        }
        if (next.empty()) break;
        level.swap(next);
        depth++;
    }
    if (dive && t.wrow[i] >= 0) {
        auto pr = t.rounded.find(t.wrow[i]);
        if (pr != t.rounded.end()) {
            level.assign(fx, fx + nfix); // the predicted path, one binary more per step
            for (int j = t.depth[i]; j < nfix; j++) {
                for (int side = 0; side < 2; side++) { // the sibling of the step, then the step itself
                    level[j] = side == 0 ? (int8_t)(1 - pr->second[j]) : pr->second[j];
                    if (t.cache.count(tree_key(level.data(), j + 1))) continue;
                    emit(level.data(), j + 1, -1); // (its parent rides in this very launch)
                }
            }
        }
    }
}

// prune / incumbent / branch, node by node in selection order (branch_and_bound.py:476-489).
// Returns 0, or 1: a selected node has no result, 2: the solver did not converge on a node the search consumes.
inline int tree_consume(FleetTree &t, const std::vector<int> &picks, int nfix, double tol)
{
    for (int i : picks) {
        auto it = t.cache.find(tree_key(t.fix.data() + (size_t)i * nfix, t.depth[i]));
        if (it == t.cache.end()) return 1;
        const FleetResult e = it->second;
        t.cache.erase(it);
        // (a speculative descendant that did not converge is an error only here, when the search gets to it: the result of a
        // step does not depend on what rode along)
        if (e.failed) return 2;
        const double obj = e.obj;
        t.solves++;
        if (e.uncertified) { // (pruned on the collapse of tau alone: counted, and reported if the step's optimum rests on it)
            t.uncertified++;
            t.unc_lb = std::min(t.unc_lb, t.lb[i]);
        }
        t.lb[i] = obj;
        t.row[i] = e.row;
        const double cutoff = t.ub - tol;
        if (obj >= cutoff) continue;
        const int d = t.depth[i];
        if (d == nfix) { // every binary fixed: new incumbent
            t.ub = obj;
            t.inc = i;
            t.inc_row = e.row; // (its primal row is fetched once, at the end of the step, with the others')
        } else { // branch on the next binary in time; child bound = parent bound + multiplier of the tightened bound
            for (int v = 0; v < 2; v++) {
                const size_t c = t.n;
                t.fix.resize((c + 1) * nfix);
                std::memcpy(t.fix.data() + c * nfix, t.fix.data() + (size_t)i * nfix, nfix);
                t.fix[c * nfix + d] = (int8_t)v;
                t.lb.push_back(obj + (v == 1 ? e.nu_lb : e.nu_ub));
                t.row.push_back(e.row);
                t.wrow.push_back(e.vertex ? e.row : -1);
                t.alive.push_back(1);
                t.depth.push_back((int16_t)(d + 1));
                t.n++;
            }
            t.alive[i] = 0;
        }
    }
    return 0;
}

inline int tree_leaves(const FleetTree &t)
{
    int c = 0;
    for (int i = 0; i < t.n; i++) c += t.alive[i];
    return c;
}

// Retain rule of the node shift (controller.py:431-501): the leaves whose first-stage binaries agree with the applied
// binaries u_b (rounded).  u: the incumbent's inputs of stage 0 (nu entries).
inline void tree_retain(const FleetTree &t, const double *u, int nuc, int nub, int nfix, std::vector<int> &keep)
{
    keep.clear();
    for (int i = 0; i < t.n; i++) {
        if (!t.alive[i]) continue;
        bool agree = true;
        for (int q = 0; q < nub && agree; q++) {
            const int fq = t.fix[(size_t)i * nfix + q];
            agree = fq < 0 || fq == (int)std::rint(u[nuc + q]);
        }
        if (agree) keep.push_back(i);
    }
}

// The shifted leaves become the next tree: identifiers move one stage towards the present; leaf j of `keep` carries pool
// row row0 + j, the bound lb[j]; flags[j] bit 1: the shift reopened it.  Returns the number of reopened leaves.
inline int tree_adopt_shifted(FleetTree &t, const std::vector<int> &keep, const double *lb, const uint8_t *flags, int32_t row0, int nub, int nfix)
{
    const size_t n = keep.size();
    std::vector<int8_t> nfixv(n * nfix, (int8_t)-1);
    std::vector<double> nlb(n);
    std::vector<int32_t> row(n);
    std::vector<int16_t> depth(n);
    int reop = 0;
    for (size_t j = 0; j < n; j++) {
        const int i = keep[j];
        std::memcpy(nfixv.data() + j * nfix, t.fix.data() + (size_t)i * nfix + nub, nfix - nub);
        nlb[j] = lb[j];
        reop += (flags[j] & 2) != 0;
        row[j] = row0 + (int32_t)j; // (a reopened leaf carries its row too: it is re-solved before anything reads it)
        depth[j] = (int16_t)std::max(0, (int)t.depth[i] - nub);
    }
    t.fix.swap(nfixv);
    t.lb.swap(nlb);
    t.row.swap(row);
    t.wrow.assign(n, -1); // (the records of the step that ends here are not handed down across the shift)
    t.depth.swap(depth);
    t.alive.assign(n, 1);
    t.n = (int)n;
    return reop;
}

#endif // HMPC_TREE_H
