// hmpc_kernel.hip -- batched QP relaxations of hybrid-MPC branch-and-bound nodes on gfx950.
//
// One workgroup of 1, 2 or 4 wavefronts owns one node's QP from start to finish (the host picks the
// number of waves from the batch size: one wave per node for throughput, more for latency):
//   * everything indexed by stage -- iterate, two sets of Newton directions, residuals, the Riccati
//     factor (elimination multipliers, reciprocal pivots, cost-to-go) -- and the one row-indexed vector
//     other lanes must see (in turn z, D = z/s, D .* rhs, dz) live in LDS, addressed through
//     address-space-3 pointers so that every access is a ds_read / ds_write; node-independent data
//     ([A B], Hessians, column and Gram lists of the regular stage) is staged there once per workgroup;
//   * per-row state (slack, multiplier / barrier weight, step, affine product) lives in REGISTERS of
//     the lane that owns the row.  For the compile-time shapes a lane owns the same local row in every
//     slot, so the row's coefficients are registers too (RowMapS); the generic kernel walks sparse
//     lists and keeps the row state in a per-workgroup slab of global memory (RowMapL);
//   * the sequential recursions (stage elimination, forward / back substitution) run in registers of
//     wave 0 with v_readlane broadcasts, software pipelined against the LDS fetches of the next stage;
//   * the terminal-set rows (a dense block that exists only at the last stage and is masked in most
//     solves) stay in global memory (read-only, L2 resident) and are touched only when active;
//   * reductions (complementarity, residual norms, step length) are wave shuffles plus one LDS
//     exchange between waves -- no atomics, so a node's result does not depend on the batch it is in;
//   * problems too large for LDS take the generic kernel's streaming form (Dims::kBig: lists and factor in a global
//     slab, one node per CU with four waves): wave 0 runs the recursions -- the panel of the elimination, the sweeps
//     of a solve on padded stage blocks staged by the other waves --, waves 1 .. 3 hold the tiles of the stage matrix
//     on the matrix cores and prepare the next stage's while wave 0 eliminates (factor_tiles, kkt_sweeps_wave).
//
// Algorithm (same as the CPU oracle, oracle/hsde_qp.c): Mehrotra predictor-corrector on the
// homogeneous embedding of the QP; each KKT solve is a Riccati sweep over the horizon with fixed
// binaries handled as prescribed variables, the factor kept in substitution form; up to two steps of
// iterative refinement on the combined direction; terminal-set rows are tried masked first ("lazy
// terminal set").
//
// The reference (warm_start_hmpc/controller.py:229-271 -> bounded_qp.py:200-228) solves these QPs
// one at a time inside Gurobi; conventions of the output record follow
// warm_start_hmpc/subproblem_solution.py:68-168 and bounded_qp.py:260-332.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "hmpc_device.h"

#define WAVE 64
// active-set polish (same constants as oracle/hsde_qp.c): penalty of the active rows, proximal weight of the
// inactive rows (must stay above eps * rho), multiplier steps per active set, active sets per attempt
#define HMPC_REFINE_FROM_IT 12
#define HMPC_POLISH_RHO 1e5
#define HMPC_POLISH_RHO2 1e7 // second level, for active sets whose multiplier steps do not settle at the first
#define HMPC_POLISH_DELTA 1e-10
#define HMPC_RHO_OF(level) ((level) == 1 ? HMPC_POLISH_RHO2 : HMPC_POLISH_RHO)
#define HMPC_POLISH_ITERS 5
#define HMPC_POLISH_ROUNDS 6
#define HMPC_POLISH_ROUNDS_LAST 10 // the last attempt, on the iterate the solve would return (round 3, with the rule that every row with a negative multiplier leaves)
#define HMPC_ESC_MAX 2 // tolerance escalations after a failed last attempt (ipm_solve) ...
#define HMPC_ESC_ITERS 6 // ... and iterations an escalated solve may spend without meeting its tolerance
// ONE FEATURE SET FOR EVERY KERNEL (round 5).  Until round 4 three accuracy features of the run-time-sized kernels -- the
// tolerance escalation of the polish, lam_0 from its stationarity row (3.11a) and the multipliers the step's
// stationarity rows define taken from the step actually made (3.11b) -- were left out of the register kernels of the two
// cart-pole shapes for their register budget (and the third out of every register kernel).  The register kernels compiled at
// hmpc_create serve arbitrary systems: on a random MLD with nx = 6, nu = 2 + 3, N = 12 the missing third feature left the dual
// residual of deep nodes growing with the barrier weights from iteration 14 on (7.6e-10, 4.9e-9, ... 3.1e-6 where the
// run-time-sized kernel has 5e-12 ... 6e-15: profiles/r05_gvs_trace.txt) -- four optimal nodes of 2048 came back unpolished and
// one undecided (the parity hole of VERDICT round 4).  All three are now part of every kernel, shipped or compiled, and the
// shipped cart-pole instantiations are the same code as the kernels compiled for a problem.
#define HMPC_ESC_ENABLED(D) true
#define HMPC_LAM0_ROW 1
#ifndef HMPC_STABLE_DEN
#define HMPC_STABLE_DEN 1
#endif
#ifndef HMPC_PAIR // (0: A/B builds without the paired solve of the register kernels, kkt_solve_reg_pair)
#define HMPC_PAIR 1
#endif
#ifndef HMPC_NARROW_PANEL // (0: diagnostic builds without the narrow panel of stages whose binaries are all fixed)
#define HMPC_NARROW_PANEL 1
#endif
#define HMPC_POLISH_ATTEMPTS 3 // per solve: a node whose active set resists is left to the interior-point iterate
#define HMPC_RETRY (-1) // internal: a hand-down attempt with the terminal-set rows did not verify, run the regular sequence
#define HMPC_POLISH_ROUNDS_WARM 3 // active sets tried when the set is handed down by the parent node
#ifndef HMPC_POLISH_ROUNDS_OWN
#define HMPC_POLISH_ROUNDS_OWN 8 // ... and when it is the node's own (second launch of the two-launch form of the lazy terminal set)
#endif
#define HMPC_POLISH_WARM_BMOVE 0.5 // a hand-down is not tried where a fixed binary lies further than this from the parent's value
#define HMPC_POLISH_WARM_VMAX 1e-2 // a handed-down set whose point misses an inactive row by more is dropped at once
#ifndef HMPC_KERNEL_ATTR
#define HMPC_KERNEL_ATTR
#endif
#define DEV __device__ __forceinline__
// Scheduling fence around a batch of LDS loads in the register kernels (ccol_dot, the Gram phase of factor_reg): keeps the
// compiler's default scheduler from pairing every load with its use.  -DHMPC_NO_FENCE: without (A/B under the ILP schedule).
#ifdef HMPC_NO_FENCE
#define HMPC_FENCE() do { } while (0)
#else
#define HMPC_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Diagnostic build only (-DHMPC_STAMPS): cycle stamps per phase of the interior-point loop,
// accumulated for node 0 into the trace buffer.  No stamp executes in the shipped kernel.
#ifdef HMPC_STAMPS
#define DBG_SKIP(bit) (p.dbg & (1 << (bit)))
#define STAMP(k) do { long long now_ = clock64(); tacc[k] += now_ - tlast; tlast = clock64(); } while (0)
#define FSTAMP(k) do { long long now_ = clock64(); facc[k] += now_ - flast; flast = clock64(); } while (0)
#define FSTAMP_DECL long long flast = clock64()
static __device__ long long facc_dump[16];
#define FSTAMP_ARGS , long long *facc
#define FSTAMP_PASS , facc
#else
#define DBG_SKIP(bit) false
#define STAMP(k) do { } while (0)
#define FSTAMP(k) do { } while (0)
#define FSTAMP_DECL do { } while (0)
#define FSTAMP_ARGS
#define FSTAMP_PASS
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (its
// release covers global memory): in a loop whose threads exchange data through LDS alone but also issue global
// stores nobody reads before the kernel's next full barrier (the multipliers of the streaming factorisation),
// that drain puts one HBM / L2 write latency on every step.
DEV void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

DEV double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
DEV double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
DEV double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

// Reduction of N values over the whole workgroup (the node's NW waves): shuffles inside each
// wave, then one LDS exchange.  ops[i]: 0 sum, 1 max, 2 min.  Deterministic (fixed tree).
template <class D, int N>
DEV void block_reduce(double (&v)[N], const int (&op)[N], __attribute__((address_space(3))) double *red, int tid)
{
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = op[i] == 0 ? wave_sum(v[i]) : op[i] == 1 ? wave_max(v[i]) : wave_min(v[i]);
    if constexpr (D::kNW > 1) {
        if ((tid & 63) == 0) {
#pragma unroll
            for (int i = 0; i < N; i++) red[(tid >> 6) * N + i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; i++) {
            double a = red[i];
#pragma unroll
            for (int w = 1; w < D::kNW; w++) {
                const double b = red[w * N + i];
                a = op[i] == 0 ? a + b : op[i] == 1 ? fmax(a, b) : fmin(a, b);
            }
            v[i] = a;
        }
        __syncthreads();
    }
}
template <class D> DEV double block_sum(double x, __attribute__((address_space(3))) double *red, int tid)
{
    double v[1] = {x};
    const int op[1] = {0};
    block_reduce<D, 1>(v, op, red, tid);
    return v[0];
}
template <class D> DEV double block_max(double x, __attribute__((address_space(3))) double *red, int tid)
{
    double v[1] = {x};
    const int op[1] = {1};
    block_reduce<D, 1>(v, op, red, tid);
    return v[0];
}
template <class D> DEV double block_min(double x, __attribute__((address_space(3))) double *red, int tid)
{
    double v[1] = {x};
    const int op[1] = {2};
    block_reduce<D, 1>(v, op, red, tid);
    return v[0];
}

// Broadcast of one lane's double to the whole wave through two v_readlane (the result is wave
// uniform and lives in SGPRs: no LDS round trip).  src must be wave uniform.
DEV double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Broadcast inside the first row of 16 lanes (source and readers in lanes 0..15) as ONE instruction: the 64-bit DPP move
// v_mov_b64_dpp row_newbcast:src.  The stage vectors of the compile-time shapes have nz <= 16 components on lanes
// 0 .. nz-1, so every broadcast of their recursions is of this kind; against the v_readlane pair it halves the instruction
// count of a broadcast, keeps the value in a VGPR (no VALU -> SGPR -> VALU hazard) and frees ~30 SGPRs that otherwise
// spill.  Lanes of the other rows read lane src of THEIR row: they hold no component, their values are never read.
// src must be a constant after unrolling (the DPP control is an immediate): the switch then folds to one case.
template <class D> DEV double bcast16(double v, int src)
{
    if constexpr (!D::kDpp) return bcast(v, src);
    const long x = __builtin_bit_cast(long, v);
    long r;
    switch (src) {
#define HMPC_DPP_CASE(n) case n: r = __builtin_amdgcn_update_dpp((long)0, x, 0x150 + n, 0xf, 0xf, false); break;
        HMPC_DPP_CASE(0) HMPC_DPP_CASE(1) HMPC_DPP_CASE(2) HMPC_DPP_CASE(3) HMPC_DPP_CASE(4) HMPC_DPP_CASE(5) HMPC_DPP_CASE(6) HMPC_DPP_CASE(7)
        HMPC_DPP_CASE(8) HMPC_DPP_CASE(9) HMPC_DPP_CASE(10) HMPC_DPP_CASE(11) HMPC_DPP_CASE(12) HMPC_DPP_CASE(13) HMPC_DPP_CASE(14)
#undef HMPC_DPP_CASE
    default: r = __builtin_amdgcn_update_dpp((long)0, x, 0x150 + 15, 0xf, 0xf, false); break;
    }
    return __builtin_bit_cast(double, r);
}

// Reciprocal to ~1 ulp: v_rcp_f64 (measured on MI355X: 4.5e-8 relative) and two Newton steps (2e-15, then
// rounding level).  The row loops divide by slacks and multipliers a dozen times per row and iteration;
// the IEEE division sequence is three times as long and its last-bit guarantees buy nothing there.
DEV double frcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}

// LDS pointers carry their address space so that every access is a ds_read / ds_write (a generic
// pointer would compile to flat_load: slower, and it ties up both memory counters).
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) int ldsi;
typedef __attribute__((address_space(3))) unsigned char ldsb;

// Sparse lists of one stage type, resident in LDS (regular stage) or in global memory (last stage).
template <class IP, class DP>
struct Lists {
    IP rptr, rcol, cptr, crow, gptr, grow;
    DP rval, cval, gval, h;
};
typedef Lists<const ldsi *, const ldsd *> ListsL;
typedef Lists<const int *, const double *> ListsG;

struct Lds {
    ldsd *w, *lam, *nuf;
    ldsd *e; // the one row-indexed LDS vector: in turn z (for C'z), D (for the Gram phase), D.*rhs and dz
    ldsd *Lm, *dinv, *Pr;         // elimination multipliers, reciprocal pivots, cost-to-go (packed)
    ldsd *rd, *rdyn, *edyn, *g, *pv;
    ldsd *w1, *lam1, *nuf1, *w2, *lam2, *nuf2;
    ldsd *Mm, *PA, *q, *mv;
    ldsd *red;         // exchange buffer of the workgroup reductions
    ldsi *flag;        // wave 0 -> workgroup: factorisation failed
    ldsd *x0;
    ldsi *fix;
    ldsd *AB, *P, *PT; // [A B] (nx x nz), scaled cost Hessians
    ldsi *ei, *ej;     // lower-triangle entry -> (i, j)
    ListsL L0;         // stage rows ([F G] and the bounds of the binaries), the same for every stage
    ListsG G0;         // the same lists left in global memory, and the Riccati factor in a global slab:
    double *LmG, *PrG; //   the streaming variant for problems whose factor does not fit in LDS (Dims::kBig)
    ldsd *Ls;          //   ... two chunks of `ring` stages of multipliers staged in LDS for the sweeps of a solve (padded rows, see kkt_sweeps_wave)
    ldsd *Lw;          //   ... and the multipliers of the stage a factorisation is working on (packed), flushed to the slab per stage
    int abst, nup, lrows; // row stride of S.AB; columns / rows of a padded stage block in S.Ls
    // generic kernel, nz >= 16 (DevProb::split_lds; Cdn null otherwise): dense stage rows (ndp x nz), their local rows,
    // per row: singleton column or -(dense index + 1), singleton coefficient; singleton rows by column
    const ldsd *Cdn, *sval;
    const ldsi *drow, *rinfo, *sptr, *srow, *nrow; // (nrow: the rows that are not dense, mreg - nd of them)
    const ldsd *ccv;   // compile-time shapes: the columns of the stage rows padded to kKC entries each
    const ldsb *cci;   //   (value, local row); the row lists are not staged at all (RowMapS holds rows in registers)
    int term_on;       // terminal-set rows active in the current solve
    unsigned long long fullfix; // bit t: every binary of stage t < 64 is fixed by the node (wave uniform)
};

// Problem dimensions: compile-time for the instantiated shapes (index arithmetic folds to
// immediates, inner loops unroll, divisions become multiplies), run-time for the generic kernel.
template <int NX_, int NU_, int NUB_, int NW_, bool DPP_ = false>
struct Dims {
    // broadcasts of the register recursions as DPP row moves (bcast16) or v_readlane pairs: the DPP form keeps the
    // broadcast values in VGPRs and pays where the row state leaves room (kernels with few row slots: the 2- and
    // 4-wave variants, -10 % latency per launch); with 15 slots per lane (1 wave per node) it spills (336 B per lane)
    static constexpr bool kDpp = DPP_;
    static constexpr int kNX = NX_, kNU = NU_, kNUB = NUB_;
    static constexpr int kNW = NW_, kNT = NW_ * 64; // waves / threads per node (workgroup)
    // NX_ < 0: the generic kernel with the stage lists and the Riccati factor (multipliers, cost-to-go) in
    // global memory (L2 resident) instead of LDS -- for problems beyond 160 KiB of LDS per node
    static constexpr bool kBig = NX_ < 0;
    // entries per padded column of the stage rows (compile-time shapes): the cart-pole systems have at
    // most 14 (nu = 7) and 8 (nu = 4); a problem of the same shape with fuller columns takes the generic kernel
#ifdef HMPC_JIT_KC // (a kernel compiled at hmpc_create for a shape without a built-in instantiation, hmpc_jit.h: the host
                   // passes the longest column of the problem's stage rows, rounded up to an even number)
    static constexpr int kKC = NX_ > 0 ? HMPC_JIT_KC : 0;
#else
    static constexpr int kKC = NX_ > 0 ? (NU_ == 7 ? 14 : 8) : 0;
#endif
    static DEV int nx(const DevProb &p) { return NX_ > 0 ? NX_ : p.nx; }
    static DEV int nu(const DevProb &p) { return NU_ > 0 ? NU_ : p.nu; }
    static DEV int nub(const DevProb &p) { return NU_ > 0 ? NUB_ : p.nub; }
    static DEV int nuc(const DevProb &p) { return NU_ > 0 ? NU_ - NUB_ : p.nuc; }
    static DEV int nz(const DevProb &p) { return NX_ > 0 ? NX_ + NU_ : p.nz; }
    static DEV int ne(const DevProb &p) { return NX_ > 0 ? (NX_ + NU_) * (NX_ + NU_ + 1) / 2 : p.ne; }
};

// row stride of [A B] in LDS: the streaming form pads it (zero column nz, zero rows from nx on: the register
// recursions of its solves read past the ends instead of testing for them); nz everywhere else (a constant there)
#define AB_STRIDE (D::kBig ? S.abst : nz)
template <class D> DEV decltype(auto) stage_lists(const Lds &S)
{
    if constexpr (D::kBig) return (S.G0);
    else return (S.L0);
}
template <class D> DEV auto fac_lm(const Lds &S)
{
    if constexpr (D::kBig) return S.LmG;
    else return S.Lm;
}
template <class D> DEV auto fac_pr(const Lds &S)
{
    if constexpr (D::kBig) return S.PrG;
    else return S.Pr;
}

// Per-row values that only the owning lane touches: slack s, multiplier z, combined step dz, affine
// product / slack step prod.  The barrier weight D = z/s (0 on inactive rows) SHARES the storage of z:
// from the factorisation of an iteration to its update the slot holds D (and z is D * s wherever it is
// needed), otherwise it holds z.  Row residuals and the constant direction's dz are recomputed from
// the stage vectors when needed (a row product each) instead of being stored.
// RS > 0: RS slots per lane, kept in registers (every loop over slots is fully unrolled so the
// arrays are statically indexed).  RS == 0: run-time number of slots, kept in a per-workgroup slab
// of global memory (coalesced, L2 resident) -- the generic kernel.
template <int RS>
struct Rows {
    double s_[RS], zd_[RS], dz_[RS], prod_[RS];
    DEV double &s(int k, int) { return s_[k]; }
    DEV double &z(int k, int) { return zd_[k]; }
    DEV double &D(int k, int) { return zd_[k]; }
    DEV double &dz(int k, int) { return dz_[k]; }
    DEV double &prod(int k, int) { return prod_[k]; }
    DEV void bind(double *, int) {}
};
template <>
struct Rows<0> {
    double *__restrict__ s_, *__restrict__ zd_, *__restrict__ dz_, *__restrict__ prod_;
    DEV double &s(int, int r) { return s_[r]; }
    DEV double &z(int, int r) { return zd_[r]; }
    DEV double &D(int, int r) { return zd_[r]; }
    DEV double &dz(int, int r) { return dz_[r]; }
    DEV double &prod(int, int r) { return prod_[r]; }
    DEV void bind(double *base, int Mpad) { s_ = base; zd_ = base + Mpad; dz_ = base + 2 * Mpad; prod_ = base + 3 * Mpad; }
};
// Loop over the rows of this lane: slot k, row handle rw (from the row map rm; rw.e is the row's index in
// S.e and in the slab).  The lane-dependent part of the handle is made opaque to the optimiser in every
// loop: otherwise everything derived from it (stage, local row, a dozen addresses per slot) is hoisted
// out of the interior-point loop for all slots at once and spills to scratch memory.
#define ROW_OPAQUE(r) asm volatile("" : "+v"(r))
// The same for the thread index at the entry of a phase: addresses derived from it are recomputed per
// phase (a few integer operations) instead of being kept in registers across the whole iteration.
#define LANE_OPAQUE(lane) asm volatile("" : "+v"(lane))
// Diagnostic build only (-DHMPC_CHECK, `make check`): every row handle and every gathered index is bounds-checked, the
// per-node LDS vectors and the row slots are poisoned with NaNs at node start (a read of something never written shows up
// as a NaN record), violations raise bits in a flag word the host reads after each launch.  Nothing of this is compiled
// into the shipped kernel.
#ifdef HMPC_CHECK
#define HMPC_CHK(cond, bit) do { if (!(cond)) atomicOr(p.check_flag, 1u << (bit)); } while (0)
#else
#define HMPC_CHK(cond, bit) do { } while (0)
#endif
#define ROWS_BEGIN(k, rw)                                  \
    _Pragma("unroll") for (int k = 0; k < nslot; k++) {   \
        typename RM::Ref rw;                               \
        if (rm.at(p, k, lane, rw)) {                       \
            HMPC_CHK(rw.e >= 0 && rw.e < p.M, 0);
#define ROWS_END }}

// Row r -> (stage t, local row lr).  Rows [0, T*mreg) are the stage rows, mreg per stage; the
// terminal-set rows follow (r >= Toff) and belong to the last stage with lr = mreg + k.
DEV void row_decode(const DevProb &p, int r, int &t, int &lr)
{
    if (r >= p.Toff) {
        t = p.T - 1;
        lr = p.mreg + (r - p.Toff);
    } else {
        t = (int)__umulhi((unsigned)r, p.mreg_magic); // r / mreg, exact for r, mreg < 2^16
        lr = r - t * p.mreg;
    }
}

template <class D> DEV bool row_active(const DevProb &p, const ldsi *fix, int t, int lr, int term_on)
{
    const int nub = D::nub(p);
    if (lr >= p.mreg) return term_on;
    if (lr < p.nc) return true;
    int b = lr - p.nc;
    if (b >= nub) b -= nub;
    return fix[t * nub + b] < 0;
}

// index of (i, l) in a symmetric matrix stored as its packed lower triangle
DEV int sym(int i, int l) { return i >= l ? i * (i + 1) / 2 + l : l * (l + 1) / 2 + i; }

// Sparse dot products.  Each term is two dependent loads (index, then the indexed value); the
// terms are independent, so the loops are unrolled four wide to keep four chains in flight.
template <class IP, class DP> DEV double list_dot(IP idx, DP val, int k0, int k1, const ldsd *v)
{
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        const int i0 = idx[k], i1 = idx[k + 1], i2 = idx[k + 2], i3 = idx[k + 3];
        const double c0 = val[k], c1 = val[k + 1], c2 = val[k + 2], c3 = val[k + 3];
        a0 += c0 * v[i0];
        a1 += c1 * v[i1];
        a2 += c2 * v[i2];
        a3 += c3 * v[i3];
    }
    if (k + 2 <= k1) {
        const int i0 = idx[k], i1 = idx[k + 1];
        const double c0 = val[k], c1 = val[k + 1];
        a0 += c0 * v[i0];
        a1 += c1 * v[i1];
        k += 2;
    }
    if (k < k1) a2 += val[k] * v[idx[k]];
    return (a0 + a1) + (a2 + a3);
}
template <class L> DEV double row_dot(const L &st, int lr, const ldsd *v)
{
    return list_dot(st.rcol, st.rval, st.rptr[lr], st.rptr[lr + 1], v);
}
template <class L> DEV double col_dot(const L &st, int j, const ldsd *v)
{
    return list_dot(st.crow, st.cval, st.cptr[j], st.cptr[j + 1], v);
}
template <class L> DEV double gram(const L &st, int e, const ldsd *D)
{
    return list_dot(st.grow, st.gval, st.gptr[e], st.gptr[e + 1], D);
}
// C_t row / column products.  `base` is a row-indexed LDS vector (z, e, ...); `v` a stage vector.
template <class D> DEV double crow_dot(const DevProb &p, const Lds &S, int lr, const ldsd *v)
{
    if constexpr (D::kBig) {
        // streaming form with the dense stage rows in LDS: a dense product there beats a list walk in global memory
        // (two dependent loads per term at L2 latency)
        if (S.Cdn && lr < p.mreg) {
            const int nz = D::nz(p);
            const int ri = S.rinfo[lr];
            if (ri >= 0) return S.sval[lr] * v[ri]; // singleton row
            const ldsd *c = S.Cdn + (-ri - 1) * nz;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            int j = 0;
            for (; j + 4 <= nz; j += 4) { // (eight loads in flight per trip: the loop's instructions are the time of a row phase)
                const double c0 = c[j], c1 = c[j + 1], c2 = c[j + 2], c3 = c[j + 3];
                const double v0 = v[j], v1 = v[j + 1], v2 = v[j + 2], v3 = v[j + 3];
                a0 += c0 * v0;
                a1 += c1 * v1;
                a2 += c2 * v2;
                a3 += c3 * v3;
            }
            for (; j < nz; j++) a0 += c[j] * v[j];
            return (a0 + a1) + (a2 + a3);
        }
    }
    if (lr < p.mreg) return row_dot(stage_lists<D>(S), lr, v);
    const int nz = D::nz(p);
    const double *c = p.Ct + (size_t)(lr - p.mreg) * nz; // dense terminal row
    double a = 0;
    for (int j = 0; j < nz; j++) a += c[j] * v[j];
    return a;
}
template <class D> DEV double ccol_dot(const DevProb &p, const Lds &S, int t, int j, const ldsd *base)
{
    double a;
    if constexpr (D::kKC > 0) {
        // padded column: a fixed number of (value, row) pairs, all loads independent of each other
        const ldsd *eb = base + t * p.mreg, *cv = S.ccv + j * D::kKC;
        const ldsb *ci = S.cci + j * D::kKC;
        // two batches: all loads of a batch are issued before its first product (the scheduling fences keep
        // the compiler from pairing every load with its use, which would expose one LDS round trip per term)
        constexpr int H = D::kKC / 2;
        static_assert(D::kKC % 2 == 0, "padded columns come in two halves");
        double a0 = 0, a1 = 0;
#pragma unroll
        for (int hb = 0; hb < D::kKC; hb += H) {
            int idx[H];
            double cc[H], ee[H];
#pragma unroll
            for (int q = 0; q < H; q++) idx[q] = ci[hb + q];
#pragma unroll
            for (int q = 0; q < H; q++) cc[q] = cv[hb + q];
            HMPC_FENCE();
#pragma unroll
            for (int q = 0; q < H; q++) { HMPC_CHK(idx[q] >= 0 && idx[q] < p.mreg && j >= 0 && j < D::nz(p) && t >= 0 && t < p.T, 1); ee[q] = eb[idx[q]]; }
            HMPC_FENCE();
#pragma unroll
            for (int q = 0; q < H; q++) {
                if (q & 1) a1 += cc[q] * ee[q];
                else a0 += cc[q] * ee[q];
            }
        }
        a = a0 + a1;
    } else if (D::kBig && S.Cdn) {
        const int nz = D::nz(p);
        const ldsd *eb = base + t * p.mreg, *c = S.Cdn + j;
        double a0 = 0, a1 = 0;
        int r = 0;
        for (; r + 2 <= p.nd; r += 2) { // dense rows
            a0 += c[r * nz] * eb[S.drow[r]];
            a1 += c[(r + 1) * nz] * eb[S.drow[r + 1]];
        }
        if (r < p.nd) a0 += c[r * nz] * eb[S.drow[r]];
        for (int k = S.sptr[j]; k < S.sptr[j + 1]; k++) { // singleton rows of this column
            const int sr = S.srow[k];
            a1 += S.sval[sr] * eb[sr];
        }
        a = a0 + a1;
    } else {
        a = col_dot(stage_lists<D>(S), j, base + t * p.mreg);
    }
    if (S.term_on && t == p.T - 1) {
        if constexpr (D::kKC > 0) {
            a += S.mv[j]; // the terminal rows' part, reduced over the rows' owners beforehand (term_cols)
        } else {
            const int nz = D::nz(p);
            const ldsd *vt = base + p.Toff;
            double a0 = 0, a1 = 0;
            int k = 0;
            for (; k + 2 <= p.nT; k += 2) {
                a0 += p.Ct[(size_t)k * nz + j] * vt[k];
                a1 += p.Ct[(size_t)(k + 1) * nz + j] * vt[k + 1];
            }
            if (k < p.nT) a0 += p.Ct[(size_t)k * nz + j] * vt[k];
            a += a0 + a1;
        }
    }
    return a;
}

// Terminal-set part of C' v for the compile-time shapes: (C_T' v_T)(j), j < nz, into S.mv, for the ccol_dot calls
// that follow.  The terminal rows are dense and stay in global memory (L1 / L2 resident, the same for every node).  As a
// loop over the rows inside ccol_dot -- run by the eleven lanes that hold a component of the last stage while the rest
// of the wave waits -- the ~100 dependent global loads cost more than the whole rest of a solve: a node that needs the
// terminal set (lazy terminal set: a second solve) ran four times slower per iteration, and with closed-loop states --
// 6 % of the nodes of real trees -- those nodes set the makespan of a launch.  Here every lane takes the rows it owns
// in the row map's terminal slots (its loads are independent and in flight together) and the products are reduced over
// the workgroup.  `base` as in ccol_dot.  Must be called by all threads; ends with a barrier.
template <class D, class RM> DEV void term_cols(const DevProb &p, const Lds &S, int lane, const ldsd *base)
{
    if constexpr (D::kKC > 0) {
        if (!S.term_on) return;
        constexpr int NZ = D::kNX + D::kNU, KT = RM::kSlots - RM::kSlotsFB;
        double acc[NZ];
#pragma unroll
        for (int j = 0; j < NZ; j++) acc[j] = 0.0;
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const int row = k * D::kNT + lane;
            const bool ok = row < p.nT;
            const int rc = ok ? row : 0; // (clamped: a load under a lane predicate waits for itself)
            const double v = ok ? base[p.Toff + rc] : 0.0;
            const double *c = p.Ct + (size_t)rc * NZ;
#pragma unroll
            for (int j = 0; j < NZ; j++) acc[j] += c[j] * v;
        }
        if constexpr (D::kNW == 1) {
#pragma unroll
            for (int j = 0; j < NZ; j++) acc[j] = wave_sum(acc[j]);
        } else {
            constexpr int CH = 40 / D::kNW; // block_reduce exchanges kNW * N values through S.red (40 doubles)
#pragma unroll
            for (int j0 = 0; j0 < NZ; j0 += CH) {
                double v[CH];
                int op[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) { v[q] = j0 + q < NZ ? acc[j0 + q] : 0.0; op[q] = 0; }
                block_reduce<D, CH>(v, op, S.red, lane);
#pragma unroll
                for (int q = 0; q < CH; q++)
                    if (j0 + q < NZ) acc[j0 + q] = v[q];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NZ; j++) S.mv[j] = acc[j];
        }
        __syncthreads();
    }
}
template <class D> DEV double hrow(const DevProb &p, const Lds &S, int lr) { return lr < p.mreg ? stage_lists<D>(S).h[lr] : p.ht[lr - p.mreg]; }
// ---------------------------------------------------------------------------------------------
// Row maps: which rows a lane owns and how it evaluates them.
//
// RowMapL (generic kernel): rows in stage-major order, row r <-> lane r % kNT, slot r / kNT; every
// evaluation walks the sparse row list of the stage in LDS (dynamic trip counts, dependent loads).
//
// RowMapS (compile-time shapes): a lane owns the SAME local row in every slot, so the row itself lives
// in registers and a slot only changes the stage:
//   slots [0, KF)        [F G] rows: lane = tsub * nc + lr, slot k holds stage k * SP + tsub
//                        (SP = kNT / nc stages per slot); the row is 4 state coefficients and at most
//                        two input coefficients (checked on the host, DevProb::static_rows)
//   slots [KF, KF + KB)  bound rows of the binaries: lane = tsub * 2 nub + c, stage kb * SB + tsub
//   slots [KF + KB, ..)  terminal-set rows (dense, global memory, only in the second solve of a node)
// A row product is then a handful of LDS loads at addresses known up front: no list, no index loads,
// no loop -- the loads of all slots are in flight together.
// ---------------------------------------------------------------------------------------------
template <class D> struct RowMapL {
    struct Ref { int e, t, lr; };
    static constexpr int kSlots = 0;
    const ldsi *drow, *nrow; // streaming form with split stage rows: dense rows, the other rows (null: rows in their own order)
    DEV void init(const DevProb &, int) { drow = nullptr; nrow = nullptr; }
    DEV void bind(const DevProb &p, const Lds &S)
    {
        if constexpr (D::kBig) {
            if (p.split_lds) { drow = S.drow; nrow = S.nrow; }
        }
    }
    DEV void prepare(const DevProb &, const Lds &, int, int) {}
    DEV bool at(const DevProb &p, int k, int lane, Ref &rw) const
    {
        int r = k * D::kNT + lane;
        ROW_OPAQUE(r);
        if (r >= p.M) return false;
        if constexpr (D::kBig) {
            if (drow) {
                // Positions are dealt so that a wave's 64 rows are of one kind: first the dense rows of all stages (a product
                // with a dense row is ~nz terms), then the others (one term) -- in row order every wave held some dense
                // rows and ran the long form in every slot.  Terminal-set rows keep their places at the end.
                const int nds = p.T * p.nd;
                if (r < nds) {
                    const int t = (int)__umulhi((unsigned)r, p.nd_magic);
                    rw.t = t;
                    rw.lr = drow[r - t * p.nd];
                    rw.e = t * p.mreg + rw.lr;
                    return true;
                }
                if (r < p.Toff) {
                    const int q = r - nds, nn = p.mreg - p.nd;
                    const int t = (int)__umulhi((unsigned)q, p.nn_magic);
                    rw.t = t;
                    rw.lr = nrow[q - t * nn];
                    rw.e = t * p.mreg + rw.lr;
                    return true;
                }
            }
        }
        rw.e = r;
        row_decode(p, r, rw.t, rw.lr);
        return true;
    }
    DEV bool active(const DevProb &p, const Lds &S, int, const Ref &rw) const { return row_active<D>(p, S.fix, rw.t, rw.lr, S.term_on); }
    DEV double h(const DevProb &p, const Lds &S, int, const Ref &rw) const { return hrow<D>(p, S, rw.lr); }
    DEV double dot(const DevProb &p, const Lds &S, int, const Ref &rw, const ldsd *vec) const
    {
        return crow_dot<D>(p, S, rw.lr, vec + rw.t * D::nz(p));
    }
};

template <class D, int KF, int KB, int KT> struct RowMapS {
    DEV void bind(const DevProb &, const Lds &) {}
    static constexpr int NX = D::kNX, NU = D::kNU, NUB = D::kNUB, NUC = NU - NUB, NZ = NX + NU;
    static constexpr int kSlots = KF + KB + KT, kSlotsFB = KF + KB;
    static constexpr int SB = D::kNT / (2 * NUB); // stages per slot of bound rows
    // (2 NUB need not divide the workgroup -- three binaries: 10 stages of 6 rows on 64 lanes --: the lanes past the last
    // whole stage hold no bound row; for the shapes whose bound rows tile the workgroup the test folds away)
    static constexpr bool kAllB = D::kNT % (2 * NUB) == 0;
    static_assert(SB >= 1, "one stage's bound rows must fit the workgroup");
    struct Ref { int e, v, row; };
    double cx[NX], cu[2], hF, sgB, hB;
    int uo[2];
    int SP, sE, sV, tF0, eF0, vF0, tB0, eB0, vB0, bB;
    bool okF, okB;
    unsigned act; // bit k: the row of slot k takes part in the current solve
    DEV void init(const DevProb &p, int lane)
    {
        const int nc = p.nc;
        SP = D::kNT / nc;
        const int tsub = lane / nc, lr = lane - tsub * nc;
        okF = tsub < SP;
        tF0 = tsub; eF0 = tsub * p.mreg + lr; vF0 = tsub * NZ; sE = SP * p.mreg; sV = SP * NZ;
        const double *c = p.Creg + (size_t)(okF ? lr : 0) * NZ;
#pragma unroll
        for (int i = 0; i < NX; i++) cx[i] = c[i];
        cu[0] = cu[1] = 0.0;
        uo[0] = uo[1] = NX;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) {
            const double v = c[NX + j];
            if (v != 0.0) {
                if (cnt == 0) { cu[0] = v; uo[0] = NX + j; } else { cu[1] = v; uo[1] = NX + j; }
                cnt++;
            }
        }
        hF = p.reg.h[okF ? lr : 0];
        const int cB = lane % (2 * NUB);
        bB = cB % NUB;
        tB0 = lane / (2 * NUB);
        okB = kAllB || lane < SB * 2 * NUB;
        eB0 = tB0 * p.mreg + nc + cB;
        vB0 = tB0 * NZ + NX + NUC + bB;
        sgB = p.Creg[(size_t)(nc + cB) * NZ + NX + NUC + bB];
        hB = p.reg.h[nc + cB];
        act = 0;
    }
    DEV bool at(const DevProb &p, int k, int lane, Ref &rw) const
    {
        if (k < KF) {
            int e0 = eF0, v0 = vF0;
            ROW_OPAQUE(e0);
            ROW_OPAQUE(v0);
            rw.e = e0 + k * sE;
            rw.v = v0 + k * sV;
            rw.row = 0;
            return okF && tF0 + k * SP < p.T;
        } else if (k < KF + KB) {
            const int kb = k - KF;
            int e0 = eB0, v0 = vB0;
            ROW_OPAQUE(e0);
            ROW_OPAQUE(v0);
            rw.e = e0 + kb * SB * p.mreg;
            rw.v = v0 + kb * SB * NZ;
            rw.row = 0;
            return (kAllB || okB) && tB0 + kb * SB < p.T;
        } else {
            int row = (k - KF - KB) * D::kNT + lane;
            ROW_OPAQUE(row);
            rw.row = row;
            rw.e = p.Toff + row;
            rw.v = (p.T - 1) * NZ;
            return row < p.nT;
        }
    }
    // which rows take part in the coming solve: all [F G] rows, the bounds of the free binaries, the
    // terminal-set rows if term_on
    DEV void prepare(const DevProb &p, const Lds &S, int lane, int term_on)
    {
        act = 0;
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
            bool on;
            if (k < KF) on = true;
            else if (k < KF + KB) {
                const int t = tB0 + (k - KF) * SB;
                on = (kAllB || okB) && t < p.T && S.fix[(t < p.T ? t : 0) * NUB + bB] < 0;
            } else on = term_on != 0;
            act |= (on ? 1u : 0u) << k;
        }
    }
    DEV bool active(const DevProb &, const Lds &, int k, const Ref &) const { return (act >> k) & 1u; }
    DEV double h(const DevProb &p, const Lds &, int k, const Ref &rw) const { return k < KF ? hF : k < KF + KB ? hB : p.ht[rw.row]; }
    // The same product with the stage offset clamped to zero where the lane has no row in the slot: every lane loads from a
    // valid address, so a phase can run the loads of ALL slots before its first store (straight-line code: a branch per
    // slot exposes one LDS round trip per slot).  [F G] and bound slots only.
    DEV double dot_all(int k, bool ok, const Ref &rw, const ldsd *vec) const
    {
        const ldsd *v = vec + (ok ? rw.v : 0);
        if (k < KF) {
            double a = cx[0] * v[0];
#pragma unroll
            for (int i = 1; i < NX; i++) a += cx[i] * v[i];
            return (a + cu[0] * v[uo[0]]) + cu[1] * v[uo[1]];
        }
        return sgB * v[0];
    }
    DEV double dot(const DevProb &p, const Lds &, int k, const Ref &rw, const ldsd *vec) const
    {
        const ldsd *v = vec + rw.v;
        HMPC_CHK(rw.v >= 0 && (k >= KF && k < KF + KB ? rw.v < p.T * NZ : rw.v + NZ <= p.T * NZ) && uo[0] < NZ && uo[1] < NZ, 3);
        if (k < KF) {
            double a = cx[0] * v[0];
#pragma unroll
            for (int i = 1; i < NX; i++) a += cx[i] * v[i];
            return (a + cu[0] * v[uo[0]]) + cu[1] * v[uo[1]];
        } else if (k < KF + KB) {
            return sgB * v[0];
        } else {
            const double *c = p.Ct + (size_t)rw.row * NZ;
            double a = 0;
#pragma unroll
            for (int j = 0; j < NZ; j++) a += c[j] * v[j];
            return a;
        }
    }
};

// (C' D C)(i, j) of stage t: Gram lists of the stage rows, plus the dense terminal block if active
template <class D> DEV double gram_entry(const DevProb &p, const Lds &S, int t, int e, int i, int j)
{
    double a = gram(stage_lists<D>(S), e, S.e + t * p.mreg); // S.e holds D during the factorisation
    if (S.term_on && t == p.T - 1) {
        const int nz = D::nz(p);
        const ldsd *Dt = S.e + p.Toff;
        double a0 = 0, a1 = 0;
        int k = 0;
        for (; k + 2 <= p.nT; k += 2) {
            a0 += p.Ct[(size_t)k * nz + i] * p.Ct[(size_t)k * nz + j] * Dt[k];
            a1 += p.Ct[(size_t)(k + 1) * nz + i] * p.Ct[(size_t)(k + 1) * nz + j] * Dt[k + 1];
        }
        if (k < p.nT) a0 += p.Ct[(size_t)k * nz + i] * p.Ct[(size_t)k * nz + j] * Dt[k];
        a += a0 + a1;
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
// Riccati factorisation of Phi_t = P + C_t' D C_t.  For every stage, backwards in time,
//   M = Phi_t + [A B]' P_{t+1} [A B]
// is reduced by eliminating its inputs (pivot order u_0 .. u_{nu-1}; the fixed binaries are
// prescribed: identity rows / columns, unit pivots, skipped).  What is kept per stage is the LDL'
// factor in the form the solves need:
//   Lm   : elimination multipliers -- rows of the states ("L_x") and of the inputs ("L_u", strictly
//          lower triangular), one dense (nx + nu) x nu block
//   dinv : reciprocal pivots
//   P_t  : Schur complement = cost-to-go Hessian (packed lower triangle)
//   mb   : sum of the columns of M of binaries fixed to one, handed to the next solve in S.g
// A solve applies the same row operations to its vector (forward substitution) and a back
// substitution with L_u'.  Explicit inverses (M_uu^{-1}, gains) are NOT formed: they are not backward
// stable and break the iteration once D = z/s spans ~24 orders of magnitude (mu ~ 1e-12); this was
// observed on the GPU and reproduced on the CPU before this form was adopted.
// ---------------------------------------------------------------------------------------------
// per stage: the multipliers of the state rows ("L_x", nx x nu, dense: a lane's row is nu contiguous
// doubles) followed by the strictly lower triangle of L_u packed by rows (row i holds its i entries)
#define LM_STAGE(nx, nu) ((nx) * (nu) + (nu) * ((nu) - 1) / 2)
#define LM_X(nx, nu, x, j) ((x) * (nu) + (j))                       /* state row x, pivot j     */
#define LM_U(nx, nu, i, j) ((nx) * (nu) + (i) * ((i) - 1) / 2 + (j)) /* input row i, pivot j < i */

template <class D> DEV void stage_matrix_mfma(const DevProb &p, const Lds &S, int lane, int t, const ldsd *Pn, int pns);
template <class D> DEV bool factor_tiles_fits(const DevProb &p);
template <class D, int MQ> DEV int factor_tiles(const DevProb &p, const Lds &S, int lane FSTAMP_ARGS);

template <class D> DEV int factor(const DevProb &p, const Lds &S, int lane FSTAMP_ARGS)
{
    if constexpr (D::kNT == 4 * WAVE) {
        // (two tiles per tile wave: the 6 lower-triangle tiles of nz <= 48 over three waves.  An instantiation with four --
        // nz <= 64 -- beside it cost the WHOLE kernel 650 bytes of scratch per lane, measured: 18 GB of HBM traffic per
        // 1024 nodes of configs[4] against 3; wider stage vectors take the LDS form below.)
        if (factor_tiles_fits<D>(p) && !DBG_SKIP(4)) return factor_tiles<D, 2>(p, S, lane FSTAMP_PASS);
    }
    FSTAMP_DECL;
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nuc = D::nuc(p), nub = D::nub(p), ne = D::ne(p);
    const int nxs = nx * (nx + 1) / 2, lms = LM_STAGE(nx, nu);
    for (int e = lane; e < nx * nx; e += D::kNT) {
        const int i = e / nx, j = e - i * nx;
        if (i >= j) fac_pr<D>(S)[T * nxs + sym(i, j)] = S.PT[e];
    }
    __syncthreads();
    int bad = 0;
    for (int t = T - 1; t >= 0; t--) {
        const auto Pn = fac_pr<D>(S) + (t + 1) * nxs;
        const ldsi *fx = S.fix + t * nub;
        const auto Lm = fac_lm<D>(S) + t * lms;
        int nfixed = 0;
        for (int b = 0; b < nub; b++) nfixed += fx[b] >= 0;
        if (DBG_SKIP(0)) {
            for (int e = lane; e < nz * nz; e += D::kNT) S.Mm[e] = (e / nz == e % nz) ? 2.0 : 0.0;
            __syncthreads();
        } else if (nz >= 16 && p.mreg < 65536) {
            // dense contractions on the matrix cores; P_{t+1} is still where the previous stage's elimination
            // left it (the leading nx x nx block of S.Mm), the terminal Hessian for the last stage
            stage_matrix_mfma<D>(p, S, lane, t, t == T - 1 ? S.PT : S.Mm, t == T - 1 ? nx : nz);
            FSTAMP(0);
        } else {
        // M = P + C' D C (sparse Gram lists) ; PA = Pn [A B]
        for (int e = lane; e < ne; e += D::kNT) {
            const int i = S.ei[e], j = S.ej[e];
            const double a = S.P[i * nz + j] + gram_entry<D>(p, S, t, e, i, j);
            S.Mm[i * nz + j] = a;
            S.Mm[j * nz + i] = a;
        }
        for (int e = lane; e < nx * nz; e += D::kNT) {
            const int i = e / nz, j = e - i * nz;
            double a = 0;
            for (int l = 0; l < nx; l++) a += Pn[sym(i, l)] * S.AB[l * AB_STRIDE + j];
            S.PA[e] = a;
        }
        __syncthreads();
        FSTAMP(0);
        for (int e = lane; e < ne; e += D::kNT) {
            const int i = S.ei[e], j = S.ej[e];
            double a = S.Mm[i * nz + j];
            for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + i] * S.PA[l * nz + j];
            S.Mm[i * nz + j] = a;
            S.Mm[j * nz + i] = a;
        }
        __syncthreads();
        }
        FSTAMP(1);
        // the stage's multipliers are collected in LDS (the streaming form flushes the block to its slab once per
        // stage, coalesced, instead of one scattered global store per pivot and row)
        ldsd *Lw;
        if constexpr (D::kBig) Lw = S.Lw;
        else Lw = S.Lm + t * lms;
        if (!DBG_SKIP(1))
            for (int e = lane; e < lms; e += D::kNT) Lw[e] = 0.0;
        if (nfixed) {
            for (int i = lane; i < nz; i += D::kNT) {
                double a = 0;
                for (int b = 0; b < nub; b++)
                    if (fx[b] == 1) a += S.Mm[i * nz + nx + nuc + b];
                S.g[t * nz + i] = a;
            }
            __syncthreads();
            for (int e = lane; e < nz * nz; e += D::kNT) {
                const int i = e / nz, j = e - i * nz;
                const bool fi = i >= nx + nuc && fx[i - nx - nuc] >= 0;
                const bool fj = j >= nx + nuc && fx[j - nx - nuc] >= 0;
                if (fi || fj) S.Mm[e] = (i == j) ? 1.0 : 0.0;
            }
            __syncthreads();
        } else {
            for (int i = lane; i < nz; i += D::kNT) S.g[t * nz + i] = 0.0;
            __syncthreads();
        }
        FSTAMP(2);
        for (int j = 0; j < (DBG_SKIP(2) ? 0 : nu); j++) {
            if (j >= nuc && fx[j - nuc] >= 0) { // decoupled unit pivot: the solves skip it as well
                if (lane == 0) S.dinv[t * nu + j] = 1.0;
                continue;
            }
            const int pj = nx + j;
            const double d = S.Mm[pj * nz + pj];
            if (!(d > 0.0)) bad = 1;
            double rinv = __builtin_amdgcn_rcp(d);
            rinv = rinv * (2.0 - d * rinv); // one Newton step on the hardware reciprocal
            // rows still to be reduced: all states and the inputs after j.  In place: an entry is read
            // and written by its own thread only; the pivot row and column, which every thread reads,
            // are not part of the trailing block.  The thread of the first trailing column keeps the
            // row's multiplier.
            const int nrem = nx + (nu - 1 - j);
            const float inv_nrem = 1.0f / (float)nrem; // e / nrem through a float multiply: exact for e < 2^16
            for (int e = lane; e < nrem * nrem; e += D::kNT) {
                const int a = (int)(((float)e + 0.5f) * inv_nrem), bcol = e - a * nrem;
                const int i = a < nx ? a : pj + 1 + (a - nx);
                const int k = bcol < nx ? bcol : pj + 1 + (bcol - nx);
                const double mij = S.Mm[i * nz + pj] * rinv;
                S.Mm[i * nz + k] -= mij * S.Mm[pj * nz + k];
                if (bcol == 0 && !DBG_SKIP(3)) Lw[a < nx ? LM_X(nx, nu, a, j) : LM_U(nx, nu, i - nx, j)] = mij;
            }
            if (lane == 0) S.dinv[t * nu + j] = rinv;
            lds_barrier(); // the pivot steps exchange LDS data only
        }
        __syncthreads();
        FSTAMP(3);
        if constexpr (D::kBig)
            for (int e = lane; e < lms; e += D::kNT) Lm[e] = Lw[e];
        for (int e = lane; e < nx * nx; e += D::kNT) {
            const int i = e / nx, j = e - i * nx;
            if (i >= j) fac_pr<D>(S)[t * nxs + sym(i, j)] = S.Mm[i * nz + j];
        }
        __syncthreads();
        FSTAMP(4);
    }
    return bad ? -1 : 0;
}

// ---------------------------------------------------------------------------------------------
// The stage matrix of the run-time-sized kernel on the matrix cores (problems with nz >= 16, e.g. BASELINE
// configs[4]: nx = 20, nu = 14, 100 rows per stage).  All three contractions of
//   M = P + C' D C + [A B]' P_{t+1} [A B]
// are dense 16 x 16 x 4 tiles of v_mfma_f64_16x16x4_f64 (lane l feeds A[row l & 15][k = l >> 4] and
// B[k = l >> 4][col l & 15]; it receives rows (l >> 4) + 4 r, r < 4, of column l & 15): C is read as the dense
// mreg x nz matrix the host keeps for the static row map (L2 resident, the same for every node), D from the
// row vector in LDS, P_{t+1} from the LDS block the previous stage's elimination left it in.  The waves of the
// node share the tiles; the lower triangle is computed and mirrored.  The list walk it replaces spent its time
// on dependent global loads (index, then value) -- 70 k cycles per stage against ~20 k here.  Operands are read from
// clamped indices and zeroed afterwards where a tile overhangs the matrix: a read under a lane predicate sits in a branch
// of its own and waits for itself, which cost a quarter of this function's time.
// ---------------------------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

template <class FA, class FB> DEV mfma_d4 mfma_tile(int wl, int K, FA a, FB b, mfma_d4 acc)
{
    const int r = wl & 15, kq = wl >> 4;
    int k0 = 0;
    for (; k0 + 8 <= K; k0 += 8) { // two steps per round: the loads of the second are in flight under the first MFMA
        const double a0 = a(r, k0 + kq), b0 = b(k0 + kq, r), a1 = a(r, k0 + 4 + kq), b1 = b(k0 + 4 + kq, r);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 4) { // (reads from a clamped index, zeroed afterwards: a read under a lane predicate waits for itself)
        const int k = k0 + kq, kc = k < K ? k : K - 1;
        const double a0 = a(r, kc), b0 = b(kc, r);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < K ? a0 : 0.0, k < K ? b0 : 0.0, acc, 0, 0, 0);
    }
    return acc;
}

// Pn: cost-to-go Hessian of stage t + 1 as a dense block with row stride pns (LDS).  Result in S.Mm (dense nz x nz).
template <class D> DEV void stage_matrix_mfma(const DevProb &p, const Lds &S, int lane, int t, const ldsd *Pn, int pns)
{
    const int nx = D::nx(p), nz = D::nz(p);
    const int wl = lane & 63, wave = lane >> 6;
    const int tn = (nz + 15) >> 4, tx = (nx + 15) >> 4;
    // (1) PA = P_{t+1} [A B]  (nx x nz)
    for (int q = wave; q < tx * tn; q += D::kNW) {
        const int ti = q / tn, tj = q - ti * tn;
        mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile(wl, nx,
                        [&](int r, int k) { const int i = ti * 16 + r; const double v = Pn[(i < nx ? i : nx - 1) * pns + k]; return i < nx ? v : 0.0; },
                        [&](int k, int c) { const int j = tj * 16 + c; const double v = S.AB[k * AB_STRIDE + (j < nz ? j : nz - 1)]; return j < nz ? v : 0.0; }, acc);
        const int j = tj * 16 + (wl & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = ti * 16 + (wl >> 4) + 4 * r;
            if (i < nx && j < nz) S.PA[i * nz + j] = acc[r];
        }
    }
    __syncthreads();
    // (2) M = P + C' D C + [A B]' PA, lower-triangle tiles
    const ldsd *Dt = S.e + t * p.mreg; // S.e holds D during the factorisation
    const bool term = S.term_on && t == p.T - 1;
    for (int q = wave; q < tn * (tn + 1) / 2; q += D::kNW) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
        const int tj = q - ti * (ti + 1) / 2;
        mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile(wl, p.mreg,
                            [&](int r, int k) { const int i = ti * 16 + r; const double v = p.Creg[(size_t)k * nz + (i < nz ? i : nz - 1)] * Dt[k]; return i < nz ? v : 0.0; },
                            [&](int k, int c) { const int j = tj * 16 + c; const double v = p.Creg[(size_t)k * nz + (j < nz ? j : nz - 1)]; return j < nz ? v : 0.0; }, acc);
        if (term) {
            const ldsd *De = S.e + p.Toff;
            acc = mfma_tile(wl, p.nT,
                            [&](int r, int k) { const int i = ti * 16 + r; const double v = p.Ct[(size_t)k * nz + (i < nz ? i : nz - 1)] * De[k]; return i < nz ? v : 0.0; },
                            [&](int k, int c) { const int j = tj * 16 + c; const double v = p.Ct[(size_t)k * nz + (j < nz ? j : nz - 1)]; return j < nz ? v : 0.0; }, acc);
        }
        acc = mfma_tile(wl, nx,
                        [&](int r, int k) { const int i = ti * 16 + r; const double v = S.AB[k * AB_STRIDE + (i < nz ? i : nz - 1)]; return i < nz ? v : 0.0; },
                        [&](int k, int c) { const int j = tj * 16 + c; const double v = S.PA[k * nz + (j < nz ? j : nz - 1)]; return j < nz ? v : 0.0; }, acc);
        const int j = tj * 16 + (wl & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = ti * 16 + (wl >> 4) + 4 * r;
            if (i < nz && j < nz) {
                const double v = acc[r] + S.P[i * nz + j];
                S.Mm[i * nz + j] = v;
                if (ti != tj) S.Mm[j * nz + i] = v;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The factorisation of the run-time-sized kernel with four waves per node, 16 <= nz <= 48 and nu <= 16 (BASELINE
// configs[4]): factor_tiles below.  The 16 x 16 tiles of
//   M = P + C' D C + [A B]' P_{t+1} [A B]
// stay in the registers the matrix-core instruction returns them in (a lane: rows (l >> 4) + 4 r of column l & 15 of its
// tiles); their operands are fetched in batches -- all loads of a batch in flight before its first product -- and two tiles
// are interleaved where a wave has two.  First the helpers the tiles are computed with.
// What the stage leaves behind is what the LDS form leaves: multipliers, reciprocal pivots, P_t, mb.
// ---------------------------------------------------------------------------------------------
template <int NB, class FA, class FB> DEV mfma_d4 mfma_tile_batched(int wl, int K, FA a, FB b, mfma_d4 acc)
{
    const int r = wl & 15, kq = wl >> 4;
    int k0 = 0;
    for (; k0 + 4 * NB <= K; k0 += 4 * NB) {
        double av[NB], bv[NB];
#pragma unroll
        for (int s = 0; s < NB; s++) { av[s] = a(r, k0 + 4 * s + kq); bv[s] = b(k0 + 4 * s + kq, r); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NB; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    if (k0 < K) { // the last, partial batch: clamped indices, zeroed operands (a read under a lane predicate waits for itself)
        double av[NB], bv[NB];
#pragma unroll
        for (int s = 0; s < NB; s++) {
            const int k = k0 + 4 * s + kq, kc = k < K ? k : K - 1;
            const double x = a(r, kc), y = b(kc, r);
            av[s] = k < K ? x : 0.0;
            bv[s] = k < K ? y : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NB; s++)
            if (k0 + 4 * s < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    return acc;
}

// The same for K a multiple of four (uniform trip counts, no clamp or select anywhere): full batches, then single steps.
template <int NB, class FA, class FB> DEV mfma_d4 mfma_tile_k4(int wl, int K, FA a, FB b, mfma_d4 acc)
{
    const int r = wl & 15, kq = wl >> 4;
    int k0 = 0;
    for (; k0 + 4 * NB <= K; k0 += 4 * NB) {
        double av[NB], bv[NB];
#pragma unroll
        for (int s = 0; s < NB; s++) { av[s] = a(r, k0 + 4 * s + kq); bv[s] = b(k0 + 4 * s + kq, r); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NB; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(r, k0 + kq), b(k0 + kq, r), acc, 0, 0, 0);
    return acc;
}

// Two tiles at once, K a multiple of four: their products alternate, so that the accumulate of one tile does not wait for
// the result of its own previous step (a matrix-core instruction that depends on the one before it waits out its latency).
template <int NB, class FA0, class FB0, class FA1, class FB1>
DEV void mfma_pair_k4(int wl, int K, FA0 a0, FB0 b0, FA1 a1, FB1 b1, mfma_d4 &acc0, mfma_d4 &acc1)
{
    const int r = wl & 15, kq = wl >> 4;
    int k0 = 0;
    for (; k0 + 4 * NB <= K; k0 += 4 * NB) {
        double av0[NB], bv0[NB], av1[NB], bv1[NB];
#pragma unroll
        for (int s = 0; s < NB; s++) {
            av0[s] = a0(r, k0 + 4 * s + kq); bv0[s] = b0(k0 + 4 * s + kq, r);
            av1[s] = a1(r, k0 + 4 * s + kq); bv1[s] = b1(k0 + 4 * s + kq, r);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NB; s++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[s], bv0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[s], bv1[s], acc1, 0, 0, 0);
        }
    }
    for (; k0 < K; k0 += 4) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0(r, k0 + kq), b0(k0 + kq, r), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1(r, k0 + kq), b1(k0 + kq, r), acc1, 0, 0, 0);
    }
}

template <class D> DEV bool factor_tiles_fits(const DevProb &p)
{
    const int nz = D::nz(p);
    return D::kNT == 4 * WAVE && nz >= 16 && nz <= 48 && D::nu(p) <= 16 && p.mreg < 65536;
}

// Wave 0 factorises the PANEL, waves 1 .. 3 (the tile waves) own the tiles.  Per stage, backwards in time:
//   (1) tile waves: PA = P_{t+1} [A B] on the matrix cores, to LDS                                        | barrier
//   (2) tile waves: M = G_t + [A B]' PA into their registers (G_t = P + C' D_t C, prepared one stage ahead, see (3));
//       the input columns of M to LDS (both triangles)                                                    | barrier
//   (3) wave 0: the elimination.  It only ever reads the input columns of M: the panel M[:, nx .. nz) (nz x nu) is
//       factorised by wave 0 ALONE, lane i holding row i in registers, the pivot row broadcast with v_readlane -- no
//       barrier between pivots -- (fixed binaries: identity rows / columns, applied as the panel is loaded).  Out: the
//       multipliers, the reciprocal pivots, mb, and the panel's columns as they stood when they were the pivot.
//       MEANWHILE the tile waves prepare G_{t-1}: the Gram part does not depend on the recursion.            | barrier
//   (4) tile waves: the state block in one step on the matrix cores, P_t = M_xx - L_x A_x' (same operations as
//       pivot by pivot, each entry's subtractions in the same order), to LDS for (1) and to the slab          | barrier
// The LDS form above (dense M in LDS, every entry read, updated and written back per pivot by all threads, one barrier
// per pivot) took 58 k cycles per stage on configs[4]; with one wave per SIMD the cost of a phase is its instruction
// count, and that form spent it on index arithmetic and on barriers.
template <class D, int MQ> DEV int factor_tiles(const DevProb &p, const Lds &S, int lane FSTAMP_ARGS)
{
    FSTAMP_DECL;
    constexpr int NUM = 16; // inputs the panel holds per lane
    constexpr int NDS = 8;  // steps of four dense rows whose D a lane keeps in registers (more dense rows: the batched loads)
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nuc = D::nuc(p), nub = D::nub(p);
    const int nxs = nx * (nx + 1) / 2, lms = LM_STAGE(nx, nu);
    const int wl = lane & 63, wave = __builtin_amdgcn_readfirstlane(lane >> 6), r0 = wl >> 4, c0 = wl & 15; // (wave: uniform, so that everything per tile is scalar)
    const int tn = (nz + 15) >> 4, tx = (nx + 15) >> 4, ntile = tn * (tn + 1) / 2;
    const bool tilew = wave > 0;
#ifdef HMPC_STAMPS
    facc[5] += 1; // (count of factorisations)
#endif
    ldsd *Mf = S.Mm;           // what the other lanes see of the matrix: dense nz x nz
    ldsd *dump = S.red + 39;   // where the entries of a tile that overhang the matrix are written
    // a tile wave's entries: tile qq -> rows row0[qq] + 4 r, column colq[qq]
    int row0[MQ], colq[MQ], trow[MQ], tcol[MQ];
    bool have[MQ];
#pragma unroll
    for (int qq = 0; qq < MQ; qq++) {
        const int q = (wave - 1) + 3 * qq;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
        const int tj = q - ti * (ti + 1) / 2;
        have[qq] = tilew && q < ntile;
        trow[qq] = ti * 16;
        tcol[qq] = tj * 16;
        row0[qq] = ti * 16 + r0;
        colq[qq] = tj * 16 + c0;
    }
    for (int e = lane; e < nx * nx; e += D::kNT) {
        const int i = e / nx, j = e - i * nx;
        if (i >= j) fac_pr<D>(S)[T * nxs + sym(i, j)] = S.PT[e];
    }
    if (lane == 0) S.flag[0] = 0;
    const bool dense8 = S.Cdn && p.ndp <= 4 * NDS;
    // G_t = P + C' D_t C of this lane's tiles (the singleton rows reach the diagonal only; the terminal-set rows, when
    // they take part, belong to the last stage)
    double gq[MQ][4];
    auto gram = [&](int t) {
        const ldsd *Dt = S.e + t * p.mreg; // S.e holds D during the factorisation
        const bool term = S.term_on && t == T - 1;
        double dk[NDS];
        int ck[NDS]; // row offset of dense row 4 s + kq in Cdn (clamped)
        if (dense8) {
#pragma unroll
            for (int s2 = 0; s2 < NDS; s2++) {
                const int k = 4 * s2 + r0, kc = k < p.ndp ? k : p.ndp - 1;
                const double x = Dt[S.drow[kc]];
                dk[s2] = k < p.ndp ? x : 0.0;
                ck[s2] = kc * nz;
            }
        }
#pragma unroll
        for (int qq = 0; qq < MQ; qq++) {
            if (have[qq]) {
                // rows / columns of a tile past nz are read from clamped indices and NOT zeroed: an entry of a product only
                // sees its own row of the first and its own column of the second operand, and those entries are never used
                const int i = trow[qq] + c0, ic = i < nz ? i : nz - 1; // first operand: row l & 15 of the tile's rows
                const int j = colq[qq], jc = j < nz ? j : nz - 1;
                mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
                if (dense8) {
                    double av[NDS], bv[NDS];
#pragma unroll
                    for (int s2 = 0; s2 < NDS; s2++) { av[s2] = S.Cdn[ck[s2] + ic] * dk[s2]; bv[s2] = S.Cdn[ck[s2] + jc]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s2 = 0; s2 < NDS; s2++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[s2], acc, 0, 0, 0);
                } else if (S.Cdn)
                    acc = mfma_tile_batched<4>(wl, p.ndp, [&](int, int k) { return S.Cdn[k * nz + ic] * Dt[S.drow[k]]; },
                                               [&](int k, int) { return S.Cdn[k * nz + jc]; }, acc);
                else
                    acc = mfma_tile_batched<5>(wl, p.mreg, [&](int, int k) { return p.Creg[(size_t)k * nz + ic] * Dt[k]; },
                                               [&](int k, int) { return p.Creg[(size_t)k * nz + jc]; }, acc);
                if (term) {
                    const ldsd *De = S.e + p.Toff;
                    acc = mfma_tile_batched<5>(wl, p.nT, [&](int, int k) { return p.Ct[(size_t)k * nz + ic] * De[k]; },
                                               [&](int k, int) { return p.Ct[(size_t)k * nz + jc]; }, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = row0[qq] + 4 * r, rc = row < nz ? row : nz - 1;
                    double v = acc[r] + S.P[rc * nz + jc];
                    if (S.Cdn && row == j && j < nz) { // diagonal: the singleton rows of this column
                        for (int k = S.sptr[j]; k < S.sptr[j + 1]; k++) {
                            const int sr = S.srow[k];
                            const double c = S.sval[sr];
                            v += c * c * Dt[sr];
                        }
                    }
                    gq[qq][r] = v;
                }
            }
        }
    };
    int bad = 0;
    for (int t = T; t >= 0; t--) {
        const bool stage = t < T; // (t == T: the prologue -- only the Gram part of the last stage, by the code of (3))
        const ldsd *Pn = t == T - 1 ? S.PT : Mf; // P_{t+1}: the terminal Hessian, or the block the previous stage left
        const int pns = t == T - 1 ? nx : nz;
        const ldsi *fx = S.fix + t * nub;
        const auto Lm = fac_lm<D>(S) + t * lms;
        ldsd *Lw;
        if constexpr (D::kBig) Lw = S.Lw;
        else Lw = S.Lm + t * lms;
        // (1) PA = P_{t+1} [A B]  (nx x nz)
        if (tilew && stage) {
            int q = wave - 1;
            if ((nx & 3) == 0) {
                for (; q + 3 < tx * tn; q += 6) { // two tiles at once
                    const int ti0 = q / tn, tj0 = q - ti0 * tn, ti1 = (q + 3) / tn, tj1 = (q + 3) - ti1 * tn;
                    const int i0 = ti0 * 16 + c0, ic0 = i0 < nx ? i0 : nx - 1, j0 = tj0 * 16 + c0, jc0 = j0 < nz ? j0 : nz - 1;
                    const int i1 = ti1 * 16 + c0, ic1 = i1 < nx ? i1 : nx - 1, j1 = tj1 * 16 + c0, jc1 = j1 < nz ? j1 : nz - 1;
                    mfma_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                    mfma_pair_k4<5>(wl, nx, [&](int, int k) { return Pn[ic0 * pns + k]; }, [&](int k, int) { return S.AB[k * AB_STRIDE + jc0]; },
                                    [&](int, int k) { return Pn[ic1 * pns + k]; }, [&](int k, int) { return S.AB[k * AB_STRIDE + jc1]; }, acc0, acc1);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int ii0 = ti0 * 16 + r0 + 4 * r, ii1 = ti1 * 16 + r0 + 4 * r;
                        if (ii0 < nx && j0 < nz) S.PA[ii0 * nz + j0] = acc0[r];
                        if (ii1 < nx && j1 < nz) S.PA[ii1 * nz + j1] = acc1[r];
                    }
                }
            }
            for (; q < tx * tn; q += 3) {
                const int ti = q / tn, tj = q - ti * tn;
                const int i = ti * 16 + c0, ic = i < nx ? i : nx - 1, j = tj * 16 + c0, jc = j < nz ? j : nz - 1;
                mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
                if ((nx & 3) == 0)
                    acc = mfma_tile_k4<5>(wl, nx, [&](int, int k) { return Pn[ic * pns + k]; }, [&](int k, int) { return S.AB[k * AB_STRIDE + jc]; }, acc);
                else
                    acc = mfma_tile_batched<5>(wl, nx, [&](int, int k) { return Pn[ic * pns + k]; },
                                               [&](int k, int) { return S.AB[k * AB_STRIDE + jc]; }, acc);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ii = ti * 16 + r0 + 4 * r;
                    if (ii < nx && j < nz) S.PA[ii * nz + j] = acc[r];
                }
            }
        }
        if (stage) {
            for (int e = lane; e < lms; e += D::kNT) Lw[e] = 0.0;
            lds_barrier();
        }
        FSTAMP(0);
        // (2) the tiles of M, in registers; their input columns to LDS
        double m[MQ][4];
#pragma unroll
        for (int qq = 0; qq < MQ; qq++) {
#pragma unroll
            for (int r = 0; r < 4; r++) m[qq][r] = 0.0;
        }
#pragma unroll
        for (int qp = 0; qp < MQ; qp += 2) { // two tiles at once where the wave has two
            if (stage && have[qp] && have[qp + 1] && (nx & 3) == 0) {
                const int i0 = trow[qp] + c0, ic0 = i0 < nz ? i0 : nz - 1, jc0 = colq[qp] < nz ? colq[qp] : nz - 1;
                const int i1 = trow[qp + 1] + c0, ic1 = i1 < nz ? i1 : nz - 1, jc1 = colq[qp + 1] < nz ? colq[qp + 1] : nz - 1;
                mfma_d4 acc0 = {gq[qp][0], gq[qp][1], gq[qp][2], gq[qp][3]}, acc1 = {gq[qp + 1][0], gq[qp + 1][1], gq[qp + 1][2], gq[qp + 1][3]};
                mfma_pair_k4<5>(wl, nx, [&](int, int k) { return S.AB[k * AB_STRIDE + ic0]; }, [&](int k, int) { return S.PA[k * nz + jc0]; },
                                [&](int, int k) { return S.AB[k * AB_STRIDE + ic1]; }, [&](int k, int) { return S.PA[k * nz + jc1]; }, acc0, acc1);
#pragma unroll
                for (int r = 0; r < 4; r++) { m[qp][r] = acc0[r]; m[qp + 1][r] = acc1[r]; }
            }
        }
#pragma unroll
        for (int qq = 0; qq < MQ; qq++) {
            if (stage && have[qq]) {
                if (!(have[qq & ~1] && have[qq | 1] && (nx & 3) == 0)) {
                const int i = trow[qq] + c0, ic = i < nz ? i : nz - 1, jc = colq[qq] < nz ? colq[qq] : nz - 1;
                mfma_d4 acc = {gq[qq][0], gq[qq][1], gq[qq][2], gq[qq][3]};
                if ((nx & 3) == 0)
                    acc = mfma_tile_k4<5>(wl, nx, [&](int, int k) { return S.AB[k * AB_STRIDE + ic]; }, [&](int k, int) { return S.PA[k * nz + jc]; }, acc);
                else
                    acc = mfma_tile_batched<5>(wl, nx, [&](int, int k) { return S.AB[k * AB_STRIDE + ic]; },
                                               [&](int k, int) { return S.PA[k * nz + jc]; }, acc);
#pragma unroll
                for (int r = 0; r < 4; r++) m[qq][r] = acc[r];
                }
                // (P_{t+1} in Mf was last read in (1), before the barrier every wave has passed; the state block of M stays
                // in registers: only the input columns are read by others)
                // (entries of a tile that overhang the matrix go to a dump slot: no predicate on the stores)
                if (tcol[qq] + 15 >= nx) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = row0[qq] + 4 * r, col = colq[qq];
                        ldsd *w = (row < nz && col < nz) ? Mf + row * nz + col : dump;
                        *w = m[qq][r];
                    }
                }
                if (trow[qq] != tcol[qq] && trow[qq] + 15 >= nx) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = row0[qq] + 4 * r, col = colq[qq];
                        ldsd *w = (row < nz && col < nz) ? Mf + col * nz + row : dump;
                        *w = m[qq][r];
                    }
                }
            }
        }
        if (stage) lds_barrier();
        FSTAMP(1);
        // (3) wave 0: mb and the panel; tile waves: the Gram part of the next stage
        if (!tilew) {
          if (stage) {
            const int i = lane, ic = i < nz ? i : 0, b0 = nx + nuc;
            const int fxv = i < nub ? fx[i < nub ? i : 0] : -1;
            const unsigned long long fm1 = __ballot(fxv >= 0), fone = __ballot(fxv == 1);
            unsigned fixmask = (unsigned)(fm1 << nuc); // bit k: input k is a binary the node fixes (wave uniform)
            if (i < nz) {
                double g = 0;
                for (int b = 0; b < nub; b++)
                    if ((fone >> b) & 1ull) g += Mf[i * nz + b0 + b];
                S.g[t * nz + i] = g;
            }
            const bool rowfixed = i >= nx && i < nz && ((fixmask >> (i - nx)) & 1u);
            // NC: columns of the panel held per lane.  A stage whose binaries are ALL fixed by the node -- the upper stages
            // of every deep node of a search -- only has its continuous inputs to eliminate: the narrow form (eight columns)
            // does a quarter of the broadcast-and-update steps of the full one.
            auto panel = [&](auto nc_, int ncols) {
                constexpr int NC = decltype(nc_)::value;
                double a[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const double x = Mf[ic * nz + nx + (k < ncols ? k : 0)];
                    const bool pres = rowfixed || ((fixmask >> k) & 1u); // prescribed component: identity row / column
                    a[k] = (k < ncols && i < nz) ? (pres ? (i == nx + k ? 1.0 : 0.0) : x) : 0.0;
                }
                // (the bounds and the mask are made opaque per pivot: tested where they are used, two scalar instructions,
                // instead of sixteen lane masks computed ahead of the loop, spilled, and fetched back at every use)
                int nuo = __builtin_amdgcn_readfirstlane(ncols);
                unsigned fmask = __builtin_amdgcn_readfirstlane(fixmask);
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    asm volatile("" : "+s"(nuo), "+s"(fmask));
                    if (j < nuo) {
                        if (i < nz) Mf[i * nz + nx + j] = a[j]; // the column as it stands: second operand of the state block's product
                        if ((fmask >> j) & 1u) { // decoupled unit pivot: the solves skip it as well
                            if (lane == 0) S.dinv[t * nu + j] = 1.0;
                        } else {
                            const double d = bcast(a[j], nx + j);
                            if (!(d > 0.0)) bad = 1;
                            double rinv = __builtin_amdgcn_rcp(d);
                            rinv = rinv * (2.0 - d * rinv); // one Newton step on the hardware reciprocal
                            const bool trailing = i < nx || (i > nx + j && i < nz);
                            const double mval = a[j] * rinv, meff = trailing ? mval : 0.0;
#pragma unroll
                            for (int k = j + 1; k < NC; k++) a[k] -= meff * bcast(a[k], nx + j); // (columns past the last are zero and stay zero)
                            if (trailing) Lw[i < nx ? LM_X(nx, nu, i, j) : LM_U(nx, nu, i - nx, j)] = mval;
                            if (lane == 0) S.dinv[t * nu + j] = rinv;
                        }
                    }
                }
            };
            constexpr int NUMC = 8;
            if (HMPC_NARROW_PANEL && nub > 0 && nuc <= NUMC && __popcll(fm1) == nub) {
                panel(std::integral_constant<int, NUMC>(), nuc);
                if (lane < nub) S.dinv[t * nu + nuc + lane] = 1.0; // (the binaries: unit pivots, their multipliers stay zero)
            } else {
                panel(std::integral_constant<int, NUM>(), nu);
            }
            if (bad && lane == 0) S.flag[0] = 1; // (wave 0 alone sees the pivots)
          }
        } else if (t > 0) {
            gram(t - 1);
        }
        if (!stage) continue;
        lds_barrier();
        FSTAMP(3);
        // (4) state block on the matrix cores, by the waves that hold its tiles
#pragma unroll
        for (int qq = 0; qq < MQ; qq++) {
            if (have[qq] && trow[qq] < nx && tcol[qq] < nx) {
                const int i = trow[qq] + c0, ixc = i < nx ? i : nx - 1, k = colq[qq], kc = k < nx ? k : nx - 1;
                // (the products are subtracted from M in pivot order, four pivots per step: the accumulator starts at M)
                mfma_d4 acc = {m[qq][0], m[qq][1], m[qq][2], m[qq][3]};
                acc = mfma_tile_batched<4>(wl, nu, [&](int, int j) { return -Lw[LM_X(nx, nu, ixc, j)]; },
                                           [&](int j, int) { return Mf[kc * nz + nx + j]; }, acc);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = row0[qq] + 4 * r, col = colq[qq];
                    if (row < nx && col <= row) {
                        const double v = acc[r];
                        Mf[row * nz + col] = v;
                        Mf[col * nz + row] = v;
                        fac_pr<D>(S)[t * nxs + sym(row, col)] = v;
                    }
                }
            }
        }
        if constexpr (D::kBig)
            for (int e = lane; e < lms; e += D::kNT) Lm[e] = Lw[e]; // (written in (3), before the last barrier)
        lds_barrier();
        FSTAMP(4);
    }
    __syncthreads(); // (the slab is read by other threads than wrote it: this barrier covers global memory)
    return (bad || S.flag[0]) ? -1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Same factorisation for the compile-time shapes, entirely in wave 0 and without a barrier inside the
// stage loop.  Per stage:
//   * Gram: lane e < ngram owns one entry (i, j) of C' D C with a nonempty term list (the host numbers
//     those first; at most 64) and sums its terms from fenced batches of LDS loads; entries without
//     terms are constant (M = P there) and written once per factorisation;
//   * W = [A B]' P_{t+1} [A B] in registers: lanes c < nx hold column c of P_{t+1} from the previous
//     stage's elimination, lane c computes y = P_{t+1} [A B](:, c) and then column c of W, every
//     operand of another lane through v_readlane (wave-uniform SGPRs) -- no LDS round trip between
//     two stages;
//   * lane c < nz then owns column c of M = P + C' D C + W.  A pivot step broadcasts the pivot column
//     with v_readlane and every lane updates its own column.  Fixed binaries are prescribed (identity
//     row / column): their pivot steps are arithmetic no-ops, so the elimination needs no branch.
// ---------------------------------------------------------------------------------------------
template <class D> DEV int factor_reg(const DevProb &p, const Lds &S, int lane FSTAMP_ARGS)
{
    constexpr int NX = D::kNX, NU = D::kNU, NUB = D::kNUB, NUC = NU - NUB, NZ = NX + NU, NE = NZ * (NZ + 1) / 2;
    // (nz = 16 fits the row of 16 lanes on paper but comes out wrong -- every node NUMERICAL, found by the first-use check of
    // the compiled kernels on a random MLD with nx = 9, nu = 3 + 4, profiles/r04_sized_shapes.txt; the hosts admit nz <= 15)
    static_assert(NZ <= 16, "the register recursions broadcast inside one row of 16 lanes");
    constexpr int NXS = NX * (NX + 1) / 2, LMS = LM_STAGE(NX, NU), KG = D::kKC, GH = KG / 2;
    const int T = p.T;
    FSTAMP_DECL;
    LANE_OPAQUE(lane);
    if (lane == 0) S.flag[0] = 0;
    if (D::kNW == 1 || lane < WAVE) {
        const int ng = p.ngram;
        const bool has = lane < ng;
        const int ge = has ? lane : 0;
        const int gi = S.ei[ge], gj = S.ej[ge];
        const int gp0 = S.L0.gptr[ge], glen = has ? S.L0.gptr[ge + 1] - gp0 : 0;
        const double pij = S.P[gi * NZ + gj];
        ldsd *TG = S.Mm + NZ * NZ; // dense terminal block of the last stage (when active)
        for (int q = ng + lane; q < NE; q += WAVE) {
            const int i = S.ei[q], j = S.ej[q];
            const double a = S.P[i * NZ + j];
            S.Mm[i * NZ + j] = a;
            S.Mm[j * NZ + i] = a;
        }
        if (S.term_on) {
            // dense terminal block C_T' D C_T: every lane takes terminal rows lane, lane + 64, ... (coefficients in
            // registers for the moment, loads independent), every entry is one sum over the wave -- as a loop over the
            // ~100 rows per entry (three dependent loads a term) this block took longer than the rest of the factorisation
            const ldsd *Dt = S.e + p.Toff;
            if (p.nT == 0) { // (a problem without terminal set runs its only solve with term_on set: an empty block)
                for (int q = lane; q < NZ * NZ; q += WAVE) TG[q] = 0.0;
            }
            for (int r0 = 0; r0 < p.nT; r0 += WAVE) { // (two trips for the cart-pole's 102 facets)
                const int row = r0 + lane;
                const bool ok = row < p.nT;
                const int rc = ok ? row : 0;
                const double dr = ok ? Dt[rc] : 0.0;
                const double *c = p.Ct + (size_t)rc * NZ;
                double cr[NZ];
#pragma unroll
                for (int j = 0; j < NZ; j++) cr[j] = c[j];
#pragma unroll
                for (int i = 0; i < NZ; i++) {
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        const double a = wave_sum(cr[i] * cr[j] * dr);
                        if (lane == 0) {
                            const double v = (r0 ? TG[i * NZ + j] : 0.0) + a;
                            TG[i * NZ + j] = v;
                            TG[j * NZ + i] = v;
                        }
                    }
                }
            }
        }
        double ABc[NX], pn[NX];
#pragma unroll
        for (int l = 0; l < NX; l++) ABc[l] = lane < NZ ? S.AB[l * NZ + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < NX; i++) pn[i] = lane < NX ? S.PT[i * NX + lane] : 0.0;
        if (lane < NX) {
#pragma unroll
            for (int i = 0; i < NX; i++)
                if (i >= lane) S.Pr[T * NXS + sym(i, lane)] = pn[i];
        }
        int bad = 0;
        for (int t = T - 1; t >= 0; t--) {
            const ldsi *fx = S.fix + t * NUB;
            ldsd *Lm = S.Lm + t * LMS;
            int fxv[NUB];
#pragma unroll
            for (int b = 0; b < NUB; b++) fxv[b] = fx[b];
            // (1) this lane's Gram entry
            double g0 = pij, g1 = 0.0;
            {
                const ldsd *Dt = S.e + t * p.mreg; // S.e holds D during the factorisation
#pragma unroll
                for (int hb = 0; hb < KG; hb += GH) {
                    int idx[GH];
                    double gv[GH], dv[GH];
#pragma unroll
                    for (int q = 0; q < GH; q++) {
                        const int qq = gp0 + (hb + q < glen ? hb + q : 0);
                        idx[q] = S.L0.grow[qq];
                        gv[q] = S.L0.gval[qq];
                    }
                    HMPC_FENCE();
#pragma unroll
                    for (int q = 0; q < GH; q++) { HMPC_CHK(idx[q] >= 0 && idx[q] < p.mreg, 2); dv[q] = Dt[idx[q]]; }
                    HMPC_FENCE();
#pragma unroll
                    for (int q = 0; q < GH; q++) {
                        const double term = hb + q < glen ? gv[q] * dv[q] : 0.0;
                        if (q & 1) g1 += term;
                        else g0 += term;
                    }
                }
            }
            if (has) {
                S.Mm[gi * NZ + gj] = g0 + g1;
                S.Mm[gj * NZ + gi] = g0 + g1;
            }
            // entries 64 .. 127 of C' D C with a term list (nz = 15 with dense state rows has ~90) take a second trip; their
            // per-lane constants are fetched per stage instead of being held.  (The cart-pole shapes have at most 64 such
            // entries: a kernel compiled with the problem's sizes folds the test away.)
            if (ng > WAVE) {
                const bool has2 = WAVE + lane < ng;
                const int ge2 = has2 ? WAVE + lane : 0;
                const int gi2 = S.ei[ge2], gj2 = S.ej[ge2];
                const int gq0 = S.L0.gptr[ge2], glen2 = has2 ? S.L0.gptr[ge2 + 1] - gq0 : 0;
                const ldsd *Dt = S.e + t * p.mreg;
                double h0 = S.P[gi2 * NZ + gj2], h1 = 0.0;
#pragma unroll
                for (int q = 0; q < KG; q++) {
                    const int qq = gq0 + (q < glen2 ? q : 0);
                    const double term = q < glen2 ? S.L0.gval[qq] * Dt[S.L0.grow[qq]] : 0.0;
                    if (q & 1) h1 += term;
                    else h0 += term;
                }
                if (has2) {
                    S.Mm[gi2 * NZ + gj2] = h0 + h1;
                    S.Mm[gj2 * NZ + gi2] = h0 + h1;
                }
            }
            // (2) column `lane` of W = [A B]' P_{t+1} [A B]
            double y[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; l++) a += bcast16<D>(pn[l < i ? l : i], l < i ? i : l) * ABc[l]; // P_{t+1}(i, l), symmetric
                y[i] = a;
            }
            double col[NZ];
#pragma unroll
            for (int i = 0; i < NZ; i++) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; l++) a += S.AB[l * NZ + i] * y[l]; // (a uniform LDS read: as a cross-lane broadcast the 44
                                                                         // constants are hoisted out of every loop and live in registers for good)
                col[i] = a;
            }
            FSTAMP(0);
            // (3) + P + C' D C (this wave wrote it: LDS operations of one wave complete in order)
            if (lane < NZ) {
#pragma unroll
                for (int i = 0; i < NZ; i++) col[i] += S.Mm[i * NZ + lane];
                if (S.term_on && t == T - 1) {
#pragma unroll
                    for (int i = 0; i < NZ; i++) col[i] += TG[i * NZ + lane];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NZ; i++) col[i] = 0.0;
            }
            FSTAMP(1);
            int nfixed = 0;
#pragma unroll
            for (int b = 0; b < NUB; b++) nfixed += fxv[b] >= 0;
            if (nfixed) {
                // columns of binaries fixed to one (for the constant direction), before prescribing
                double mbv[NZ];
#pragma unroll
                for (int i = 0; i < NZ; i++) mbv[i] = 0.0;
#pragma unroll
                for (int b = 0; b < NUB; b++)
                    if (fxv[b] == 1) {
#pragma unroll
                        for (int i = 0; i < NZ; i++) mbv[i] += bcast16<D>(col[i], NX + NUC + b);
                    }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NZ; i++) S.g[t * NZ + i] = mbv[i];
                }
#pragma unroll
                for (int b = 0; b < NUB; b++)
                    if (fxv[b] >= 0) {
                        const int cb = NX + NUC + b;
#pragma unroll
                        for (int i = 0; i < NZ; i++)
                            col[i] = (lane == cb) ? (i == cb ? 1.0 : 0.0) : (i == cb ? 0.0 : col[i]);
                    }
            } else if (lane < NZ) {
                S.g[t * NZ + lane] = 0.0;
            }
            FSTAMP(2);
            // The multiplier of row i at pivot j is M(i, pj) / d; by symmetry lane i holds it as
            // col[pj] / d, so every lane keeps the multipliers of its own row and stores them once.
            double myrow[NU], dinv[NU];
            auto pivot = [&](const int j) {
                const int pj = NX + j;
                const double d = bcast16<D>(col[pj], pj); // 1 at a prescribed pivot: the step below changes nothing
                if (!(d > 0.0)) bad = 1;
                double rinv = __builtin_amdgcn_rcp(d);
                rinv = rinv * (2.0 - d * rinv);
                dinv[j] = rinv;
                const double cr = col[pj] * rinv;
                myrow[j] = (lane < NX || lane > pj) ? cr : 0.0;
#pragma unroll
                for (int i = 0; i < NZ; i++) {
                    if (i < NX || i > pj) col[i] -= bcast16<D>(col[i], pj) * cr; // rows still to be reduced
                }
            };
#pragma unroll
            for (int j = 0; j < NUC; j++) pivot(j);
            // a stage whose binaries are all fixed (most stages of a deep node): their pivot steps are no-ops
            if ((S.fullfix >> t) & 1ull) {
#pragma unroll
                for (int j = NUC; j < NU; j++) { myrow[j] = 0.0; dinv[j] = 1.0; }
            } else {
#pragma unroll
                for (int j = NUC; j < NU; j++) pivot(j);
            }
            if (lane < NZ) { // rows of the states: NU entries; row i of the inputs: its i entries (zeros at skipped pivots)
                ldsd *row = Lm + (lane < NX ? lane * NU : LM_U(NX, NU, lane - NX, 0));
#pragma unroll
                for (int j = 0; j < NU; j++)
                    if (lane < NX || j < lane - NX) row[j] = myrow[j];
            }
            if (lane < NU) {
                double dv = 1.0;
#pragma unroll
                for (int j = 0; j < NU; j++) dv = (lane == j) ? dinv[j] : dv;
                S.dinv[t * NU + lane] = dv;
            }
            FSTAMP(3);
#pragma unroll
            for (int i = 0; i < NX; i++) pn[i] = col[i];
            if (lane < NX) {
#pragma unroll
                for (int i = 0; i < NX; i++)
                    if (i >= lane) S.Pr[t * NXS + sym(i, lane)] = col[i];
            }
            FSTAMP(4);
        }
        if (lane == 0 && bad) S.flag[0] = 1;
    }
    __syncthreads();
    return S.flag[0] ? -1 : 0;
}

// ---------------------------------------------------------------------------------------------
// One KKT solve  K d = rhs  by a backward / forward Riccati sweep.
//   rhs_d  : gs * gsrc (n entries; gsrc may be null; gsrc may alias S.g)   stage gradients
//   x_0    : x0 if usex0 else 0                                            prescribed initial state
//   cdyn   : cs * csrc (T*nx; csrc may be null)                            dynamics offsets
//   useb   : fixed binaries take their value v (constant direction, S.g holds the factorisation's
//            mb on entry) or 0
//   S.e    : D .* rhs_c on entry ; the multiplier step dz on exit
// Generic form: every phase is spread over the threads, a barrier after each; the substitutions take
// one barrier per pivot.
// ---------------------------------------------------------------------------------------------
// The two sweeps of a solve for the run-time-sized kernel when a stage vector fits one wave (nz <= 64): wave 0 runs the
// recursion with lane j holding component j and v_readlane broadcasts -- no workgroup barrier per pivot (the barrier
// form below pays one per pivot and, in the streaming form, a global-memory latency on top).  In the streaming form
// the stage's multipliers are staged in LDS: the other waves fetch the next stage's block into registers while wave 0
// works on the current one.  Skipped pivots (fixed binaries) have zero multipliers: their steps are no-ops.
// On entry: S.g = stage gradients, S.pv[T] set.  On exit: dw (x and u), S.pv (p_t) as the barrier form leaves them.
template <class D>
DEV void kkt_sweeps_wave(const DevProb &p, const Lds &S, int lane, bool usex0, const ldsd *csrc, double cs, bool useb, ldsd *dw FSTAMP_ARGS)
{
    FSTAMP_DECL;
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nuc = D::nuc(p), nub = D::nub(p);
    const int nxs = nx * (nx + 1) / 2, lms = LM_STAGE(nx, nu);
    const bool w0 = lane < WAVE;
    // S.pv[t] <- P_{t+1} c_t: the part of q_t = p_{t+1} + P_{t+1} c_t that does not depend on the recursion
    for (int o = lane; o < T * nx; o += D::kNT) {
        const int t = o / nx, i = o - t * nx;
        double a = 0.0;
        if (csrc) {
            const auto Pn = fac_pr<D>(S) + (t + 1) * nxs;
            int l = 0;
            for (; l + 4 <= nx; l += 4) { // (four loads of the slab in flight: a load inside the sum that uses it waits for itself)
                const double p0 = Pn[sym(i, l)], p1 = Pn[sym(i, l + 1)], p2 = Pn[sym(i, l + 2)], p3 = Pn[sym(i, l + 3)];
                a += p0 * (cs * csrc[t * nx + l]) + p1 * (cs * csrc[t * nx + l + 1]) + p2 * (cs * csrc[t * nx + l + 2]) + p3 * (cs * csrc[t * nx + l + 3]);
            }
            for (; l < nx; l++) a += Pn[sym(i, l)] * (cs * csrc[t * nx + l]);
        }
        S.pv[o] = a;
    }
    double pvr = 0.0; // lane i < nx of wave 0: p_{t+1}[i]
    double xr = 0.0;  //                        x_t[i] (forward sweep)
    // The recursions take their coefficients (a column of [A B], a row / column of the multipliers) from LDS in blocks of
    // eight: all loads of a block are in flight before its first product, indices past the end are clamped and their
    // coefficients zeroed -- a load inside the step that uses it puts one LDS round trip on every step of the chain
    // (measured: 5.7 k cycles per stage and sweep on configs[4], against 1.5 k in this form).
    constexpr int BK = 8;
    // one stage of the backward sweep (wave 0; Lm: the stage's multipliers in LDS)
    auto backward = [&](int t, const ldsd *Lm) {
        const ldsi *fx = S.fix + t * nub;
        const int j = lane;
        const double qv = lane < nx ? pvr + S.pv[t * nx + lane] : 0.0;
        const int jc = j < nz ? j : 0;
        double v = j < nz ? S.g[t * nz + jc] : 0.0, v1 = 0.0;
        for (int l0 = 0; l0 < nx; l0 += BK) {
            double c[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int l = l0 + i;
                const double x = S.AB[(l < nx ? l : nx - 1) * AB_STRIDE + jc];
                c[i] = (l < nx && j < nz) ? x : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int l = l0 + i;
                const double q = bcast(qv, l < WAVE ? l : WAVE - 1);
                if (i & 1) v1 += c[i] * q;
                else v += c[i] * q;
            }
        }
        v += v1;
        if (j >= nx + nuc && j < nz) {
            const int f = fx[j - nx - nuc];
            if (f >= 0) v = (useb && f == 1) ? -1.0 : 0.0;
        }
        // forward substitution: the factorisation's row operations applied to the vector
        const int rowoff = j < nx ? LM_X(nx, nu, j, 0) : j < nz ? LM_U(nx, nu, j - nx, 0) : 0;
        const int rowlen = j < nx ? nu : j < nz ? j - nx : 0;
        for (int j0 = 0; j0 < nu; j0 += BK) {
            double c[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int jj = j0 + i;
                const double x = Lm[rowoff + (jj < rowlen ? jj : 0)];
                c[i] = jj < rowlen ? x : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int src = nx + j0 + i;
                v -= c[i] * bcast(v, src < WAVE ? src : WAVE - 1);
            }
        }
        if (j >= nx && j < nz) dw[t * nz + j] = v; // y = L_u^{-1} m_u, parked in the input slots
        pvr = v;
        if (j < nx) S.pv[t * nx + j] = v;
    };
    // one stage of the forward sweep
    auto forward = [&](int t, const ldsd *Lm) {
        const int j = lane;
        // c = L_x' x + dinv .* y ; x_next = c_dyn + A x (+ B u below)
        double cur = j < nu ? S.dinv[t * nu + (j < nu ? j : 0)] * dw[t * nz + nx + (j < nu ? j : 0)] : 0.0;
        double xn = (csrc && j < nx) ? cs * csrc[t * nx + j] : 0.0;
        const int ju = j < nu ? j : 0, jx = j < nx ? j : 0;
        for (int l0 = 0; l0 < nx; l0 += BK) {
            double cl[BK], ca[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int l = l0 + i, lc = l < nx ? l : nx - 1;
                const double x = Lm[LM_X(nx, nu, lc, ju)], y = S.AB[jx * AB_STRIDE + lc];
                cl[i] = (l < nx && j < nu) ? x : 0.0;
                ca[i] = (l < nx && j < nx) ? y : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int l = l0 + i;
                const double xl = bcast(xr, l < WAVE ? l : WAVE - 1);
                cur += cl[i] * xl;
                xn += ca[i] * xl;
            }
        }
        // back substitution with L_u' ; u_jj = -(value of lane jj once its turn has come)
        for (int j0 = (nu - 1) / BK * BK; j0 >= 0; j0 -= BK) {
            double cl[BK], ca[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int jj = j0 + i, jjc = jj < nu ? jj : nu - 1;
                const double x = Lm[LM_U(nx, nu, (j < jjc ? jjc : 1), (j < jjc ? j : 0))], y = S.AB[jx * AB_STRIDE + nx + jjc];
                cl[i] = (jj < nu && j < jj) ? x : 0.0;
                ca[i] = (jj < nu && j < nx) ? y : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = BK - 1; i >= 0; i--) {
                const int jj = j0 + i;
                const double uj = bcast(cur, jj < WAVE ? jj : WAVE - 1);
                cur -= cl[i] * uj;
                xn -= ca[i] * uj;
            }
        }
        if (j < nx) dw[t * nz + j] = xr;
        if (j < nu) dw[t * nz + nx + j] = -cur;
        xr = xn;
    };
    if constexpr (!D::kBig) {
        __syncthreads();
        if (w0) {
            if (lane < nx) pvr = S.pv[T * nx + lane];
            for (int t = T - 1; t >= 0; t--) backward(t, S.Lm + t * lms);
        }
        __syncthreads();
        if (w0) {
            xr = (lane < nx && usex0) ? S.x0[lane] : 0.0;
            for (int t = 0; t < T; t++) forward(t, S.Lm + t * lms);
            if (lane < nx) dw[T * nz + lane] = xr;
        }
        __syncthreads();
    } else {
        // Streaming form: the multipliers live in the global slab.  The other waves (one wave per node: the wave itself)
        // bring them to LDS in CHUNKS of `ring` stages, two chunk buffers: while wave 0 runs the recursion over chunk g
        // the loads of chunk g + 1 are in flight, one workgroup barrier per chunk.  Chunks 0 .. nch-1 walk the horizon
        // backwards, chunks nch .. 2 nch - 1 forwards.
        // In LDS a stage block is PADDED: row i (component i of the stage vector) holds its multipliers in nup columns,
        // zero beyond its entries, and zero rows follow the last component; [A B] carries a zero column and zero rows
        // likewise.  Wave 0 -- the only wave at work, its instruction count IS the time of a solve -- then runs fixed
        // four-wide steps with no bound tests, clamps or selects: a term past the end multiplies a zero (measured on
        // configs[4]: 7 k cycles per stage and sweep with tested / clamped loads, 3.3 k in this form).
        constexpr int RMAX = 2, STG = 8; // stages per chunk (DevProb::ring <= RMAX); doubles per fetching thread and stage: lms <= 8 * fetchers (checked by the caller)
        const int R = p.ring, nch = (T + R - 1) / R;
        const int fetchers = D::kNT > WAVE ? D::kNT - WAVE : D::kNT, fid = D::kNT > WAVE ? lane - WAVE : lane;
        const bool helper = fid >= 0;
        const int nup = S.nup, slot = S.lrows * nup, abst = S.abst;
        const int nxr = (nx + 3) / 4 * 4, nur = (nu + 3) / 4 * 4;
        double stage_reg[RMAX][STG];
        int dsto[STG]; // packed entry e = fid + q fetchers of a stage block -> its place in the padded block
#pragma unroll
        for (int q = 0; q < STG; q++) {
            int e = fid + q * fetchers;
            e = e < lms && e >= 0 ? e : lms - 1;
            int row, col;
            if (e < nx * nu) {
                row = e / nu;
                col = e - row * nu;
            } else {
                const int k = e - nx * nu; // L_u packed by rows: row i holds its i entries
                int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
                while (i * (i - 1) / 2 > k) i--;
                while ((i + 1) * i / 2 <= k) i++;
                row = nx + i;
                col = k - i * (i - 1) / 2;
            }
            dsto[q] = row * nup + col;
        }
        auto stage_of = [&](int g, int sidx) { // stage of slot sidx of chunk g, -1: none
            const int t = g < nch ? T - 1 - g * R - sidx : (g - nch) * R + sidx;
            return (t >= 0 && t < T) ? t : -1;
        };
        auto fetch = [&](int g) {
#pragma unroll
            for (int sidx = 0; sidx < RMAX; sidx++) {
                const int t = sidx < R ? stage_of(g, sidx) : -1;
                if (t >= 0) {
                    const double *src = S.LmG + (size_t)t * lms;
#pragma unroll
                    for (int q = 0; q < STG; q++) {
                        // (clamped, not predicated: a load under a lane predicate sits in a branch of its own and waits for
                        // itself -- one slab latency per load instead of one per chunk)
                        const int e = fid + q * fetchers;
                        if (q * fetchers < lms) stage_reg[sidx][q] = src[e < lms ? e : lms - 1];
                    }
                }
            }
        };
        auto commit = [&](int g) {
            ldsd *dst = S.Ls + (g & 1) * R * slot;
#pragma unroll
            for (int sidx = 0; sidx < RMAX; sidx++) {
                const int t = sidx < R ? stage_of(g, sidx) : -1;
                if (t >= 0) {
#pragma unroll
                    for (int q = 0; q < STG; q++) {
                        const int e = fid + q * fetchers;
                        if (e < lms) dst[sidx * slot + dsto[q]] = stage_reg[sidx][q];
                    }
                }
            }
        };
        // one stage of the backward sweep on a padded block
        auto backward_p = [&](int t, const ldsd *Lp) {
            const ldsi *fx = S.fix + t * nub;
            const int j = lane, jc = j < nz ? j : nz; // (column nz of [A B] and row nz of the block are zero)
            const double qv = lane < nx ? pvr + S.pv[t * nx + lane] : 0.0;
            double v0 = j < nz ? S.g[t * nz + (j < nz ? j : 0)] : 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
            // (q goes through LDS -- one write, then uniform reads that every lane shares -- instead of two v_readlane and a
            // wait state per term; the chain of the substitution below cannot: its terms depend on each other)
            if (lane < nxr) S.q[lane] = qv; // (lanes nx .. nxr-1: zero)
            const ldsd *ab = S.AB + jc;
            for (int l = 0; l < nxr; l += 4) { // (rows nx .. nxr-1 of [A B] are zero rows)
                const double c0 = ab[0], c1 = ab[abst], c2 = ab[2 * abst], c3 = ab[3 * abst];
                const double q0 = S.q[l], q1 = S.q[l + 1], q2 = S.q[l + 2], q3 = S.q[l + 3];
                ab += 4 * abst;
                v0 += c0 * q0;
                v1 += c1 * q1;
                v2 += c2 * q2;
                v3 += c3 * q3;
            }
            double v = (v0 + v1) + (v2 + v3);
            if (j >= nx + nuc && j < nz) {
                const int f = fx[j - nx - nuc];
                if (f >= 0) v = (useb && f == 1) ? -1.0 : 0.0;
            }
            // forward substitution: the factorisation's row operations applied to the vector
            // (a stage whose binaries are all fixed: their pivots were skipped, their multipliers are zero -- the chain ends
            // with the continuous inputs)
            const ldsd *lr = Lp + jc * nup;
            const int nsub = (HMPC_NARROW_PANEL && t < 64 && ((S.fullfix >> t) & 1ull)) ? (nuc + 3) / 4 * 4 : nur;
            for (int jj = 0; jj < nsub; jj += 4) {
                const double c0 = lr[jj], c1 = lr[jj + 1], c2 = lr[jj + 2], c3 = lr[jj + 3];
                const int s0 = nx + jj;
                v -= c0 * bcast(v, s0 < WAVE ? s0 : WAVE - 1);
                v -= c1 * bcast(v, s0 + 1 < WAVE ? s0 + 1 : WAVE - 1);
                v -= c2 * bcast(v, s0 + 2 < WAVE ? s0 + 2 : WAVE - 1);
                v -= c3 * bcast(v, s0 + 3 < WAVE ? s0 + 3 : WAVE - 1);
            }
            if (j >= nx && j < nz) dw[t * nz + j] = v; // y = L_u^{-1} m_u, parked in the input slots
            pvr = v;
            if (j < nx) S.pv[t * nx + j] = v;
        };
        // one stage of the forward sweep on a padded block
        auto forward_p = [&](int t, const ldsd *Lp) {
            const int j = lane, jr = j < nx ? j : nx, ju = j < nu ? j : nu; // (row nx of [A B] and column nu of the block are zero)
            // c = L_x' x + dinv .* y ; x_next = c_dyn + A x (+ B u below)
            double cur = j < nu ? S.dinv[t * nu + (j < nu ? j : 0)] * dw[t * nz + nx + (j < nu ? j : 0)] : 0.0;
            double xn = (csrc && j < nx) ? cs * csrc[t * nx + (j < nx ? j : 0)] : 0.0;
            const ldsd *lc = Lp + ju, *ar = S.AB + jr * abst;
            if (lane < nxr) S.q[lane] = xr; // (x through LDS, as q in the backward sweep; lanes nx .. hold zero)
            for (int l = 0; l < nxr; l += 4) { // (entries nx .. of x are zero: what a term past nx multiplies does not matter)
                const double a0 = ar[l], a1 = ar[l + 1], a2 = ar[l + 2], a3 = ar[l + 3];
                const double c0 = lc[0], c1 = lc[nup], c2 = lc[2 * nup], c3 = lc[3 * nup];
                lc += 4 * nup;
                const double x0 = S.q[l], x1 = S.q[l + 1], x2 = S.q[l + 2], x3 = S.q[l + 3];
                cur += c0 * x0; xn += a0 * x0;
                cur += c1 * x1; xn += a1 * x1;
                cur += c2 * x2; xn += a2 * x2;
                cur += c3 * x3; xn += a3 * x3;
            }
            // back substitution with L_u' ; u_jj = -(value of lane jj once its turn has come)
            const ldsd *lu = Lp + (nx + nur) * nup + ju, *au = ar + nx + nur;
            for (int jj = nur - 4; jj >= 0; jj -= 4) {
                lu -= 4 * nup;
                au -= 4;
                const double c3 = lu[3 * nup], c2 = lu[2 * nup], c1 = lu[nup], c0 = lu[0];
                const double a3 = au[3], a2 = au[2], a1 = au[1], a0 = au[0];
                double uj = bcast(cur, jj + 3);
                cur -= c3 * uj; xn -= a3 * uj;
                uj = bcast(cur, jj + 2);
                cur -= c2 * uj; xn -= a2 * uj;
                uj = bcast(cur, jj + 1);
                cur -= c1 * uj; xn -= a1 * uj;
                uj = bcast(cur, jj);
                cur -= c0 * uj; xn -= a0 * uj;
            }
            if (j < nx) dw[t * nz + j] = xr;
            if (j < nu) dw[t * nz + nx + j] = -cur;
            xr = xn;
        };
        FSTAMP(11);
#ifdef HMPC_STAMPS
        facc[15] += 1; // (count of solves)
#endif
        if (helper) { fetch(0); commit(0); }
        __syncthreads();
        FSTAMP(12);
        if (w0 && lane < nx) pvr = S.pv[T * nx + lane];
        for (int g = 0; g < 2 * nch; g++) {
            const bool more = g + 1 < 2 * nch;
            if (helper && more) fetch(g + 1);
            if (w0) {
                if (g == nch) xr = (lane < nx && usex0) ? S.x0[lane] : 0.0;
                const ldsd *buf = S.Ls + (g & 1) * R * slot;
                for (int sidx = 0; sidx < R; sidx++) {
                    const int t = stage_of(g, sidx);
                    if (t < 0) break;
                    if (g < nch) backward_p(t, buf + sidx * slot);
                    else forward_p(t, buf + sidx * slot);
                }
            }
            FSTAMP(13);
            if (helper && more) commit(g + 1);
            __syncthreads();
            FSTAMP(14);
        }
        if (w0 && lane < nx) dw[T * nz + lane] = xr;
        __syncthreads();
    }
}

template <class D, int RS, class RM>
DEV void kkt_solve(const DevProb &p, const Lds &S, Rows<RS> &R, const RM &rm, int lane, const ldsd *gsrc, double gs, bool usex0,
                   const ldsd *csrc, double cs, bool useb, ldsd *dw, ldsd *dlam, ldsd *dnuf FSTAMP_ARGS)
{
    FSTAMP_DECL;
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nuc = D::nuc(p), nub = D::nub(p);
    const int nxs = nx * (nx + 1) / 2, lms = LM_STAGE(nx, nu);
    bool gdone = false;
    if constexpr (D::kBig && D::kNT == 4 * WAVE) {
        if (S.Cdn && p.ndp <= 32) {
            // C' e of all stages at once on the matrix cores: (nz x dense rows) (dense rows x T), 16 x 16 tiles over the four
            // waves, the rows of e gathered through the dense rows' index; the singleton rows (one term each) and the
            // terminal-set rows are added by the lane that receives the entry.  (As a sum per entry -- 32 gathered terms,
            // four entries per thread -- this was 21 k cycles of every solve.)
            const int wl = lane & 63, wave = __builtin_amdgcn_readfirstlane(lane >> 6), r0 = wl >> 4, c0 = wl & 15;
            const int tn = (nz + 15) >> 4, tt = (T + 15) >> 4;
            for (int q = wave; q < tn * tt; q += D::kNW) {
                const int ti = q / tt, tj = q - ti * tt;
                const int i = ti * 16 + c0, ic = i < nz ? i : nz - 1; // first operand: column i of the dense rows
                const int tcol = tj * 16 + c0, tc = tcol < T ? tcol : T - 1;
                const ldsd *eb = S.e + tc * p.mreg;
                double av[8], bv[8];
#pragma unroll
                for (int s2 = 0; s2 < 8; s2++) {
                    const int k = 4 * s2 + r0, kc = k < p.ndp ? k : p.ndp - 1;
                    const double x = S.Cdn[kc * nz + ic], y = eb[S.drow[kc]];
                    av[s2] = k < p.ndp ? x : 0.0;
                    bv[s2] = y;
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < 8; s2++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[s2], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = ti * 16 + r0 + 4 * r;
                    if (j < nz && tcol < T) {
                        const int o = tcol * nz + j;
                        const ldsd *er = S.e + tcol * p.mreg;
                        double a = (gsrc ? gs * gsrc[o] : 0.0) + acc[r];
                        for (int k = S.sptr[j]; k < S.sptr[j + 1]; k++) { // singleton rows of this column
                            const int sr = S.srow[k];
                            a += S.sval[sr] * er[sr];
                        }
                        if (S.term_on && tcol == T - 1) {
                            const ldsd *vt = S.e + p.Toff;
                            for (int k = 0; k < p.nT; k++) a += p.Ct[(size_t)k * nz + j] * vt[k];
                        }
                        S.g[o] = (useb ? S.g[o] : 0.0) - a;
                    }
                }
            }
            gdone = true;
        }
    }
    if (!gdone)
    for (int o = lane; o < T * nz; o += D::kNT) {
        const int t = o / nz, j = o - t * nz;
        const double a = (gsrc ? gs * gsrc[o] : 0.0) + ccol_dot<D>(p, S, t, j, S.e);
        S.g[o] = (useb ? S.g[o] : 0.0) - a;
    }
    for (int j = lane; j < nx; j += D::kNT) S.pv[T * nx + j] = -(gsrc ? gs * gsrc[T * nz + j] : 0.0);
    __syncthreads();
    const bool wave_sweeps = nz <= WAVE && (!D::kBig || lms <= 8 * (D::kNT > WAVE ? D::kNT - WAVE : D::kNT));
    FSTAMP(6);
    if (wave_sweeps) kkt_sweeps_wave<D>(p, S, lane, usex0, csrc, cs, useb, dw FSTAMP_PASS);
    FSTAMP(7);
    // backward sweep (barrier form: stage vectors wider than a wave)
    for (int t = wave_sweeps ? -1 : T - 1; t >= 0; t--) {
        const auto Lm = fac_lm<D>(S) + t * lms;
        const ldsi *fx = S.fix + t * nub;
        const ldsd *qv = S.pv + (t + 1) * nx;
        if (csrc) {
            const auto Pn = fac_pr<D>(S) + (t + 1) * nxs;
            for (int i = lane; i < nx; i += D::kNT) {
                double a = S.pv[(t + 1) * nx + i];
                for (int l = 0; l < nx; l++) a += Pn[sym(i, l)] * cs * csrc[t * nx + l];
                S.q[i] = a;
            }
            __syncthreads();
            qv = S.q;
        }
        for (int j = lane; j < nz; j += D::kNT) {
            double a = S.g[t * nz + j];
            for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + j] * qv[l];
            if (j >= nx + nuc) {
                const int f = fx[j - nx - nuc];
                if (f >= 0) a = (useb && f == 1) ? -1.0 : 0.0;
            }
            S.mv[j] = a;
        }
        __syncthreads();
        // forward substitution = the row operations of the factorisation applied to the vector
        for (int j = 0; j < nu; j++) {
            if (j >= nuc && fx[j - nuc] >= 0) continue;
            const double yj = S.mv[nx + j];
            for (int i = lane; i < nz; i += D::kNT) {
                if (i < nx) S.mv[i] -= Lm[LM_X(nx, nu, i, j)] * yj;
                else if (i > nx + j) S.mv[i] -= Lm[LM_U(nx, nu, i - nx, j)] * yj;
            }
            __syncthreads();
        }
        for (int i = lane; i < nz; i += D::kNT) {
            if (i < nx) S.pv[t * nx + i] = S.mv[i];
            else dw[t * nz + i] = S.mv[i]; // y = L_u^{-1} m_u, parked in the input slots until the forward sweep
        }
        __syncthreads();
    }
    // forward sweep
    if (!wave_sweeps) {
        for (int i = lane; i < nx; i += D::kNT) dw[i] = usex0 ? S.x0[i] : 0.0;
        __syncthreads();
    }
    for (int t = wave_sweeps ? T : 0; t < T; t++) {
        const auto Lm = fac_lm<D>(S) + t * lms;
        const ldsi *fx = S.fix + t * nub;
        const ldsd *x = dw + t * nz;
        // c = L_x' x + dinv .* y ; then back substitution with L_u', u = -(result)
        for (int j = lane; j < nu; j += D::kNT) {
            double a = S.dinv[t * nu + j] * dw[t * nz + nx + j];
            const bool fixed = j >= nuc && fx[j - nuc] >= 0;
            if (!fixed)
                for (int l = 0; l < nx; l++) a += Lm[LM_X(nx, nu, l, j)] * x[l];
            S.mv[j] = a;
        }
        __syncthreads();
        for (int j = nu - 1; j >= 1; j--) {
            const double uj = S.mv[j];
            for (int i = lane; i < j; i += D::kNT) {
                const bool fixed = i >= nuc && fx[i - nuc] >= 0;
                if (!fixed) S.mv[i] -= Lm[LM_U(nx, nu, j, i)] * uj;
            }
            __syncthreads();
        }
        for (int i = lane; i < nu; i += D::kNT) dw[t * nz + nx + i] = -S.mv[i];
        __syncthreads();
        for (int i = lane; i < nx; i += D::kNT) {
            double a = csrc ? cs * csrc[t * nx + i] : 0.0;
            for (int l = 0; l < nz; l++) a += S.AB[i * AB_STRIDE + l] * dw[t * nz + l];
            dw[(t + 1) * nz + i] = a;
        }
        __syncthreads();
    }
    // equality multipliers lam_t = -(P_t x_t + p_t) ; dz = D (C dw) - e
    for (int o = lane; o < (T + 1) * nx; o += D::kNT) {
        const int t = o / nx, i = o - t * nx;
        double a = S.pv[o];
        const auto Pt = fac_pr<D>(S) + t * nxs;
        int l = 0;
        for (; l + 4 <= nx; l += 4) { // (as in the prepass of the sweeps: four loads of the slab in flight)
            const double p0 = Pt[sym(i, l)], p1 = Pt[sym(i, l + 1)], p2 = Pt[sym(i, l + 2)], p3 = Pt[sym(i, l + 3)];
            a += p0 * dw[t * nz + l] + p1 * dw[t * nz + l + 1] + p2 * dw[t * nz + l + 2] + p3 * dw[t * nz + l + 3];
        }
        for (; l < nx; l++) a += Pt[sym(i, l)] * dw[t * nz + l];
        dlam[o] = -a;
    }
    FSTAMP(8);
    {
        const int nslot = RS > 0 ? RS : p.Mpad / D::kNT;
        ROWS_BEGIN(k, rw)
            const double d = R.D(k, rw.e);
            if (d != 0.0) // inactive rows keep e = 0
                S.e[rw.e] = d * rm.dot(p, S, k, rw, dw) - S.e[rw.e];
        ROWS_END
    }
    __syncthreads();
    FSTAMP(9);
    // multipliers of the fixed binaries from the stationarity row of their component
    const bool own_g = gsrc && gsrc != S.g; // a refinement call passes its residual in S.g (zero there)
    for (int o = lane; o < T * nub; o += D::kNT) {
        const int t = o / nub, b = o - t * nub;
        double a = 0;
        if (S.fix[o] >= 0) {
            const int c = nx + nuc + b;
            a = own_g ? gs * gsrc[t * nz + c] : 0.0;
            for (int j = 0; j < nz; j++) a -= S.P[c * nz + j] * dw[t * nz + j];
            a -= ccol_dot<D>(p, S, t, c, S.e);
            for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + c] * dlam[(t + 1) * nx + l];
        }
        dnuf[o] = a;
    }
    // lam_0 from the stationarity row of x_0 (prescribed: the row defines its multiplier, as for the fixed binaries).
    // Through the recursion -- lam_0 = -(P_0 x_0 + p_0) above -- it carries the rounding of the cost-to-go of stage 0,
    // eps |P_0| with |P_0| ~ max D: 2e-6 in that row of the dual residual on deep nodes of BASELINE configs[4] (every other
    // row: 1e-15), which the refinement -- it takes prescribed components as met -- never saw.  Same as oracle/hsde_qp.c.
    for (int i = lane; i < nx; i += D::kNT) {
        double a = own_g ? gs * gsrc[i] : 0.0;
        for (int j = 0; j < nz; j++) a -= S.P[i * nz + j] * dw[j];
        a -= ccol_dot<D>(p, S, 0, i, S.e);
        for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + i] * dlam[nx + l];
        dlam[i] = a;
    }
    __syncthreads();
    FSTAMP(10);
}

// ---------------------------------------------------------------------------------------------
// The same solve for the compile-time shapes, with the two recursions in registers of wave 0: lane j
// holds component j of the stage vector; a substitution step broadcasts one value with v_readlane and
// every lane updates its own component with its own multiplier (its row of L for the forward, its
// column of L_u for the back substitution).  Everything that does not depend on the recursion (C' e)
// is prepared by all waves before the sweep.  No barrier and no LDS round trip on the critical path.
// ---------------------------------------------------------------------------------------------
template <class D, int RS, class RM>
DEV void kkt_solve_reg(const DevProb &p, const Lds &S, Rows<RS> &R, const RM &rm, int lane, const ldsd *gsrc, double gs, bool usex0,
                       const ldsd *csrc, double cs, bool useb, ldsd *dw, ldsd *dlam, ldsd *dnuf FSTAMP_ARGS)
{
    constexpr int NX = D::kNX, NU = D::kNU, NUB = D::kNUB, NUC = NU - NUB, NZ = NX + NU;
    constexpr int NXS = NX * (NX + 1) / 2, LMS = LM_STAGE(NX, NU);
    const int T = p.T;
    FSTAMP_DECL;
    LANE_OPAQUE(lane);
    // S.g <- (mb if useb) - (rhs_d + C' e): the part of the stage gradient the recursion does not touch
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NZ; o += D::kNT) {
        const int t = o / NZ, j = o - t * NZ;
        const double a = (gsrc ? gs * gsrc[o] : 0.0) + ccol_dot<D>(p, S, t, j, S.e);
        S.g[o] = (useb ? S.g[o] : 0.0) - a;
    }
    // S.pv[t] <- P_{t+1} c_t (lanes (t, i)): the part of q_t = p_{t+1} + P_{t+1} c_t that does not depend on the
    // recursion; stage t reads it before it stores p_t in the same place
    for (int o = lane; o < T * NX; o += D::kNT) {
        const int t = o / NX, i = o - t * NX;
        double a = 0.0;
        if (csrc) {
#pragma unroll
            for (int l = 0; l < NX; l++) a += S.Pr[(t + 1) * NXS + sym(i, l)] * (cs * csrc[t * NX + l]);
        }
        S.pv[o] = a;
    }
    double pvr = 0.0; // lane i < NX: p_{t+1}[i]
    if (lane < NX) {
        pvr = -(gsrc ? gs * gsrc[T * NZ + lane] : 0.0);
        S.pv[T * NX + lane] = pvr;
    }
    __syncthreads();
    FSTAMP(6);
    LANE_OPAQUE(lane);
    // Both sweeps are software pipelined: everything stage t reads from LDS does not depend on the
    // recursion and is fetched while stage t +- 1 runs its chain of v_readlane / FMA steps, so that no
    // LDS latency sits on the critical path.  Skipped pivots (fixed binaries) have zero multipliers:
    // their substitution steps are no-ops and need no branch.
    if (D::kNW == 1 || lane < WAVE) { // backward sweep (wave 0)
        double ABcol[NX];
#pragma unroll
        for (int l = 0; l < NX; l++) ABcol[l] = lane < NZ ? S.AB[l * NZ + lane] : 0.0;
        // this lane's row of the multipliers: a state row has NU entries, input row i its i entries
        const int rowoff = lane < NX ? lane * NU : lane < NZ ? LM_U(NX, NU, lane - NX, 0) : 0;
        const int rowlen = lane < NX ? NU : lane < NZ ? lane - NX : 0;
        const int gl = lane < NZ ? lane : 0, bl = (lane >= NX + NUC && lane < NZ) ? lane - NX - NUC : 0;
        const int pl = lane < NX ? lane : 0;
        double mrow_n[NU], mpre_n, qc_n;
        int f_n;
        {
            const ldsd *row = S.Lm + (T - 1) * LMS + rowoff;
#pragma unroll
            for (int j = 0; j < NU; j++) mrow_n[j] = row[j];
            mpre_n = S.g[(T - 1) * NZ + gl];
            f_n = S.fix[(T - 1) * NUB + bl];
            qc_n = S.pv[(T - 1) * NX + pl];
        }
        for (int t = T - 1; t >= 0; t--) {
            double mrow[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) mrow[j] = j < rowlen ? mrow_n[j] : 0.0;
            const double mpre = lane < NZ ? mpre_n : 0.0;
            const int f = (lane >= NX + NUC && lane < NZ) ? f_n : -1;
            const double qv = pvr + qc_n;
            {
                const int tn = t > 0 ? t - 1 : 0; // the last round fetches stage 0 again (unused)
                const ldsd *row = S.Lm + tn * LMS + rowoff;
#pragma unroll
                for (int j = 0; j < NU; j++) mrow_n[j] = row[j];
                mpre_n = S.g[tn * NZ + gl];
                f_n = S.fix[tn * NUB + bl];
                qc_n = S.pv[tn * NX + pl];
            }
            double v = mpre;
#pragma unroll
            for (int l = 0; l < NX; l++) v += ABcol[l] * bcast16<D>(qv, l);
            if (f >= 0) v = (useb && f == 1) ? -1.0 : 0.0;
            // forward substitution: the factorisation's row operations applied to the vector
#pragma unroll
            for (int j = 0; j < NUC; j++) v -= mrow[j] * bcast16<D>(v, NX + j);
            if (!((S.fullfix >> t) & 1ull)) { // the binaries' steps are no-ops (zero multipliers) when all are fixed
#pragma unroll
                for (int j = NUC; j < NU; j++) v -= mrow[j] * bcast16<D>(v, NX + j);
            }
            if (lane >= NX && lane < NZ) dw[t * NZ + lane] = v; // y = L_u^{-1} m_u, parked in the input slots
            pvr = v;                                            // lanes < NX: p_t
            if (lane < NX) S.pv[t * NX + lane] = v;
        }
    }
    __syncthreads();
    FSTAMP(7);
    LANE_OPAQUE(lane);
    if (D::kNW == 1 || lane < WAVE) { // forward sweep (wave 0)
        double ABrow[NZ];
#pragma unroll
        for (int l = 0; l < NZ; l++) ABrow[l] = lane < NX ? S.AB[lane * NZ + l] : 0.0;
        double xr = (lane < NX && usex0) ? S.x0[lane] : 0.0;
        const int cl = lane < NU ? lane : 0, pl = lane < NX ? lane : 0;
        // lane j < NU: column j of L_x and of L_u (entries below the diagonal), reciprocal pivot, y_j
        double lx_n[NX], lcol_n[NU], dinv_n, y_n, cdy_n;
        {
            const ldsd *colp = S.Lm + cl;
#pragma unroll
            for (int l = 0; l < NX; l++) lx_n[l] = colp[LM_X(NX, NU, l, 0)];
#pragma unroll
            for (int i = 1; i < NU; i++) lcol_n[i] = colp[LM_U(NX, NU, i, 0)];
            dinv_n = S.dinv[cl];
            y_n = dw[NX + cl];
            cdy_n = csrc ? csrc[pl] : 0.0;
        }
        for (int t = 0; t < T; t++) {
            double lx[NX], lcol[NU];
#pragma unroll
            for (int l = 0; l < NX; l++) lx[l] = lane < NU ? lx_n[l] : 0.0;
            lcol[0] = 0.0;
#pragma unroll
            for (int i = 1; i < NU; i++) lcol[i] = lane < i ? lcol_n[i] : 0.0; // row i has entries in columns < i
            double cur = lane < NU ? dinv_n * y_n : 0.0;
            double xn = (csrc && lane < NX) ? cs * cdy_n : 0.0;
            {
                const int tn = t + 1 < T ? t + 1 : t;
                const ldsd *colp = S.Lm + tn * LMS + cl;
#pragma unroll
                for (int l = 0; l < NX; l++) lx_n[l] = colp[LM_X(NX, NU, l, 0)];
#pragma unroll
                for (int i = 1; i < NU; i++) lcol_n[i] = colp[LM_U(NX, NU, i, 0)];
                dinv_n = S.dinv[tn * NU + cl];
                y_n = dw[tn * NZ + NX + cl];
                cdy_n = csrc ? csrc[tn * NX + pl] : 0.0;
            }
#pragma unroll
            for (int l = 0; l < NX; l++) {
                const double xl = bcast16<D>(xr, l);
                cur += lx[l] * xl;
                xn += ABrow[l] * xl;
            }
            // back substitution with L_u' ; u_j = -(value of lane j once its turn has come)
#pragma unroll
            for (int j = NU - 1; j >= 0; j--) {
                const double uj = bcast16<D>(cur, j);
                cur -= lcol[j] * uj; // only lanes i < j hold a nonzero (j, i) entry
                xn -= ABrow[NX + j] * uj;
            }
            if (lane < NX) dw[t * NZ + lane] = xr;
            if (lane < NU) dw[t * NZ + NX + lane] = -cur;
            xr = xn;
        }
        if (lane < NX) dw[T * NZ + lane] = xr;
    }
    __syncthreads();
    FSTAMP(9);
    LANE_OPAQUE(lane);
    // equality multipliers lam_t = -(P_t x_t + p_t) ; dz = D (C dw) - e
    for (int o = lane; o < (T + 1) * NX; o += D::kNT) {
        const int t = o / NX, i = o - t * NX;
        double a = S.pv[o];
#pragma unroll
        for (int l = 0; l < NX; l++) a += S.Pr[t * NXS + sym(i, l)] * dw[t * NZ + l];
        dlam[o] = -a;
    }
    if constexpr (!D::kDpp) { // (15 slots per lane: the batched form below spills)
        const int nslot = RS;
        ROWS_BEGIN(k, rw)
            const double d = R.D(k, rw.e);
            if (d != 0.0) // inactive rows keep e = 0
                S.e[rw.e] = d * rm.dot(p, S, k, rw, dw) - S.e[rw.e];
        ROWS_END
    } else {
        // dz = D (C dw) - e on the [F G] and bound slots: all loads, then all stores (inactive rows keep e = 0)
        constexpr int KS = RM::kSlotsFB;
        double val[KS];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            typename RM::Ref rw;
            const bool ok = rm.at(p, k, lane, rw);
            const double d = ok ? R.D(k, rw.e) : 0.0;
            val[k] = d * rm.dot_all(k, ok, rw, dw) - S.e[ok ? rw.e : 0];
            if (d == 0.0) val[k] = __longlong_as_double(0x7ff8000000000000LL); // marks "no store"
        }
#pragma unroll
        for (int k = 0; k < KS; k++) {
            typename RM::Ref rw;
            (void)rm.at(p, k, lane, rw);
            if (val[k] == val[k]) S.e[rw.e] = val[k];
        }
        if (S.term_on) { // terminal-set rows (dense, global memory): only in the second solve of a node
#pragma unroll
            for (int k = KS; k < RS; k++) {
                typename RM::Ref rw;
                if (rm.at(p, k, lane, rw)) {
                    const double d = R.D(k, rw.e);
                    if (d != 0.0) S.e[rw.e] = d * rm.dot(p, S, k, rw, dw) - S.e[rw.e];
                }
            }
        }
    }
    __syncthreads();
    // multipliers of the fixed binaries from the stationarity row of their component
    const bool own_g = gsrc && gsrc != S.g;
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NUB; o += D::kNT) {
        const int t = o / NUB, b = o - t * NUB;
        double a = 0;
        if (S.fix[o] >= 0) {
            const int c = NX + NUC + b;
            a = own_g ? gs * gsrc[t * NZ + c] : 0.0;
#pragma unroll
            for (int j = 0; j < NZ; j++) a -= S.P[c * NZ + j] * dw[t * NZ + j];
            a -= ccol_dot<D>(p, S, t, c, S.e);
#pragma unroll
            for (int l = 0; l < NX; l++) a += S.AB[l * NZ + c] * dlam[(t + 1) * NX + l];
        }
        dnuf[o] = a;
    }
    if constexpr (HMPC_LAM0_ROW) { // lam_0 from the stationarity row of x_0 (see kkt_solve)
        if (lane < NX) {
            double a = own_g ? gs * gsrc[lane] : 0.0;
#pragma unroll
            for (int j = 0; j < NZ; j++) a -= S.P[lane * NZ + j] * dw[j];
            a -= ccol_dot<D>(p, S, 0, lane, S.e);
#pragma unroll
            for (int l = 0; l < NX; l++) a += S.AB[l * NZ + lane] * dlam[NX + l];
            dlam[lane] = a;
        }
    }
    __syncthreads();
    FSTAMP(10);
}

// ---------------------------------------------------------------------------------------------
// TWO solves of one iteration in one pair of sweeps (round 5; register kernels, interior-point iterations only).  The constant
// direction (right-hand side (0; f; h): what kkt_solve_reg(.., nullptr, 0, true, nullptr, 0, true, S.w1 ..) computes) and the
// affine direction of the predictor (.., S.rd, -1, false, S.rdyn, -1, false, S.w2 ..) depend on the factorisation only, not on
// each other.  A sweep is a chain of broadcast / multiply-add steps on wave 0 that nothing overlaps with (one wave per SIMD):
// two right-hand sides through the same chain share every multiplier fetch and give the chain's latency something to hide
// behind.  Row phases, column products and the multipliers of the equalities are what they are in two separate solves, in
// another order: the arithmetic of each direction is unchanged.
//   On entry : R.D holds D, S.g the factorisation's mb, S.rd / S.rdyn the residuals, S.w the iterate.
//   On exit  : S.w1, S.lam1, S.nuf1 the constant direction; S.w2, S.lam2, S.nuf2 the affine direction; S.e the affine
//              direction's dz; returns this lane's part of sum dz1^2 / D over its rows (the tau step's denominator).
//   Storage  : the affine direction's stage gradient lives in S.w2 itself (the backward sweep reads stage t - 1 before it parks
//              stage t's y there), its p_t in S.edyn (dead outside the refinement; carved (T + 1) nx long for this).
// ---------------------------------------------------------------------------------------------
template <class D, int RS, class RM>
DEV double kkt_solve_reg_pair(const DevProb &p, const Lds &S, Rows<RS> &R, const RM &rm, int lane, double tau FSTAMP_ARGS)
{
    constexpr int NX = D::kNX, NU = D::kNU, NUB = D::kNUB, NUC = NU - NUB, NZ = NX + NU;
    constexpr int NXS = NX * (NX + 1) / 2, LMS = LM_STAGE(NX, NU);
    const int T = p.T;
    const int nslot = RS;
    ldsd *pv2 = S.edyn;
    FSTAMP_DECL;
    LANE_OPAQUE(lane);
    // ---- constant direction: e = D h ; S.g <- mb - C' e ----
    ROWS_BEGIN(k, rw)
        S.e[rw.e] = R.D(k, rw.e) * rm.h(p, S, k, rw);
    ROWS_END
    __syncthreads();
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NZ; o += D::kNT) {
        const int t = o / NZ, j = o - t * NZ;
        S.g[o] = S.g[o] - ccol_dot<D>(p, S, t, j, S.e);
    }
    for (int o = lane; o < (T + 1) * NX; o += D::kNT) S.pv[o] = 0.0;
    __syncthreads();
    // ---- affine direction: e = z - D rc (the predictor's right-hand side) ; S.w2 <- rd - C' e ----
    ROWS_BEGIN(k, rw)
        double v = 0;
        const double d = R.D(k, rw.e);
        if (d != 0.0) {
            const double sr = R.s(k, rw.e), zr = d * sr;
            const double rc = sr - rm.h(p, S, k, rw) * tau + rm.dot(p, S, k, rw, S.w); // row residual
            const double dsr = sr * zr;
            v = dsr * frcp(sr) - 1.0 * (d * rc);
        }
        S.e[rw.e] = v;
        R.dz(k, rw.e) = v; // (kept: S.e carries the constant direction's dz for a while below)
    ROWS_END
    __syncthreads();
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NZ; o += D::kNT) {
        const int t = o / NZ, j = o - t * NZ;
        const double a = -1.0 * S.rd[o] + ccol_dot<D>(p, S, t, j, S.e);
        S.w2[o] = 0.0 - a;
    }
    for (int o = lane; o < T * NX; o += D::kNT) {
        const int t = o / NX, i = o - t * NX;
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; l++) a += S.Pr[(t + 1) * NXS + sym(i, l)] * (-1.0 * S.rdyn[t * NX + l]);
        pv2[o] = a;
    }
    double pvrA = 0.0, pvrB = 0.0; // lane i < NX: p_{t+1}[i] of the two directions
    if (lane < NX) {
        pvrB = -(-1.0 * S.rd[T * NZ + lane]);
        pv2[T * NX + lane] = pvrB;
    }
    __syncthreads();
    FSTAMP(6);
    LANE_OPAQUE(lane);
    if (D::kNW == 1 || lane < WAVE) { // backward sweep (wave 0), both directions
        double ABcol[NX];
#pragma unroll
        for (int l = 0; l < NX; l++) ABcol[l] = lane < NZ ? S.AB[l * NZ + lane] : 0.0;
        const int rowoff = lane < NX ? lane * NU : lane < NZ ? LM_U(NX, NU, lane - NX, 0) : 0;
        const int rowlen = lane < NX ? NU : lane < NZ ? lane - NX : 0;
        const int gl = lane < NZ ? lane : 0, bl = (lane >= NX + NUC && lane < NZ) ? lane - NX - NUC : 0;
        const int pl = lane < NX ? lane : 0;
        double mrow_n[NU], mpreA_n, mpreB_n, qcA_n, qcB_n;
        int f_n;
        {
            const ldsd *row = S.Lm + (T - 1) * LMS + rowoff;
#pragma unroll
            for (int j = 0; j < NU; j++) mrow_n[j] = row[j];
            mpreA_n = S.g[(T - 1) * NZ + gl];
            mpreB_n = S.w2[(T - 1) * NZ + gl];
            f_n = S.fix[(T - 1) * NUB + bl];
            qcA_n = S.pv[(T - 1) * NX + pl];
            qcB_n = pv2[(T - 1) * NX + pl];
        }
        for (int t = T - 1; t >= 0; t--) {
            double mrow[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) mrow[j] = j < rowlen ? mrow_n[j] : 0.0;
            const double mpreA = lane < NZ ? mpreA_n : 0.0, mpreB = lane < NZ ? mpreB_n : 0.0;
            const int f = (lane >= NX + NUC && lane < NZ) ? f_n : -1;
            const double qvA = pvrA + qcA_n, qvB = pvrB + qcB_n;
            {
                const int tn = t > 0 ? t - 1 : 0; // the last round fetches stage 0 again (unused)
                const ldsd *row = S.Lm + tn * LMS + rowoff;
#pragma unroll
                for (int j = 0; j < NU; j++) mrow_n[j] = row[j];
                mpreA_n = S.g[tn * NZ + gl];
                mpreB_n = S.w2[tn * NZ + gl];
                f_n = S.fix[tn * NUB + bl];
                qcA_n = S.pv[tn * NX + pl];
                qcB_n = pv2[tn * NX + pl];
            }
            double vA = mpreA, vB = mpreB;
#pragma unroll
            for (int l = 0; l < NX; l++) {
                vA += ABcol[l] * bcast16<D>(qvA, l);
                vB += ABcol[l] * bcast16<D>(qvB, l);
            }
            if (f >= 0) { vA = (f == 1) ? -1.0 : 0.0; vB = 0.0; }
#pragma unroll
            for (int j = 0; j < NUC; j++) {
                vA -= mrow[j] * bcast16<D>(vA, NX + j);
                vB -= mrow[j] * bcast16<D>(vB, NX + j);
            }
            if (!((S.fullfix >> t) & 1ull)) {
#pragma unroll
                for (int j = NUC; j < NU; j++) {
                    vA -= mrow[j] * bcast16<D>(vA, NX + j);
                    vB -= mrow[j] * bcast16<D>(vB, NX + j);
                }
            }
            if (lane >= NX && lane < NZ) { S.w1[t * NZ + lane] = vA; S.w2[t * NZ + lane] = vB; }
            pvrA = vA;
            pvrB = vB;
            if (lane < NX) { S.pv[t * NX + lane] = vA; pv2[t * NX + lane] = vB; }
        }
    }
    __syncthreads();
    FSTAMP(7);
    LANE_OPAQUE(lane);
    if (D::kNW == 1 || lane < WAVE) { // forward sweep (wave 0), both directions
        double ABrow[NZ];
#pragma unroll
        for (int l = 0; l < NZ; l++) ABrow[l] = lane < NX ? S.AB[lane * NZ + l] : 0.0;
        double xrA = lane < NX ? S.x0[lane] : 0.0, xrB = 0.0;
        const int cl = lane < NU ? lane : 0, pl = lane < NX ? lane : 0;
        double lx_n[NX], lcol_n[NU], dinv_n, yA_n, yB_n, cdy_n;
        {
            const ldsd *colp = S.Lm + cl;
#pragma unroll
            for (int l = 0; l < NX; l++) lx_n[l] = colp[LM_X(NX, NU, l, 0)];
#pragma unroll
            for (int i = 1; i < NU; i++) lcol_n[i] = colp[LM_U(NX, NU, i, 0)];
            dinv_n = S.dinv[cl];
            yA_n = S.w1[NX + cl];
            yB_n = S.w2[NX + cl];
            cdy_n = S.rdyn[pl];
        }
        for (int t = 0; t < T; t++) {
            double lx[NX], lcol[NU];
#pragma unroll
            for (int l = 0; l < NX; l++) lx[l] = lane < NU ? lx_n[l] : 0.0;
            lcol[0] = 0.0;
#pragma unroll
            for (int i = 1; i < NU; i++) lcol[i] = lane < i ? lcol_n[i] : 0.0;
            double curA = lane < NU ? dinv_n * yA_n : 0.0, curB = lane < NU ? dinv_n * yB_n : 0.0;
            double xnA = 0.0, xnB = lane < NX ? -1.0 * cdy_n : 0.0;
            {
                const int tn = t + 1 < T ? t + 1 : t;
                const ldsd *colp = S.Lm + tn * LMS + cl;
#pragma unroll
                for (int l = 0; l < NX; l++) lx_n[l] = colp[LM_X(NX, NU, l, 0)];
#pragma unroll
                for (int i = 1; i < NU; i++) lcol_n[i] = colp[LM_U(NX, NU, i, 0)];
                dinv_n = S.dinv[tn * NU + cl];
                yA_n = S.w1[tn * NZ + NX + cl];
                yB_n = S.w2[tn * NZ + NX + cl];
                cdy_n = S.rdyn[tn * NX + pl];
            }
#pragma unroll
            for (int l = 0; l < NX; l++) {
                const double xlA = bcast16<D>(xrA, l), xlB = bcast16<D>(xrB, l);
                curA += lx[l] * xlA;
                curB += lx[l] * xlB;
                xnA += ABrow[l] * xlA;
                xnB += ABrow[l] * xlB;
            }
#pragma unroll
            for (int j = NU - 1; j >= 0; j--) {
                const double ujA = bcast16<D>(curA, j), ujB = bcast16<D>(curB, j);
                curA -= lcol[j] * ujA;
                curB -= lcol[j] * ujB;
                xnA -= ABrow[NX + j] * ujA;
                xnB -= ABrow[NX + j] * ujB;
            }
            if (lane < NX) { S.w1[t * NZ + lane] = xrA; S.w2[t * NZ + lane] = xrB; }
            if (lane < NU) { S.w1[t * NZ + NX + lane] = -curA; S.w2[t * NZ + NX + lane] = -curB; }
            xrA = xnA;
            xrB = xnB;
        }
        if (lane < NX) { S.w1[T * NZ + lane] = xrA; S.w2[T * NZ + lane] = xrB; }
    }
    __syncthreads();
    FSTAMP(9);
    LANE_OPAQUE(lane);
    // equality multipliers lam_t = -(P_t x_t + p_t) of both directions ; the constant direction's dz = D (C w1) - D h into S.e
    for (int o = lane; o < (T + 1) * NX; o += D::kNT) {
        const int t = o / NX, i = o - t * NX;
        double a = S.pv[o], b = pv2[o];
#pragma unroll
        for (int l = 0; l < NX; l++) {
            const double pr = S.Pr[t * NXS + sym(i, l)];
            a += pr * S.w1[t * NZ + l];
            b += pr * S.w2[t * NZ + l];
        }
        S.lam1[o] = -a;
        S.lam2[o] = -b;
    }
    double q2 = 0.0;
    ROWS_BEGIN(k, rw)
        const double d = R.D(k, rw.e);
        double z1 = 0.0;
        if (d != 0.0) { // inactive rows keep e = 0
            z1 = d * rm.dot(p, S, k, rw, S.w1) - d * rm.h(p, S, k, rw);
            q2 += z1 * z1 * frcp(d);
        }
        S.e[rw.e] = z1;
    ROWS_END
    __syncthreads();
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NUB; o += D::kNT) { // multipliers of the fixed binaries from the stationarity row of their component
        const int t = o / NUB, b = o - t * NUB;
        double a = 0;
        if (S.fix[o] >= 0) {
            const int c = NX + NUC + b;
#pragma unroll
            for (int j = 0; j < NZ; j++) a -= S.P[c * NZ + j] * S.w1[t * NZ + j];
            a -= ccol_dot<D>(p, S, t, c, S.e);
#pragma unroll
            for (int l = 0; l < NX; l++) a += S.AB[l * NZ + c] * S.lam1[(t + 1) * NX + l];
        }
        S.nuf1[o] = a;
    }
    if (lane < NX) { // lam_0 from the stationarity row of x_0 (see kkt_solve)
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NZ; j++) a -= S.P[lane * NZ + j] * S.w1[j];
        a -= ccol_dot<D>(p, S, 0, lane, S.e);
#pragma unroll
        for (int l = 0; l < NX; l++) a += S.AB[l * NZ + lane] * S.lam1[NX + l];
        S.lam1[lane] = a;
    }
    __syncthreads();
    // the affine direction's dz = D (C w2) - e into S.e (e was kept in the rows' dz slots)
    ROWS_BEGIN(k, rw)
        const double d = R.D(k, rw.e);
        double z2 = 0.0;
        if (d != 0.0) z2 = d * rm.dot(p, S, k, rw, S.w2) - R.dz(k, rw.e);
        S.e[rw.e] = z2;
    ROWS_END
    __syncthreads();
    term_cols<D, RM>(p, S, lane, S.e);
    for (int o = lane; o < T * NUB; o += D::kNT) {
        const int t = o / NUB, b = o - t * NUB;
        double a = 0;
        if (S.fix[o] >= 0) {
            const int c = NX + NUC + b;
            a = -1.0 * S.rd[t * NZ + c];
#pragma unroll
            for (int j = 0; j < NZ; j++) a -= S.P[c * NZ + j] * S.w2[t * NZ + j];
            a -= ccol_dot<D>(p, S, t, c, S.e);
#pragma unroll
            for (int l = 0; l < NX; l++) a += S.AB[l * NZ + c] * S.lam2[(t + 1) * NX + l];
        }
        S.nuf2[o] = a;
    }
    if (lane < NX) {
        double a = -1.0 * S.rd[lane];
#pragma unroll
        for (int j = 0; j < NZ; j++) a -= S.P[lane * NZ + j] * S.w2[j];
        a -= ccol_dot<D>(p, S, 0, lane, S.e);
#pragma unroll
        for (int l = 0; l < NX; l++) a += S.AB[l * NZ + lane] * S.lam2[NX + l];
        S.lam2[lane] = a;
    }
    __syncthreads();
    FSTAMP(10);
    return q2;
}

template <class D, int RS, class RM>
DEV void kkt_dispatch(const DevProb &p, const Lds &S, Rows<RS> &R, const RM &rm, int lane, const ldsd *gsrc, double gs, bool usex0,
                      const ldsd *csrc, double cs, bool useb, ldsd *dw, ldsd *dlam, ldsd *dnuf FSTAMP_ARGS)
{
    if constexpr (D::kNX > 0) kkt_solve_reg<D, RS>(p, S, R, rm, lane, gsrc, gs, usex0, csrc, cs, useb, dw, dlam, dnuf FSTAMP_PASS);
    else kkt_solve<D, RS>(p, S, R, rm, lane, gsrc, gs, usex0, csrc, cs, useb, dw, dlam, dnuf FSTAMP_PASS);
}

// f'y + h'z of a direction / iterate (lam_0, multipliers of binaries fixed to one, row multipliers in zrow)
template <class D, class RM>
DEV double lin_obj(const DevProb &p, const Lds &S, const RM &rm, int lane, const ldsd *lam, const ldsd *nuf, const ldsd *zrow)
{
    double a = 0;
    LANE_OPAQUE(lane);
    for (int j = lane; j < D::nx(p); j += D::kNT) a += S.x0[j] * lam[j];
    for (int o = lane; o < p.T * D::nub(p); o += D::kNT)
        if (S.fix[o] == 1) a += nuf[o];
    const int nslot = RM::kSlots > 0 ? RM::kSlots : p.Mpad / D::kNT;
    ROWS_BEGIN(k, rw)
        if (rm.active(p, S, k, rw)) a += rm.h(p, S, k, rw) * zrow[rw.e];
    ROWS_END
    return a; // per-thread partial sum: the caller reduces it together with its other sums
}

template <class D> DEV void set_prescribed(const DevProb &p, const Lds &S, int lane, double tau)
{
    const int nub = D::nub(p);
    for (int i = lane; i < D::nx(p); i += D::kNT) S.w[i] = S.x0[i] * tau;
    for (int o = lane; o < p.T * nub; o += D::kNT)
        if (S.fix[o] >= 0) S.w[(o / nub) * D::nz(p) + D::nx(p) + D::nuc(p) + (o % nub)] = S.fix[o] * tau;
}

// w' P v (per-lane partial sum) for a direction v
template <class D> DEV double wPv(const DevProb &p, const Lds &S, int lane, const ldsd *v)
{
    const int nx = D::nx(p), nz = D::nz(p), T = p.T, n = T * nz + nx;
    double acc = 0;
    LANE_OPAQUE(lane);
    for (int o = lane; o < n; o += D::kNT) {
        const int t = o / nz < T ? o / nz : T;
        const int i = o - t * nz, dim = t < T ? nz : nx;
        const ldsd *PP = t < T ? S.P : S.PT;
        double a = 0;
        for (int j = 0; j < dim; j++) a += PP[i * dim + j] * v[t * nz + j];
        acc += a * S.w[o];
    }
    return acc;
}

// (s w - v)' P (s w - v) (per-lane partial sum) for a direction v
template <class D> DEV double dPd(const DevProb &p, const Lds &S, int lane, const ldsd *v, double sc)
{
    const int nx = D::nx(p), nz = D::nz(p), T = p.T, n = T * nz + nx;
    double acc = 0;
    LANE_OPAQUE(lane);
    for (int o = lane; o < n; o += D::kNT) {
        const int t = o / nz < T ? o / nz : T;
        const int i = o - t * nz, dim = t < T ? nz : nx;
        const ldsd *PP = t < T ? S.P : S.PT;
        double a = 0;
        for (int j = 0; j < dim; j++) a += PP[i * dim + j] * (sc * S.w[t * nz + j] - v[t * nz + j]);
        acc += a * (sc * S.w[o] - v[o]);
    }
    return acc;
}

// One interior-point solve of the node with / without the terminal-set rows.
// Returns status; tau and the iteration count through references.
// WARM: the instantiation that accepts a parent's record (hmpc_warm).  The cold kernels are compiled without any of it:
// the hand-down code in front of the main loop costs the loop registers (scratch 116 -> 140 B per lane, 8 % on every
// launch, measured) -- a launch without hand-down runs the kernel it always ran.
template <class D, int RS, class RM, bool WARM>
DEV int ipm_solve(const DevProb &p, const Lds &S, Rows<RS> &R, RM &rm, int lane, int term_on, int &iters, double &tau_out,
                  bool &polished_out, int &weak_out, bool &handed_out, double *trace, const double *wprim, const double *wdual,
                  bool attempt_only, bool own = false)
{
    const int nx = D::nx(p), nz = D::nz(p), T = p.T, nuc = D::nuc(p), nub = D::nub(p), M = p.M, n = T * nz + nx;
    const int nslot = RS > 0 ? RS : p.Mpad / D::kNT;
    int mact = 0;
    rm.prepare(p, S, lane, term_on);
    ROWS_BEGIN(k, rw)
        const bool on = rm.active(p, S, k, rw);
        mact += on;
        R.s(k, rw.e) = 1.0;
        R.z(k, rw.e) = on ? 1.0 : 0.0;
    ROWS_END
    mact = (int)block_sum<D>((double)mact, S.red, lane);
    for (int i = lane; i < n; i += D::kNT) S.w[i] = 0.0;
    for (int i = lane; i < (T + 1) * nx; i += D::kNT) S.lam[i] = 0.0;
    for (int i = lane; i < T * nub; i += D::kNT) S.nuf[i] = 0.0;
    double tau = 1.0, kap = 1.0;
    __syncthreads();
    set_prescribed<D>(p, S, lane, tau);
    __syncthreads();
    double x0inf = 0;
    for (int i = lane; i < nx; i += D::kNT) x0inf = fmax(x0inf, fabs(S.x0[i]));
    x0inf = block_max<D>(x0inf, S.red, lane);

    int status = HMPC_MAXITER, it = 0, extra_done = 0;
    polished_out = false;
    weak_out = 0;
    handed_out = false;
    bool tried = false; // the polish has been tried (and failed) on the current iterate
    int attempts = 0;
    double last_alpha = 0, last_dtau = 0, last_dkap = 0;
#ifdef HMPC_TRACE_STEP
    double dbg_den = 0;
#endif
#ifdef HMPC_STAMPS
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
    long long facc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // One loop runs the interior-point iterations (mode 0) and the passes of the ACTIVE-SET POLISH (mode 1: a new
    // active set, with factorisation; mode 2: a further multiplier step with the factorisation at hand): the
    // polish reuses the factorisation and the constant-direction solve below instead of owning copies of them.
    int mode = 0, round = 0, al = 0;
    // penalty of the polish in progress: 0 first level; 1 second; 2 back at the first for the last digits.  p.polish_l1 (a cost
    // whose curvature is of order one, DevProb): the polish starts at the second level and ends there
    int level = p.polish_l1;
    double rg = 0, mu = 0, wPw = 0, winf = 0, zinf = 0;
    // PARENT -> CHILD HAND-DOWN (the reference hands the parent's simplex basis to the child: controller.py:260-264,
    // subproblem_solution.py:37-43).  wprim / wdual: the parent's record.  Its active set -- the rows with a positive
    // multiplier that still exist in this node -- is tried by the polish passes below BEFORE the first interior-point
    // iteration, multipliers and proximal centre from the parent: a child whose optimum lies on the same set (the branch
    // that fixes a binary where the relaxation had it) verifies after one factorisation and a few solves.  A set that does
    // not verify within HMPC_POLISH_ROUNDS_WARM rounds, or misses a row by HMPC_POLISH_WARM_VMAX, is dropped and the cold
    // start below runs untouched.  Same steps as oracle/hsde_qp.c.
    bool warm_try = false;
    if constexpr (WARM) if (wprim != nullptr && p.polish) {
        const int nu = D::nu(p);
        for (int o = lane; o < n; o += D::kNT) {
            const int t = o / nz < T ? o / nz : T;
            const int j = o - t * nz;
            S.w[o] = j < nx ? wprim[t * nx + j] : wprim[(T + 1) * nx + t * nu + (j - nx)];
        }
        __syncthreads();
        // a binary this node fixes far from where the parent's relaxation had it (the 1-branch of a binary relaxed to 0: most
        // infeasible children): the parent's set is not near this node's optimum, nothing is tried
        double bmove = 0.0;
        for (int o = lane; o < T * nub; o += D::kNT)
            if (S.fix[o] >= 0) bmove = fmax(bmove, fabs(S.w[(o / nub) * nz + nx + nuc + (o % nub)] - (double)S.fix[o]));
        bmove = block_max<D>(bmove, S.red, lane);
        set_prescribed<D>(p, S, lane, 1.0);
        __syncthreads();
        const int o_mu = (T + 1) * nx, o_lb = o_mu + (T - 1) * p.nc + p.ncL, o_ub = o_lb + T * nub;
        double wi = 0, zi = 0;
        for (int o = lane; o < n; o += D::kNT) wi = fmax(wi, fabs(S.w[o]));
        for (int o = lane; o < (T + 1) * nx; o += D::kNT) zi = fmax(zi, fabs(wdual[o]) * p.cs);
        ROWS_BEGIN(k, rw)
            double d = 0.0;
            if (rm.active(p, S, k, rw)) {
                int t, lr;
                row_decode(p, rw.e, t, lr);
                double zr; // the parent's multiplier of this row in the units of the scaled problem
                if (lr < p.nc) zr = wdual[o_mu + t * p.nc + lr] * p.cs / p.reg.scale[lr];
                else if (lr >= p.mreg) zr = wdual[o_mu + t * p.nc + p.nc + (lr - p.mreg)] * p.cs / p.sct[lr - p.mreg];
                else if (lr < p.nc + nub) zr = wdual[o_lb + t * nub + (lr - p.nc)] * p.cs;
                else zr = wdual[o_ub + t * nub + (lr - p.nc - nub)] * p.cs;
                zi = fmax(zi, zr);
                R.prod(k, rw.e) = 1.0; // the cold start's multiplier, for the way back
                if (zr > 0.0) { d = HMPC_RHO_OF(p.polish_l1); R.dz(k, rw.e) = zr; }
                else { d = HMPC_POLISH_DELTA; R.dz(k, rw.e) = rm.dot(p, S, k, rw, S.w); }
                R.D(k, rw.e) = d;
            }
            S.e[rw.e] = d;
        ROWS_END
        {
            double v[2] = {wi, zi};
            const int op[2] = {1, 1};
            block_reduce<D, 2>(v, op, S.red, lane);
            winf = v[0]; zinf = v[1];
        }
        if (winf == winf && zinf == zinf && bmove <= HMPC_POLISH_WARM_BMOVE) { mode = 1; warm_try = true; }
        else { // (a parent record that is not a finite point: nothing to hand down)
            ROWS_BEGIN(k, rw)
                if (R.D(k, rw.e) != 0.0) R.z(k, rw.e) = 1.0;
            ROWS_END
            for (int o = lane; o < n; o += D::kNT) S.w[o] = 0.0;
            __syncthreads();
            set_prescribed<D>(p, S, lane, tau);
        }
        __syncthreads();
        if (attempt_only && !warm_try) { // (nothing to try: the caller runs the regular sequence of solves)
            iters = 0;
            tau_out = tau;
            return HMPC_RETRY;
        }
    }
    for (it = 0; it <= p.max_iter;) {
      if (mode == 0) {
        STAMP(7);
        LANE_OPAQUE(lane);
        // ---------------- residuals ----------------
        ROWS_BEGIN(k, rw)
            S.e[rw.e] = R.z(k, rw.e); // S.e <- z for the C'z products below
        ROWS_END
        __syncthreads();
        wPw = 0;
        for (int o = lane; o < n; o += D::kNT) {
            const int t = o / nz < T ? o / nz : T;
            const int i = o - t * nz, dim = t < T ? nz : nx;
            const ldsd *PP = t < T ? S.P : S.PT;
            double a = 0;
            for (int j = 0; j < dim; j++) a += PP[i * dim + j] * S.w[t * nz + j];
            wPw += a * S.w[o];
        }
        double rdinf = 0, certinf = 0, fy = 0, yinf = 0;
        winf = 0;
        term_cols<D, RM>(p, S, lane, S.e);
        for (int o = lane; o < n; o += D::kNT) {
            const int t = o / nz < T ? o / nz : T;
            const int j = o - t * nz;
            double a = 0; // E'y + C'z
            if (t == T) {
                a = S.lam[T * nx + j];
            } else {
                if (j < nx) a += S.lam[t * nx + j];
                for (int l = 0; l < nx; l++) a -= S.AB[l * AB_STRIDE + j] * S.lam[(t + 1) * nx + l];
                if (j >= nx + nuc && S.fix[t * nub + (j - nx - nuc)] >= 0) a += S.nuf[t * nub + (j - nx - nuc)];
                a += ccol_dot<D>(p, S, t, j, S.e);
            }
            double pw = 0; // (P w)_o
            {
                const int dim = t < T ? nz : nx;
                const ldsd *PP = t < T ? S.P : S.PT;
                for (int l = 0; l < dim; l++) pw += PP[j * dim + l] * S.w[t * nz + l];
            }
            const double v = pw + a;
            S.rd[o] = v;
            rdinf = fmax(rdinf, fabs(v));
            certinf = fmax(certinf, fabs(a));
            winf = fmax(winf, fabs(S.w[o]));
        }
        double rcinf = 0;
        for (int o = lane; o < T * nx; o += D::kNT) {
            const int t = o / nx, i = o - t * nx;
            double a = S.w[(t + 1) * nz + i];
            for (int l = 0; l < nz; l++) a -= S.AB[i * AB_STRIDE + l] * S.w[t * nz + l];
            S.rdyn[o] = a;
            rcinf = fmax(rcinf, fabs(a));
        }
        double hz = 0, sz = 0;
        zinf = 0;
        ROWS_BEGIN(k, rw)
            double a = 0;
            if (rm.active(p, S, k, rw)) {
                const double zr = R.z(k, rw.e), sr = R.s(k, rw.e), hh = rm.h(p, S, k, rw);
                a = sr - hh * tau + rm.dot(p, S, k, rw, S.w);
                hz += hh * zr;
                sz += sr * zr;
                zinf = fmax(zinf, zr);
            }
            rcinf = fmax(rcinf, fabs(a));
        ROWS_END
        for (int o = lane; o < (T + 1) * nx; o += D::kNT) yinf = fmax(yinf, fabs(S.lam[o]));
        for (int o = lane; o < T * nub; o += D::kNT) {
            yinf = fmax(yinf, fabs(S.nuf[o]));
            if (S.fix[o] == 1) fy += S.nuf[o];
        }
        for (int j = lane; j < nx; j += D::kNT) fy += S.x0[j] * S.lam[j];
        {
            double v[9] = {rdinf, certinf, winf, rcinf, fmax(zinf, yinf), fy, hz, sz, wPw};
            const int op[9] = {1, 1, 1, 1, 1, 0, 0, 0, 0};
            block_reduce<D, 9>(v, op, S.red, lane);
            rdinf = v[0]; certinf = v[1]; winf = v[2]; rcinf = v[3]; zinf = v[4]; fy = v[5]; hz = v[6]; sz = v[7]; wPw = v[8];
        }
        rg = wPw / tau + fy + hz + kap;
        mu = (sz + tau * kap) / (mact + 1);
        STAMP(0);

        // ---------------- termination ----------------
        const double pobj = 0.5 * wPw / (tau * tau), dob = -0.5 * wPw / (tau * tau) - (fy + hz) / tau;
        const double gap = fabs(pobj - dob), eta = -(fy + hz);
        if (trace && lane == 0 && it < 64) {
            double *tr = trace + it * 8;
            tr[0] = tau; tr[1] = kap; tr[2] = mu; tr[3] = rcinf / tau; tr[4] = rdinf / tau; tr[5] = gap; tr[6] = eta; tr[7] = certinf;
#ifdef HMPC_TRACE_STEP // (diagnostic builds: the step that led here instead of the certificate columns)
            tr[6] = last_dtau; tr[7] = last_alpha; tr[1] = dbg_den;
#endif
        }
        // Two levels (as in Ipopt's acceptable / desired tolerances).  ACCEPTABLE: scaled residuals and
        // gap <= tol.  DESIRED: acceptable and gap, dual residual <= 1e-2 tol -- they bound the suboptimality, and
        // with the curvature of this cost a 1e-8 gap still leaves ~2e-5 in the trajectory; this kernel
        // and the CPU oracle may stop an iteration apart and must still agree to 1e-5.  Once an
        // acceptable iterate exists, up to 3 more iterations are spent on the desired level; if one of
        // them is worse (precision floor of the linear algebra) it is undone -- the direction is still
        // in place -- and the acceptable iterate returned.
        bool polish = false;
        // TOLERANCE ESCALATION (round 4; same rule as oracle/hsde_qp.c solve_one).  On relaxations without strict
        // complementarity (the random MLD of BASELINE configs[4]) the active-set exchange of the polish cycles when the set
        // is read from an iterate of gap 1e-8, and the iterate such a solve returned instead is ~sqrt(gap) off in the
        // trajectory.  When the LAST attempt has failed on the iterate the solve would return, the stopping tolerance
        // drops by 100 (twice at most), the iteration goes on, and the next iterate that meets it gets a last attempt of
        // its own.  The level lives in `attempts` (ATTEMPTS + 2 esc, + 1 once the level's attempt is spent): no state of
        // its own in a loop that is short of registers.  A node whose polish verifies never gets here.
        int action = 0; // 0: go on iterating, 1: leave the loop, 2: undo the last step and leave
        for (;;) {
            constexpr int kEscMax = HMPC_ESC_ENABLED(D) ? HMPC_ESC_MAX : 0;
            const int esc = (kEscMax > 0 && attempts > HMPC_POLISH_ATTEMPTS) ? (attempts - HMPC_POLISH_ATTEMPTS) >> 1 : 0;
            const double tole = esc == 0 ? p.tol : esc == 1 ? 1e-2 * p.tol : 1e-4 * p.tol;
            const double gtol0 = p.tol * (1 + fmin(fabs(pobj), fabs(dob))), gtol = tole * (1 + fmin(fabs(pobj), fabs(dob)));
            const bool acceptable = rcinf / tau <= tole * (1 + winf / tau + x0inf) &&
                                    rdinf / tau <= tole * (1 + zinf / tau) && gap <= gtol;
            // (acceptable at the tolerance the caller asked for: what an escalated solve falls back on)
            const bool acc0 = rcinf / tau <= p.tol * (1 + winf / tau + x0inf) && rdinf / tau <= p.tol * (1 + zinf / tau) && gap <= gtol0;
            // The polish (below) is tried as soon as the iterate is good enough to read the active set from
            // (ptol), once per iterate, and not on an iterate that is about to be undone.
            const double gptol = p.ptol * (1 + fmin(fabs(pobj), fabs(dob)));
            // the barrier parameter is exhausted and the point is optimal to 100 x tol (1e-6, a simplex code's
            // default): nothing more can be gained from the interior-point iteration on an interior-free node
            const bool exh0 = mu < 1e-11 && rcinf / tau <= 100 * p.tol * (1 + winf / tau + x0inf) &&
                              rdinf / tau <= 100 * p.tol * (1 + zinf / tau) && gap <= 100 * gtol0;
            const bool exhausted = status != HMPC_OPTIMAL && exh0;
            // an escalated solve that no longer gets closer returns what it has (acceptable to the caller's tolerance)
            const bool stalled = esc > 0 && status == HMPC_OPTIMAL && !acceptable && (acc0 || exh0) && extra_done >= HMPC_ESC_ITERS;
            // (the iterate this solve would return gets one last attempt even when the regular ones are used up -- they were
            // spent on immature iterates; 5 of 4 000 optimal nodes of the one-wall system at N=40 ended that way)
            const bool desired = gap <= 1e-2 * gtol && rdinf / tau <= 1e-2 * tole * (1 + zinf / tau);
            const bool final_exit = (acceptable && (desired || extra_done >= 3 || it == p.max_iter)) || exhausted || stalled;
            polish = p.polish && !tried &&
                     (attempts < HMPC_POLISH_ATTEMPTS
                          ? (acceptable || exhausted || (status != HMPC_OPTIMAL && rcinf / tau <= p.ptol * (1 + winf / tau + x0inf) &&
                                                         rdinf / tau <= p.ptol * (1 + zinf / tau) && gap <= gptol))
                          : (((attempts - HMPC_POLISH_ATTEMPTS) & 1) == 0 && final_exit));
            if (polish) break; // decided by the passes of the polish
            if (p.polish && tried && attempts > HMPC_POLISH_ATTEMPTS && ((attempts - HMPC_POLISH_ATTEMPTS) & 1) == 1 && final_exit && esc < kEscMax &&
                it < p.max_iter) {
                // the last attempt of this level has failed on the iterate the solve would return: next level
                attempts++;
                extra_done = 0;
                status = HMPC_OPTIMAL;
                continue; // (the exits below are decided at the new tolerance)
            }
            if (acceptable) {
                status = HMPC_OPTIMAL;
                if (desired || extra_done >= 3 || it == p.max_iter) action = 1;
                else extra_done++;
            } else if (exhausted) {
                status = HMPC_OPTIMAL;
                action = 1;
            } else if (status == HMPC_OPTIMAL) {
                if (esc > 0 && (acc0 || exh0)) { // on the way to a tighter tolerance
                    if (extra_done >= HMPC_ESC_ITERS) action = 1;
                    else extra_done++;
                } else action = 2;
            }
            break;
        }
        if (action == 2) {
            for (int o = lane; o < n; o += D::kNT) S.w[o] -= last_alpha * S.w2[o];
            for (int o = lane; o < (T + 1) * nx; o += D::kNT) S.lam[o] -= last_alpha * S.lam2[o];
            for (int o = lane; o < T * nub; o += D::kNT) S.nuf[o] -= last_alpha * S.nuf2[o];
            ROWS_BEGIN(k, rw)
                if (R.D(k, rw.e) != 0.0) {
                    R.z(k, rw.e) -= last_alpha * R.dz(k, rw.e);
                    R.s(k, rw.e) -= last_alpha * R.prod(k, rw.e);
                }
            ROWS_END
            tau -= last_alpha * last_dtau;
            kap -= last_alpha * last_dkap;
            __syncthreads();
            set_prescribed<D>(p, S, lane, tau); // x_0 and the fixed binaries follow tau exactly, also after the way back
            __syncthreads();
        }
        if (action) break;
        // Every infeasibility exit carries a bound on the certificate residual: a ray that is not a proof must
        // not prune a subtree (a node whose tau vanishes without one ends MAXITER / NUMERICAL and is surfaced).
        if (!polish && eta > 0 && (certinf <= p.tol_inf * eta || (tau <= 1e-8 * kap && certinf <= 1e-3 * eta))) {
            status = HMPC_INFEASIBLE;
            break;
        }
        // tau has collapsed but the ray is no proof to tolerance: the node is infeasible by about the accuracy of the
        // linear algebra (its least-violated point misses the rows by ~1e-6; met on the published sd = 0.01 runs).  The
        // embedding's conclusion (tau -> 0 with kappa > 0 and a cost bounded below: no feasible point) is taken -- the
        // node is pruned, as by a simplex code with a 1e-6 feasibility tolerance -- but the ray is flagged WEAK
        // (HMPC_ITERS_WEAK): it prunes this node only, the warm-start shift never carries it to the next step.
        // Same rule as oracle/hsde_qp.c.
        if (!polish && ((eta > 0 && tau <= 1e-8 * kap && certinf <= 0.5 * eta) || tau <= 1e-12 * kap)) {
            // (second clause, round 4: tau has collapsed four more decades and still no ray verifies -- a node on the very
            // boundary between feasible and infeasible: f'y + h'z and E'y + C'z both go to zero, eta changes sign from one
            // iteration to the next.  Met when the published sd = .003 runs are replayed with cold searches only
            // (tests/golden/hard_node_sd003.npz): the kernel ran such a node to 100 iterations, twice, and could end MAXITER;
            // the oracle left at iteration 37 through the clause above by the luck of its rounding.  Same conclusion, same
            // flag: pruned at this step, never carried to the next.  Same rule as oracle/hsde_qp.c.)
            status = HMPC_INFEASIBLE;
            // (2: through the second clause alone -- no ray meets even the loose bound of the first: HMPC_ITERS_UNCERTIFIED, which
            // the drivers count; a numerical collapse of tau on a feasible node would look the same)
            weak_out = (eta > 0 && tau <= 1e-8 * kap && certinf <= 0.5 * eta) ? 1 : 2;
            break;
        }
        if (!polish && it == p.max_iter) break;

        // ---------------- barrier weights / active set ----------------
        // ACTIVE-SET POLISH: the vertex solution of the active set read from the iterate -- what a simplex /
        // crossover code returns; an interior-point iterate that meets the stopping test still carries
        // (dual residual) / (curvature of the stage cost) ~ 1e-5 in the trajectory on nodes without an interior.
        // Rows with z > s are taken as active; the equality-constrained QP on them is solved by the method of
        // multipliers with the machinery at hand -- D = rho on the active rows, D = delta on the others (a
        // proximal term in the metric of the inactive rows: unique step where the cost sees no input), one
        // factorisation per active set and a few solves with the right-hand side of the constant direction
        // shifted by the multipliers:  e = rho h - z (active), delta C w_c (inactive; w_c the proximal centre,
        // which follows the iterate).  Rows that end on the wrong side (negative multiplier / violated) change
        // sides and a new round starts.  The result is exactly complementary and stationary and is taken only
        // if it verifies; otherwise the iteration goes on from the untouched iterate.  Same steps, constants
        // and tolerances as oracle/hsde_qp.c polish().
        LANE_OPAQUE(lane);
        __syncthreads(); // every thread is done reading z from S.e
        if (polish) {
            // active: z > s, or -- Tapia indicators over the last step (dz and ds are still in place) -- the slack
            // shrinks faster than the multiplier: s+/s < z+/z; this reads weakly active rows right far more often than
            // z > s alone (2.2 instead of 3.3 active sets per polish).  Decided in a pass of its own (one bit per
            // slot): folded into the pass below it kept four values per slot alive across the row products.
            unsigned long long amask = 0;
            if (last_alpha > 0) {
                ROWS_BEGIN(k, rw)
                    const double zr = R.z(k, rw.e), sr = R.s(k, rw.e);
                    const double sp = sr - last_alpha * R.prod(k, rw.e), zp = zr - last_alpha * R.dz(k, rw.e);
                    if (sp > 0 && zp > 0 && zr * sp > sr * zp) amask |= 1ull << (k & 63);
                ROWS_END
            }
            ROWS_BEGIN(k, rw)
                const double zr = R.z(k, rw.e); // zero on rows that take no part in this solve
                double d = 0.0;
                if (zr != 0.0) {
                    const bool active = zr > R.s(k, rw.e) || ((amask >> (k & 63)) & 1ull);
                    R.prod(k, rw.e) = zr; // kept for the way back
                    if (active) { d = HMPC_RHO_OF(p.polish_l1); R.dz(k, rw.e) = zr / tau; }
                    else { d = HMPC_POLISH_DELTA; R.dz(k, rw.e) = rm.dot(p, S, k, rw, S.w) / tau; }
                    R.D(k, rw.e) = d;
                }
                S.e[rw.e] = d;
            ROWS_END
            mode = 1; round = 0; al = 0;
            level = p.polish_l1;
            attempts++;
        } else {
            ROWS_BEGIN(k, rw)
                const double zr = R.z(k, rw.e); // zero on inactive rows
                const double d = zr != 0.0 ? zr * frcp(R.s(k, rw.e)) : 0.0;
                R.D(k, rw.e) = d; // the slot holds D from here to the update of this iteration
                S.e[rw.e] = d;    // S.e <- D for the Gram phase of the factorisation
            ROWS_END
        }
        __syncthreads();
        STAMP(1);
      } // mode == 0

        // ---------------- factorisation ----------------
        if (mode != 2) {
            int frc;
            if constexpr (D::kNX > 0) frc = factor_reg<D>(p, S, lane FSTAMP_PASS);
            else frc = factor<D>(p, S, lane FSTAMP_PASS);
            STAMP(2);
            if (frc != 0) {
                if (mode == 0) {
                    ROWS_BEGIN(k, rw)
                        R.z(k, rw.e) = R.D(k, rw.e) * R.s(k, rw.e); // the slots hold z again
                    ROWS_END
                    if (status != HMPC_OPTIMAL) status = HMPC_NUMERICAL;
                    break;
                }
                mode = 3; // the polish gives up
            } else if (mode == 1) {
                // the solves consume the factorisation's mb in S.g: keep a copy (S.rd is recomputed anyway)
                for (int o = lane; o < T * nz; o += D::kNT) S.rd[o] = S.g[o];
            }
        }
        // (register kernels, interior-point iterations: the constant direction and the predictor's affine direction in ONE pair
        // of sweeps, kkt_solve_reg_pair; -DHMPC_PAIR=0: two solves as before)
        constexpr bool kPair = D::kNX > 0 && HMPC_PAIR != 0;
        static_assert(!kPair || HMPC_STABLE_DEN, "the paired solve hands the denominator's row sum back");
        double q2_pair = 0.0;
        bool paired = false;
        if (mode != 3) {
            if constexpr (kPair) {
                if (mode == 0) {
                    q2_pair = kkt_solve_reg_pair<D, RS, RM>(p, S, R, rm, lane, tau FSTAMP_PASS);
                    paired = true;
                }
            }
            if (!paired) {
                LANE_OPAQUE(lane);
                ROWS_BEGIN(k, rw)
                    const double d = R.D(k, rw.e);
                    double v = d * rm.h(p, S, k, rw); // right-hand side of the constant direction
                    if (mode != 0) v = d >= 1.0 ? v - R.dz(k, rw.e) : d * R.dz(k, rw.e);
                    S.e[rw.e] = v;
                ROWS_END
                if (mode == 2)
                    for (int o = lane; o < T * nz; o += D::kNT) S.g[o] = S.rd[o];
                __syncthreads();

                // ---------------- constant direction: rhs = (0 ; f ; h) ----------------
                kkt_dispatch<D, RS>(p, S, R, rm, lane, nullptr, 0.0, true, nullptr, 0.0, true, S.w1, S.lam1, S.nuf1 FSTAMP_PASS);
            }
        }
        if (mode != 0) {
            // ---------------- polish: multiplier step, verification, next pass ----------------
            int outcome = 0; // 0 give up, 1 verified, 2 next pass
            // (own: the second launch of the two-launch form -- the set is the node's OWN masked solve's, and what is missing
            // are the terminal rows that solve violates: they enter over the rounds, whatever their violation to begin with)
            const int max_rounds = (WARM && warm_try) ? (own ? HMPC_POLISH_ROUNDS_OWN : HMPC_POLISH_ROUNDS_WARM)
                                                      : attempts > HMPC_POLISH_ATTEMPTS ? HMPC_POLISH_ROUNDS_LAST : HMPC_POLISH_ROUNDS;
            if (mode != 3) {
                double pinf = 0, pmove = 0;
                ROWS_BEGIN(k, rw)
                    const double d = R.D(k, rw.e);
                    if (d >= 1.0) { // multiplier step
                        const double zn = S.e[rw.e];
                        pinf = fmax(pinf, fabs(zn - R.dz(k, rw.e)));
                        R.dz(k, rw.e) = zn;
                    } else if (d != 0.0) { // the proximal centre follows the iterate
                        const double a = rm.dot(p, S, k, rw, S.w1);
                        pmove = fmax(pmove, fabs(a - R.dz(k, rw.e)));
                        R.dz(k, rw.e) = a;
                    }
                ROWS_END
                {
                    double v[2] = {pinf, pmove};
                    const int op[2] = {1, 1};
                    block_reduce<D, 2>(v, op, S.red, lane);
                    pinf = v[0] / HMPC_RHO_OF(level); pmove = v[1];
                }
                // (the second level gets twice the steps: it is there for the slow sets)
                if (!((al >= 1 && pinf <= 1e-12 * (1 + winf / tau) && HMPC_POLISH_DELTA * pmove <= 1e-13) || al == (level == 1 ? 2 * HMPC_POLISH_ITERS : HMPC_POLISH_ITERS) - 1)) {
                    al++;
                    mode = 2;
                    outcome = 2;
                } else {
                    // what next: -1 give up, 0 verified, 1 rows on the wrong side change sides, 2 other penalty level
                    int act = -1;
                    double vmax = 0, zmin = 0;
                    const double ez = 1e-11, es = 1e-9 * (1 + winf / tau);
                    if (pinf <= (level == 0 ? 1e-12 : 1e-10) * (1 + winf / tau) && HMPC_POLISH_DELTA * pmove <= 1e-12 * (1 + zinf / tau)) {
                        // The active rows are met and the proximal term -- dropped from the multipliers, hence the
                        // stationarity residual of the result -- has died out.  Sign of the multipliers, slack of the
                        // inactive rows: the sign test is absolute and tight (a clipped negative multiplier costs |z| /
                        // curvature of the stage cost in the trajectory, a violated inactive row only its violation).
                        ROWS_BEGIN(k, rw)
                            const double d = R.D(k, rw.e);
                            if (d >= 1.0) zmin = fmin(zmin, R.dz(k, rw.e));
                            else if (d != 0.0) vmax = fmax(vmax, R.dz(k, rw.e) - rm.h(p, S, k, rw));
                        ROWS_END
                        {
                            double v[2] = {vmax, zmin};
                            const int op[2] = {1, 2};
                            block_reduce<D, 2>(v, op, S.red, lane);
                            vmax = v[0]; zmin = v[1];
                        }
                        if (vmax <= es && zmin >= -ez) {
                            // verified; at the second level: the same active set once more at the first, from these
                            // multipliers (what is left to settle are the components that matter, of the size of the
                            // second level's rounding; the slow ones are in place)
                            act = (level == 1 && !p.polish_l1 && round + 1 < max_rounds) ? 2 : 0;
                        } else if (WARM && warm_try && !own && vmax > HMPC_POLISH_WARM_VMAX * (1 + winf / tau)) {
                            // the handed-down set is not near this node's optimum (the node is infeasible, or fixing the
                            // binary moved the solution): dropped after this one factorisation
                        } else if (round + 1 < max_rounds) {
                            // Rows on the wrong side change sides: of the violated inactive rows only those within a factor
                            // two of the worst violation (a missing active row drags others across their bounds; the next
                            // round shows which are real), of the active rows EVERY one with a negative multiplier (with the
                            // factor-two rule there as well, the most negative multiplier of the random MLD's relaxations only
                            // halved per round and 28 % of the optimal nodes of BASELINE configs[4] never verified; now -- with ten
                            // rounds in the last attempt -- 3 %, in fewer factorisations; the cart-pole systems take the same rounds as before).
                            act = 1;
                        }
                    } else if (level == 0 && pinf == pinf && round + 1 < max_rounds) {
                        // The multiplier steps contract by 1 / (1 + rho lambda), lambda the eigenvalues of C_A Phi^-1 C_A':
                        // active rows that nearly depend on each other do not settle at the first level.  They do at the
                        // second -- not the first choice: eps rho is no longer below the proximal weight there (1e-9 in the
                        // multipliers, ~1e-6 in the trajectory) --, used once per attempt where the first fails; its result
                        // goes through the first level once more (level 2).  Same as oracle/hsde_qp.c polish().
                        act = 2;
                    }
                    if (act == 0) {
                        outcome = 1;
                    } else if (act > 0) {
                        if (act == 2) level++;
                        const double rho = HMPC_RHO_OF(level);
                        const bool flip = act == 1;
                        round++;
                        ROWS_BEGIN(k, rw)
                            double d = R.D(k, rw.e);
                            if (d != 0.0) {
                                if (d < 1.0) {
                                    if (flip && vmax > es && R.dz(k, rw.e) - rm.h(p, S, k, rw) > 0.5 * vmax) { d = rho; R.dz(k, rw.e) = 0.0; }
                                } else {
                                    d = (flip && zmin < -ez && R.dz(k, rw.e) < 0.0) ? HMPC_POLISH_DELTA : rho;
                                }
                                R.D(k, rw.e) = d;
                                // next pass: the proximal centre starts at the interior-point iterate again
                                if (d == HMPC_POLISH_DELTA) R.dz(k, rw.e) = rm.dot(p, S, k, rw, S.w) / tau;
                            }
                            S.e[rw.e] = d;
                        ROWS_END
                        __syncthreads();
                        al = 0;
                        mode = 1;
                        outcome = 2;
                    }
                }
            }
            if (outcome == 2) continue;
            if (outcome == 1) { // the polished point replaces the iterate (tau = 1 units)
                for (int o = lane; o < n; o += D::kNT) S.w[o] = S.w1[o];
                for (int o = lane; o < (T + 1) * nx; o += D::kNT) S.lam[o] = S.lam1[o];
                for (int o = lane; o < T * nub; o += D::kNT) S.nuf[o] = S.nuf1[o];
                ROWS_BEGIN(k, rw)
                    const double d = R.D(k, rw.e);
                    if (d != 0.0) R.z(k, rw.e) = d >= 1.0 ? fmax(R.dz(k, rw.e), 0.0) : 0.0;
                ROWS_END
                tau = 1.0;
                status = HMPC_OPTIMAL;
                polished_out = true;
                handed_out = WARM && warm_try;
                __syncthreads();
                break;
            }
            // not verified: the iterate is intact but for the slots that held the classes; its residuals are
            // recomputed (S.rd served as scratch) and the interior-point iteration goes on
            ROWS_BEGIN(k, rw)
                if (R.D(k, rw.e) != 0.0) R.z(k, rw.e) = R.prod(k, rw.e);
            ROWS_END
            if (WARM && warm_try) { // the hand-down did not verify: back to the cold start (multipliers and slacks are at 1 again)
                if (attempt_only) { // (only an attempt: the caller runs the regular sequence of solves)
                    status = HMPC_RETRY;
                    break;
                }
                warm_try = false;
                for (int o = lane; o < n; o += D::kNT) S.w[o] = 0.0;
                __syncthreads();
                set_prescribed<D>(p, S, lane, tau);
                __syncthreads();
                mode = 0;
                continue;
            }
            tried = true;
            mode = 0;
            continue;
        }
        tried = false;
        STAMP(3);
        // Denominator of the tau step:  kap/tau + w'Pw/tau^2 - 2 w'P w1/tau - (f'y1 + h'z1).  The constant direction solves
        // P w1 + E'y1 + C'z1 = 0, E w1 = f, C w1 - z1/D = h, hence -(f'y1 + h'z1) = w1'P w1 + z1'D^-1 z1 and
        //     den = kap/tau + (w/tau - w1)' P (w/tau - w1) + sum_i z1_i^2 / D_i :
        // a sum of nonnegative terms.  As the difference of four O(1 .. 100) terms (until round 4) it lost everything in the
        // last iterations -- the quadratic terms go to zero like kap/tau -- once the constant direction was only good to
        // 1e-8: on a deep node of BASELINE configs[4] den came out ten times too small at mu = 3e-11, tau fell from 0.20 to
        // 0.046 in one step and the dual residual rose from 5e-8 to 3e-5 (profiles/r04_den_cancellation.txt).
        double den;
        if constexpr (HMPC_STABLE_DEN) {
            double q1 = dPd<D>(p, S, lane, S.w1, 1.0 / tau), q2 = q2_pair;
            if (!paired) {
                ROWS_BEGIN(k, rw)
                    const double d = R.D(k, rw.e);
                    if (d != 0.0) { const double z1 = S.e[rw.e]; q2 += z1 * z1 * frcp(d); } // (S.e: the constant direction's dz)
                ROWS_END
            }
            double v[2] = {q1, q2};
            const int op[2] = {0, 0};
            block_reduce<D, 2>(v, op, S.red, lane);
            den = kap / tau + v[0] + v[1];
        } else {
            double g1 = wPv<D>(p, S, lane, S.w1);
            double fyhz1 = lin_obj<D>(p, S, rm, lane, S.lam1, S.nuf1, S.e);
            double v[2] = {g1, fyhz1};
            const int op[2] = {0, 0};
            block_reduce<D, 2>(v, op, S.red, lane);
            den = kap / tau + wPw / (tau * tau) - v[0] * 2.0 / tau - v[1];
        }
#ifdef HMPC_TRACE_STEP
        dbg_den = den / (kap / tau);
#endif

        double dtau_a = 0, dkap_a = 0, sigma = 0;
        for (int pass = 0; pass < 2; pass++) {
            const double lin = pass == 0 ? 1.0 : 1.0 - sigma;
            const double dkap_rhs = tau * kap + (pass ? dtau_a * dkap_a - sigma * mu : 0.0);
            LANE_OPAQUE(lane);
            if (!(paired && pass == 0)) { // (the affine direction of a paired iteration is in place already)
                __syncthreads();
                ROWS_BEGIN(k, rw)
                    double v = 0;
                    const double d = R.D(k, rw.e);
                    if (d != 0.0) { // active row
                        const double sr = R.s(k, rw.e), zr = d * sr;
                        const double rc = sr - rm.h(p, S, k, rw) * tau + rm.dot(p, S, k, rw, S.w); // row residual
                        const double dsr = sr * zr + (pass ? R.prod(k, rw.e) - sigma * mu : 0.0);
                        v = dsr * frcp(sr) - lin * (d * rc); // d (-lin rc + dsr / z) with z = d s
                    }
                    S.e[rw.e] = v;
                ROWS_END
                __syncthreads();
                STAMP(4);
                kkt_dispatch<D, RS>(p, S, R, rm, lane, S.rd, -lin, false, S.rdyn, -lin, false, S.w2, S.lam2, S.nuf2 FSTAMP_PASS);
                STAMP(3);
            }
            double g2 = 0;
            g2 = wPv<D>(p, S, lane, S.w2);
            double fyhz2 = lin_obj<D>(p, S, rm, lane, S.lam2, S.nuf2, S.e);
            {
                double v[2] = {g2, fyhz2};
                const int op[2] = {0, 0};
                block_reduce<D, 2>(v, op, S.red, lane);
                g2 = v[0] * 2.0 / tau; fyhz2 = v[1];
            }
            const double dtau = (lin * rg - dkap_rhs / tau + g2 + fyhz2) / den;
            const double dkap = -(dkap_rhs + kap * dtau) / tau;
            // combined direction d = v2 + dtau v1
            for (int o = lane; o < n; o += D::kNT) S.w2[o] += dtau * S.w1[o];
            for (int o = lane; o < (T + 1) * nx; o += D::kNT) S.lam2[o] += dtau * S.lam1[o];
            for (int o = lane; o < T * nub; o += D::kNT) S.nuf2[o] += dtau * S.nuf1[o];
            ROWS_BEGIN(k, rw)
                double v = 0.0;
                const double d = R.D(k, rw.e);
                if (d != 0.0) {
                    // multiplier step of the constant direction: D (C w1 - h)
                    const double z1 = d * (rm.dot(p, S, k, rw, S.w1) - rm.h(p, S, k, rw));
                    v = S.e[rw.e] + dtau * z1;
                }
                S.e[rw.e] = v;
                R.dz(k, rw.e) = v;
            ROWS_END
            __syncthreads();
            // Iterative refinement against the three linear blocks of the Newton system at this dtau:
            // one step once mu < 1e-3, two once mu < 1e-7 (one step squares the relative error of a
            // solve; the stage cost's small curvature needs the dual residual well below the stopping
            // tolerance for the trajectory to be accurate to 1e-5).
            // (Refinement from iteration HMPC_REFINE_FROM_IT on, or once a polish has been tried and did not verify -- from
            // then on the iterate itself may be the answer.  The nodes that end before, nearly all, take the same
            // iterations with and without it and return the polished point or a ray: measured, 6.46 -> 6.12 ms.)
            const int nref = (pass == 1 && p.refine && (it >= HMPC_REFINE_FROM_IT || attempts > 0 || !p.polish)) ? (mu < 1e-7 ? 2 : mu < 1e-3 ? 1 : 0) : 0;
            for (int rf = 0; rf < nref; rf++) {
                LANE_OPAQUE(lane);
                if (rf > 0) { // S.e <- current dz for the C' dz products (the previous round left its correction there)
                    ROWS_BEGIN(k, rw)
                        S.e[rw.e] = R.dz(k, rw.e);
                    ROWS_END
                    __syncthreads();
                }
                // residual of the three linear blocks at the combined direction (x_0 and fixed
                // binaries are met by construction), then one correction solve
                term_cols<D, RM>(p, S, lane, S.e);
                for (int o = lane; o < n; o += D::kNT) {
                    const int t = o / nz < T ? o / nz : T;
                    const int j = o - t * nz, dim = t < T ? nz : nx;
                    const ldsd *PP = t < T ? S.P : S.PT;
                    double a = -lin * S.rd[o];
                    for (int l = 0; l < dim; l++) a -= PP[j * dim + l] * S.w2[t * nz + l];
                    if (t == T) {
                        a -= S.lam2[T * nx + j];
                    } else {
                        if (j < nx) a -= S.lam2[t * nx + j];
                        for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + j] * S.lam2[(t + 1) * nx + l];
                        a -= ccol_dot<D>(p, S, t, j, S.e);
                        if (j >= nx + nuc && S.fix[t * nub + (j - nx - nuc)] >= 0) a = 0.0;
                        if (t == 0 && j < nx) a = 0.0;
                    }
                    S.g[o] = a;
                }
                for (int o = lane; o < T * nx; o += D::kNT) {
                    const int t = o / nx, i = o - t * nx;
                    double a = -lin * S.rdyn[o] - S.w2[(t + 1) * nz + i];
                    for (int l = 0; l < nz; l++) a += S.AB[i * AB_STRIDE + l] * S.w2[t * nz + l];
                    S.edyn[o] = a;
                }
                __syncthreads();
                ROWS_BEGIN(k, rw)
                    double v = 0;
                    const double d = R.D(k, rw.e);
                    if (d != 0.0) {
                        const double sr = R.s(k, rw.e), zr = d * sr, hh = rm.h(p, S, k, rw);
                        const double rc = sr - hh * tau + rm.dot(p, S, k, rw, S.w);
                        const double dsr = sr * zr + R.prod(k, rw.e) - sigma * mu;
                        // d (-lin rc + dsr / z + dtau h + dz s / z - C w2) with z = d s
                        const double a = -lin * rc + dtau * hh - rm.dot(p, S, k, rw, S.w2);
                        v = d * a + (dsr * frcp(sr) + S.e[rw.e]);
                    }
                    S.e[rw.e] = v;
                ROWS_END
                __syncthreads();
                STAMP(5);
                kkt_dispatch<D, RS>(p, S, R, rm, lane, S.g, 1.0, false, S.edyn, 1.0, false, S.w1, S.lam1, S.nuf1 FSTAMP_PASS);
                STAMP(3);
                for (int o = lane; o < n; o += D::kNT) S.w2[o] += S.w1[o];
                for (int o = lane; o < (T + 1) * nx; o += D::kNT) S.lam2[o] += S.lam1[o];
                for (int o = lane; o < T * nub; o += D::kNT) S.nuf2[o] += S.nuf1[o];
                ROWS_BEGIN(k, rw)
                    if (R.D(k, rw.e) != 0.0) R.dz(k, rw.e) += S.e[rw.e];
                ROWS_END
            }
            // Multipliers of the fixed binaries from the step that is ACTUALLY taken (run-time-sized kernels; round 4).  Their
            // stationarity row defines them, and the refinement above leaves that row alone; but what was accumulated --
            // nuf2 + dtau nuf1 (+ corrections) -- goes with the multiplier steps as the SOLVES returned them, while the step
            // uses the constant direction's dz recomputed as D (C w1 - h) (it is not stored): the two differ by the rounding of
            // D (C w1) against D h, ~eps D, and with D = z / s at 1e9 .. 1e12 in the last iterations the fixed binaries' rows of
            // the dual residual stood at 1e-7 .. 4e-6 on deep nodes of BASELINE configs[4] (every other row: 1e-13) -- above
            // the stopping tolerance, so that such nodes left through the exhausted-barrier exit with a gap of 1e-6, or not
            // at all once the tolerance escalation asked for 1e-10.  (The oracle stores that dz.  Until round 4 the register
            // kernels kept the accumulated form -- the cart-pole's D stays four orders smaller --; a register kernel compiled
            // for another system showed the same defect, see the top of this file: every kernel now takes them from the step.)
            {
                if (pass == 1) {
                    __syncthreads();
                    if (nref > 0) { // (S.e holds the last correction, not the step)
                        ROWS_BEGIN(k, rw)
                            S.e[rw.e] = R.D(k, rw.e) != 0.0 ? R.dz(k, rw.e) : 0.0;
                        ROWS_END
                        __syncthreads();
                    }
                    term_cols<D, RM>(p, S, lane, S.e); // (register kernels: the terminal rows' part of the column products below)
                    for (int o = lane; o < T * nub; o += D::kNT) {
                        if (S.fix[o] >= 0) {
                            const int t = o / nub, c = nx + nuc + (o - t * nub);
                            double a = -lin * S.rd[t * nz + c];
                            for (int j = 0; j < nz; j++) a -= S.P[c * nz + j] * S.w2[t * nz + j];
                            a -= ccol_dot<D>(p, S, t, c, S.e);
                            for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + c] * S.lam2[(t + 1) * nx + l];
                            S.nuf2[o] = a;
                        }
                    }
                    for (int i = lane; i < nx; i += D::kNT) { // (lam_0: the other multiplier a stationarity row defines)
                        double a = -lin * S.rd[i];
                        for (int j = 0; j < nz; j++) a -= S.P[i * nz + j] * S.w2[j];
                        a -= ccol_dot<D>(p, S, 0, i, S.e);
                        for (int l = 0; l < nx; l++) a += S.AB[l * AB_STRIDE + i] * S.lam2[nx + l];
                        S.lam2[i] = a;
                    }
                    __syncthreads();
                }
            }
            // slack step from the complementarity row ; step to the boundary
            LANE_OPAQUE(lane);
            double amax = 1e30;
            if (dtau < 0) amax = fmin(amax, -tau / dtau);
            if (dkap < 0) amax = fmin(amax, -kap / dkap);
            ROWS_BEGIN(k, rw)
                if (R.D(k, rw.e) != 0.0) {
                    const double dz = R.dz(k, rw.e), sr = R.s(k, rw.e), zr = R.D(k, rw.e) * sr;
                    const double dsr = sr * zr + (pass ? R.prod(k, rw.e) - sigma * mu : 0.0);
                    const double ds = -(dsr + sr * dz) * frcp(zr);
                    if (dz < 0) amax = fmin(amax, -zr * frcp(dz));
                    if (ds < 0) amax = fmin(amax, -sr * frcp(ds));
                    if (pass == 0) R.prod(k, rw.e) = ds * dz;
                    else R.prod(k, rw.e) = ds; // the affine product is consumed: keep the slack step here (also for an undo)
                }
            ROWS_END
            amax = block_min<D>(amax, S.red, lane);
            if (pass == 0) {
                const double aa = fmin(1.0, amax);
                sigma = (1 - aa) * (1 - aa) * (1 - aa);
                dtau_a = dtau;
                dkap_a = dkap;
            } else {
                // fraction to the boundary: 0.99; in the END GAME of an infeasible node -- tau below kappa and falling: the step wants
                // tau -> 0 exactly and the certificate's residual goes with tau -- it follows the barrier, 1 - max(mu, 1e-5): one iteration
                // less on every infeasible node, optimal nodes bit for bit as before (oracle/hsde_qp.c, same place)
                const double alpha = fmin(1.0, ((dtau < 0 && tau < kap) ? fmax(0.99, 1.0 - fmax(mu, 1e-5)) : 0.99) * amax);
                __syncthreads();
                for (int o = lane; o < n; o += D::kNT) S.w[o] += alpha * S.w2[o];
                for (int o = lane; o < (T + 1) * nx; o += D::kNT) S.lam[o] += alpha * S.lam2[o];
                for (int o = lane; o < T * nub; o += D::kNT) S.nuf[o] += alpha * S.nuf2[o];
                ROWS_BEGIN(k, rw)
                    if (R.D(k, rw.e) != 0.0) { // the slot holds z again from here on
                        R.z(k, rw.e) = R.D(k, rw.e) * R.s(k, rw.e) + alpha * R.dz(k, rw.e);
                        R.s(k, rw.e) += alpha * R.prod(k, rw.e);
                    }
                ROWS_END
                tau += alpha * dtau;
                kap += alpha * dkap;
                last_alpha = alpha; last_dtau = dtau; last_dkap = dkap;
                __syncthreads();
                set_prescribed<D>(p, S, lane, tau);
                __syncthreads();
            }
        }
        STAMP(6);
        if (!(tau > 0) || !(kap >= 0)) { status = HMPC_NUMERICAL; break; }
        it++;
    }
#ifdef HMPC_STAMPS
    // (`trace` points at this solve's iteration rows: the stamp block follows the rows of both solves)
    double *tb = trace ? trace - (term_on ? 64 * 8 : 0) + 2 * 64 * 8 : nullptr;
    if (tb && lane == 0)
        for (int k = 0; k < 8; k++) tb[(term_on ? 8 : 0) + k] = (double)tacc[k];
    if (tb && lane == 0 && (term_on == 0 || p.nT == 0))
        for (int k = 0; k < 16; k++) tb[16 + k] = (double)facc[k];
#endif
    iters = it;
    tau_out = tau;
    return status;
}

// Largest value of (scaled terminal row) - h at the current (optimal) iterate.
template <class D> DEV double terminal_violation(const DevProb &p, const Lds &S, int lane, double tau)
{
    double tv = -1e300;
    for (int k = lane; k < p.nT; k += D::kNT) {
        const double a = crow_dot<D>(p, S, p.mreg + k, S.w + (p.T - 1) * D::nz(p)) - p.ht[k] * tau;
        tv = fmax(tv, a / tau);
    }
    return block_max<D>(tv, S.red, lane);
}

// Output record in the reference's conventions (subproblem_solution.py:68-168).  S.e holds z here.
template <class D>
DEV void write_record(const DevProb &p, const Lds &S, int lane, int status, double tau, int qp, const DevOut &out)
{
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nub = D::nub(p), M = p.M;
    const int nmu = (T - 1) * p.nc + p.ncL;
    const bool inf = status == HMPC_INFEASIBLE;
    double scale;
    if (inf) {
        double big = 0;
        for (int r = lane; r < M; r += D::kNT) big = fmax(big, S.e[r]);
        for (int o = lane; o < (T + 1) * nx; o += D::kNT) big = fmax(big, fabs(S.lam[o]));
        for (int o = lane; o < T * nub; o += D::kNT) big = fmax(big, fabs(S.nuf[o]));
        scale = 1.0 / block_max<D>(big, S.red, lane); // Farkas ray: scale is arbitrary
    } else {
        scale = 1.0 / (tau * p.cs);
    }
    double *dual = out.dual ? out.dual + (size_t)qp * p.n_dual : nullptr;
    double *prim = out.primal ? out.primal + (size_t)qp * p.n_primal : nullptr;
    const int o_mu = (T + 1) * nx, o_lb = o_mu + nmu, o_ub = o_lb + T * nub, o_rho = o_ub + T * nub;
    const int o_sig = o_rho + T * p.nq + p.nqT;
    double farkas = 0;
    for (int o = lane; o < (T + 1) * nx; o += D::kNT) {
        const double v = S.lam[o] * scale;
        if (dual) dual[o] = v;
        if (o < nx) farkas -= S.x0[o] * v;
    }
    for (int r = lane; r < M; r += D::kNT) {
        int t, lr;
        row_decode(p, r, t, lr);
        if (lr < p.nc || lr >= p.mreg) { // [F G] rows and terminal-set rows are the reference's mu_t
            const double sc = lr < p.nc ? p.reg.scale[lr] : p.sct[lr - p.mreg];
            const double v = S.e[r] * scale * sc;
            if (dual) dual[o_mu + t * p.nc + (lr < p.nc ? lr : p.nc + (lr - p.mreg))] = v;
            farkas -= (hrow<D>(p, S, lr) / sc) * v;
        }
    }
    for (int o = lane; o < T * nub; o += D::kNT) {
        const int t = o / nub, b = o - t * nub;
        double lo, hi;
        if (S.fix[o] < 0) {
            lo = S.e[t * p.mreg + p.nc + b] * scale;
            hi = S.e[t * p.mreg + p.nc + nub + b] * scale;
            farkas -= hi;
        } else {
            const double v = S.nuf[o] * scale;
            hi = v > 0 ? v : 0.0;
            lo = v < 0 ? -v : 0.0;
            farkas -= S.fix[o] * (hi - lo);
        }
        if (dual) { dual[o_lb + o] = lo; dual[o_ub + o] = hi; }
    }
    farkas = block_sum<D>(farkas, S.red, lane);
    double cost = 0, dq = 0;
    if (inf) {
        if (prim) for (int o = lane; o < p.n_primal; o += D::kNT) prim[o] = __longlong_as_double(0x7ff8000000000000LL);
        if (dual) for (int o = o_rho + lane; o < p.n_dual; o += D::kNT) dual[o] = 0.0;
    } else {
        if (prim) {
            for (int o = lane; o < (T + 1) * nx; o += D::kNT) prim[o] = S.w[(o / nx) * nz + (o % nx)] / tau;
            for (int o = lane; o < T * nu; o += D::kNT) prim[(T + 1) * nx + o] = S.w[(o / nu) * nz + nx + (o % nu)] / tau;
        }
        for (int o = lane; o < T * p.nq + p.nqT; o += D::kNT) {
            const int t = (p.nq > 0 && o / p.nq < T) ? o / p.nq : T;
            const int r = o - t * p.nq;
            const double *QQ = t < T ? p.Q : p.QT;
            double a = 0;
            for (int j = 0; j < nx; j++) a += QQ[r * nx + j] * S.w[t * nz + j];
            a /= tau;
            if (dual) dual[o_rho + o] = 2 * a;
            cost += a * a;
        }
        for (int o = lane; o < T * p.nr; o += D::kNT) {
            const int t = o / p.nr, r = o - t * p.nr;
            double a = 0;
            for (int j = 0; j < nu; j++) a += p.R[r * nu + j] * S.w[t * nz + nx + j];
            a /= tau;
            if (dual) dual[o_sig + o] = 2 * a;
            cost += a * a;
        }
        cost = block_sum<D>(cost, S.red, lane);
        dq = 4.0 * cost; // sum rho^2 + sigma^2
    }
    if (lane == 0) {
        if (out.obj) out.obj[qp] = inf ? __longlong_as_double(0x7ff0000000000000LL) : cost;
        if (out.dual_obj) out.dual_obj[qp] = inf ? farkas : -0.25 * dq + farkas;
        if (out.status) out.status[qp] = status;
    }
}

// KF / KB / KT: register slots of the static row map ([F G] rows, bound rows, terminal-set rows) for the
// compile-time shapes; all zero for the generic kernel (list row map, rows in the global slab).
template <int NX_, int NU_, int NUB_, int KF, int KB, int KT, int NW, bool WARM = false>
__global__ void __launch_bounds__(NW * WAVE) HMPC_KERNEL_ATTR
hmpc_qp_kernel(const DevProb p_arg, const double *__restrict__ x0g, int x0_stride, const int8_t *__restrict__ fixg, int B,
               const DevOut out, double *__restrict__ rows_ws, double *__restrict__ trace, const int32_t *__restrict__ order, const DevWarm warm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef HMPC_SIZED
    // A kernel compiled for ONE problem (hmpc_jit.h): the integer sizes of DevProb are constants of the translation unit --
    // the same code paths with immediates for strides, trip counts and divisions (time of the run-time-sized kernels is the
    // instruction count of wave 0).  The host launches it only with the DevProb the constants were taken from.
    DevProb p = p_arg;
    HMPC_SIZED(p)
#else
    const DevProb &p = p_arg;
#endif
    constexpr int RS = KF + KB + KT;
#if defined(HMPC_SIZED) && !defined(HMPC_DPP_FEW)
    // A kernel compiled with the problem's sizes has the registers for the DPP broadcasts whatever its number of row slots
    // (N = 40, two waves: +8.5 %; a shape compiled at hmpc_create, nx = 6, nu = 2 + 3: +11 %; one-wave kernel of the headline:
    // 215 AGPRs, no scratch -> 92 B of scratch, +4.6 % -- but that kernel gains more from the compiler's ILP schedule, which
    // does not go together with the DPP form: hmpc_jit.h compiles it with -DHMPC_DPP_FEW; profiles/r04_dpp_ab.txt, r04_sched_ab.txt)
    typedef Dims<NX_, NU_, NUB_, NW, (NX_ > 0 && RS > 0)> D;
#else
    typedef Dims<NX_, NU_, NUB_, NW, (NX_ > 0 && RS > 0 && RS <= 8)> D;
#endif
#ifdef HMPC_SIZED
    // (a sized kernel may keep the row state of the list row map in registers: KF = Mpad / (64 NW) slots, KB = KT = 0)
    static_assert(NX_ <= 0 || RS > 0, "a sized register kernel keeps its row slots");
#else
    static_assert((NX_ > 0) == (RS > 0), "compile-time shapes use the static row map, the generic kernels the lists");
#endif
    typedef typename std::conditional<(NX_ > 0), RowMapS<D, KF, KB, KT>, RowMapL<D>>::type RM;
    const int lane = threadIdx.x; // thread of the workgroup; wave 0 (lane < 64) runs the recursions
    const int nx = D::nx(p), nu = D::nu(p), nz = D::nz(p), T = p.T, nub = D::nub(p), M = p.M, n = T * nz + nx, ne = D::ne(p);
    Lds S;
    {
        // the carve below must mirror hmpc_lds_bytes() in hmpc_device.h
        ldsd *q = (ldsd *)smem;
        auto take = [&](int cnt) { ldsd *r = q; q += cnt; return r; };
        S.w = take(n); S.lam = take((T + 1) * nx); S.nuf = take(T * nub);
        S.e = take(M);
        const int nxs = nx * (nx + 1) / 2;
        S.Lm = take(D::kBig ? 0 : T * LM_STAGE(nx, nu)); S.dinv = take(T * nu); S.Pr = take(D::kBig ? 0 : (T + 1) * nxs);
        S.LmG = p.fac_ws + (size_t)blockIdx.x * p.fac_stride; S.PrG = S.LmG + (size_t)T * LM_STAGE(nx, nu);
        S.rd = take(n); S.rdyn = take(T * nx); S.edyn = take((T + 1) * nx); S.g = take(n); S.pv = take((T + 1) * nx); // (edyn: one stage longer, kkt_solve_reg_pair keeps a p_t there)
        S.w1 = take(n); S.lam1 = take((T + 1) * nx); S.nuf1 = take(T * nub);
        // the second direction is dead while a factorisation runs: its storage doubles as the
        // factorisation scratch (stage matrix, carried identity block, Pn [A B])
        // (the register factorisation keeps the dense terminal block of the last stage behind the stage matrix)
        const int dir2 = n + (T + 1) * nx + T * nub, fscr = nz * nz + (D::kKC > 0 ? nz * nz : nx * nz);
        S.w2 = take(dir2 > fscr ? dir2 : fscr); S.lam2 = S.w2 + n; S.nuf2 = S.lam2 + (T + 1) * nx;
        S.Mm = S.w2; S.PA = S.Mm + nz * nz;
        S.q = take((nx + 3) / 4 * 4); S.mv = take(nz); S.red = take(40);
        S.x0 = take(nx);
        // streaming form: [A B] padded (zero column nz, zero rows nx ..), stage blocks of multipliers padded to
        // lrows x nup (zero beyond a row's entries) -- must mirror hmpc_lds_bytes() in hmpc_device.h
        S.abst = nz; S.nup = 0; S.lrows = 0;
        int abrows = nx;
        if constexpr (D::kBig) {
            const int nur = (nu + 3) / 4 * 4;
            S.nup = (nu + 1 + 3) / 4 * 4;
            S.lrows = nz + 1 > nx + nur ? nz + 1 : nx + nur;
            S.abst = S.lrows;
            abrows = (nx + 1 + 3) / 4 * 4;
        }
        S.AB = take(abrows * S.abst); S.P = take(nz * nz); S.PT = take(nx * nx);
        // lists of the regular stage: the generic kernel stages rows, columns and Gram lists; the
        // compile-time shapes padded columns and Gram lists (their rows live in registers)
        constexpr int KC = D::kKC;
        constexpr bool NL = KC > 0 || D::kBig; // no row / column lists in LDS
        ldsd *h0 = take(D::kBig ? 0 : p.mreg), *rval0 = take(NL ? 0 : p.nnz0), *cval0 = take(KC ? nz * KC : NL ? 0 : p.nnz0);
        ldsd *gval0 = take(D::kBig ? 0 : p.nng0);
        // (the dense / singleton split of the stage rows belongs to the run-time-sized kernels: a register kernel of a problem
        // with nz >= 16 -- DevProb::split_lds is set from the sizes alone -- carved these arrays too, which hmpc_lds_bytes()
        // does not count for it: everything behind them lay beyond the workgroup's LDS, where loads return zero and stores are
        // dropped.  That was the "defect of the register recursions at nz = 16" of round 4.)
        const bool split = KC == 0 && p.split_lds;
        ldsd *cdn0 = take(split ? p.ndp * nz : 0), *sval0 = take(split ? p.mreg : 0);
        S.Ls = take(D::kBig ? 2 * p.ring * S.lrows * S.nup : 0); // two chunk buffers of `ring` padded stage blocks (kkt_sweeps_wave)
        S.Lw = take(D::kBig ? LM_STAGE(nx, nu) : 0);
        if constexpr (D::kBig)
            for (int i = lane; i < 2 * p.ring * S.lrows * S.nup; i += D::kNT) S.Ls[i] = 0.0; // (the padding stays zero: only entries are ever written)
        ldsi *qi = (ldsi *)q;
        auto takei = [&](int cnt) { ldsi *r = qi; qi += cnt; return r; };
        S.flag = takei(2);
        S.fix = takei(T * nub);
        S.ei = takei(ne); S.ej = takei(ne);
        ldsi *rptr0 = takei(NL ? 0 : p.mreg + 1), *rcol0 = takei(NL ? 0 : p.nnz0), *cptr0 = takei(NL ? 0 : nz + 1);
        ldsi *crow0 = takei(NL ? 0 : p.nnz0);
        ldsi *gptr0 = takei(D::kBig ? 0 : ne + 1), *grow0 = takei(D::kBig ? 0 : p.nng0);
        ldsi *drow0 = takei(split ? p.ndp : 0), *rinfo0 = takei(split ? p.mreg : 0), *sptr0 = takei(split ? nz + 1 : 0), *srow0 = takei(split ? p.ns : 0), *nrow0 = takei(split ? p.mreg - p.nd : 0);
        ldsb *cci0 = (ldsb *)qi; // nz * KC bytes (rounded up to a multiple of 4 in hmpc_lds_bytes)
        // stage the node-independent data
        const SparseStage &g0 = p.reg;
        if constexpr (D::kBig)
            for (int i = lane; i < abrows * S.abst; i += D::kNT) S.AB[i] = 0.0;
        __syncthreads();
        for (int i = lane; i < nx * nz; i += D::kNT) {
            const int l = i / nz, j = i - l * nz;
            S.AB[l * AB_STRIDE + j] = j < nx ? p.A[l * nx + j] : p.B[l * nu + (j - nx)];
        }
        for (int i = lane; i < nz * nz; i += D::kNT) S.P[i] = p.P[i];
        for (int i = lane; i < nx * nx; i += D::kNT) S.PT[i] = p.PT[i];
        if constexpr (!D::kBig)
            for (int i = lane; i < p.mreg; i += D::kNT) h0[i] = g0.h[i];
        if constexpr (D::kBig) {
            // nothing staged: the lists are walked in global memory (read-only, L2 resident)
        } else if constexpr (KC > 0) {
            for (int i = lane; i < nz * KC; i += D::kNT) {
                const int j = i / KC, qq = i - j * KC;
                cval0[i] = p.ccv[j * HMPC_KC_STRIDE + qq];
                cci0[i] = (unsigned char)p.cci[j * HMPC_KC_STRIDE + qq];
            }
        } else {
            for (int i = lane; i < p.nnz0; i += D::kNT) { rval0[i] = g0.rval[i]; cval0[i] = g0.cval[i]; rcol0[i] = g0.rcol[i]; crow0[i] = g0.crow[i]; }
            for (int i = lane; i < p.mreg + 1; i += D::kNT) rptr0[i] = g0.rptr[i];
            for (int i = lane; i < nz + 1; i += D::kNT) cptr0[i] = g0.cptr[i];
        }
        if constexpr (!D::kBig) {
            for (int i = lane; i < p.nng0; i += D::kNT) { gval0[i] = g0.gval[i]; grow0[i] = g0.grow[i]; }
            for (int i = lane; i < ne + 1; i += D::kNT) gptr0[i] = g0.gptr[i];
        }
        for (int i = lane; i < ne; i += D::kNT) { S.ei[i] = p.ei[i]; S.ej[i] = p.ej[i]; }
        S.L0 = ListsL{rptr0, rcol0, cptr0, crow0, gptr0, grow0, rval0, cval0, gval0, h0};
        S.G0 = ListsG{g0.rptr, g0.rcol, g0.cptr, g0.crow, g0.gptr, g0.grow, g0.rval, g0.cval, g0.gval, g0.h};
        S.ccv = cval0;
        S.Cdn = split ? cdn0 : nullptr;
        S.sval = sval0; S.drow = drow0; S.rinfo = rinfo0; S.sptr = sptr0; S.srow = srow0; S.nrow = nrow0;
        if (split) {
            for (int i = lane; i < p.ndp * nz; i += D::kNT) cdn0[i] = p.Cdn[i];
            for (int i = lane; i < p.mreg; i += D::kNT) { sval0[i] = p.sval[i]; rinfo0[i] = p.rinfo[i]; }
            for (int i = lane; i < p.ndp; i += D::kNT) drow0[i] = p.drow[i];
            for (int i = lane; i < nz + 1; i += D::kNT) sptr0[i] = p.sptr[i];
            for (int i = lane; i < p.ns; i += D::kNT) srow0[i] = p.srow[i];
            for (int i = lane; i < p.mreg - p.nd; i += D::kNT) nrow0[i] = p.nrow[i];
        }
        S.cci = cci0;
        S.term_on = 0;
        S.fullfix = 0;
    }
    Rows<RS> R;
    R.bind(rows_ws + (size_t)blockIdx.x * 4 * p.Mpad, p.Mpad);
    RM rm;
    rm.init(p, lane);
    rm.bind(p, S);
    // Nodes differ in work (7 to 25 interior-point iterations, a second solve when the terminal set binds): the
    // first gridDim.x nodes go to the workgroups by index, every further node to the first workgroup that is free
    // (one atomic per node on a counter the host zeroes before the launch).  A record does not depend on the
    // workgroup that computes it.
    int Bn = B; // nodes of this launch (a second pass -- two-launch form, second opinion -- reads their number from the device)
    if constexpr (WARM) { if (warm.second) Bn = warm.pend[0]; }
    // warm.second == 2: SECOND OPINION (hmpc_solve_batch_device): the nodes a kernel compiled at hmpc_create left undecided
    // (MAXITER / NUMERICAL), listed in warm.pend, are solved again by the shipped kernel -- this one, launched through its
    // hand-down instantiation so that the cold kernels do not carry the list mode; a node is handed what the first launch
    // handed it (warm.index), nothing else differs from a regular launch.
    // (guard: a workgroup never takes more nodes than the launch has -- a binary whose node counter came out of the compiler wrong
    // has been seen to spin here for good, profiles/r05_compiler_bisect.md; with the guard it ends, loudly wrong)
    for (int slot = blockIdx.x, guard = 0; slot < Bn && guard <= Bn; guard++) {
        // `order` (optional): the nodes sorted by the number of fixed binaries, shallow first -- shallow nodes take more
        // iterations (correlation -0.5 .. -0.75 on random frontiers), and with ~4 nodes per workgroup handing out the
        // long ones first shortens the tail of a launch
        int qp = order ? order[slot] : slot;
        if constexpr (WARM) { if (warm.second) qp = warm.pend[1 + slot]; }
        __syncthreads();
        for (int o = lane; o < T * nub; o += D::kNT) S.fix[o] = fixg[(size_t)qp * T * nub + o];
        for (int i = lane; i < nx; i += D::kNT) S.x0[i] = x0g[(size_t)qp * x0_stride + i];
        __syncthreads();
        {
            const int st = lane & 63;
            bool all = st < T && nub > 0;
            for (int bq = 0; bq < nub; bq++) all = all && S.fix[(st < T ? st : 0) * nub + bq] >= 0;
            S.fullfix = __ballot(all);
        }
#ifdef HMPC_CHECK
        {
#ifdef HMPC_POISON_FINITE // (bisecting: is it the VALUE that is read, or only the code around the stores that differs)
            const double poison = 1.0;
#else
            const double poison = __longlong_as_double(0x7ff8dead0000beefLL);
#endif
            // everything per node between the iterate and the staged constants (S.w .. S.mv); the node's inputs and the
            // staged problem data are not touched
            for (ldsd *q = S.w + lane; q < S.red; q += D::kNT) *q = poison;
            const int nslot = RS > 0 ? RS : p.Mpad / D::kNT;
#pragma unroll
            for (int k = 0; k < nslot; k++) {
#ifndef HMPC_POISON_MASK
#define HMPC_POISON_MASK 15 // (which of the four row arrays are poisoned: bisecting a read of something never written)
#endif
                if constexpr (RS > 0) {
                    if (HMPC_POISON_MASK & 1) R.s_[k] = poison;
                    if (HMPC_POISON_MASK & 2) R.zd_[k] = poison;
                    if (HMPC_POISON_MASK & 4) R.dz_[k] = poison;
                    if (HMPC_POISON_MASK & 8) R.prod_[k] = poison;
                }
            }
            __syncthreads();
        }
#endif
        int it1 = 0, it2 = 0, status = HMPC_MAXITER;
        bool polished = false, handed = false, second = false;
        int weak = 0;
        double tau = 1.0;
        double *tr = (trace && qp == 0) ? trace : nullptr;
        const double *wprim = nullptr, *wdual = nullptr; // the parent's record, if one is handed down
        if constexpr (WARM) {
            const int wrow = warm.second == 1 ? qp : warm.index ? warm.index[qp] : -1;
            wprim = wrow >= 0 ? warm.primal + (size_t)wrow * p.n_primal : nullptr;
            wdual = wrow >= 0 ? warm.dual + (size_t)wrow * p.n_dual : nullptr;
        }
        // Lazy terminal set: an infeasibility proof without the terminal-set rows is a proof for the
        // node and carries no terminal multipliers; an optimum that satisfies the masked rows strictly
        // is the node's optimum.  Otherwise solve again with every row.
        const int first = (p.nT > 0 && p.lazy) ? 0 : 1;
        if constexpr (!WARM) {
            for (int term_on = first; term_on < 2; term_on++) {
                int its = 0;
                S.term_on = term_on;
                status = ipm_solve<D, RS, RM, WARM>(p, S, R, rm, lane, term_on, its, tau, polished, weak, handed, tr ? tr + term_on * 64 * 8 : nullptr, wprim, wdual, false);
                if (term_on == 0) it1 = its; else it2 = its;
                if (term_on == 0) {
                    bool done = status == HMPC_INFEASIBLE;
                    if (status == HMPC_OPTIMAL) done = terminal_violation<D>(p, S, lane, tau) < 0.0;
                    if (done) break;
                    if (warm.pend) { // two-launch form: the second solve is the second launch's (DevWarm)
                        if (lane == 0) warm.pend[1 + atomicAdd(warm.pend, 1)] = qp;
                        break;
                    }
                }
            }
            // (HMPC_ITERS_TERMINAL is raised by the hand-down instantiations only: in this one the flag -- one more value
            // alive across the inlined solve -- cost 6 % of every launch by register allocation alone, 666 k -> 625 k QP/s measured)
        } else {
            // A parent whose optimum lies on terminal-set rows hands those down too: its set is first tried WITH the terminal
            // rows (with them masked the point is far from the parent's, the hand-down drops out at once and a full first
            // solve runs before the second one verifies).  Only an attempt (stage 2): if it does not verify the regular
            // sequence -- masked first, so that an infeasible node's ray carries no terminal multipliers -- runs as without
            // it.  Same steps as oracle/hsde_qp.c.
            int stage = first;
            if (warm.second == 1) { // second launch of the two-launch form: terminal rows on, from the node's own first record
                stage = 2;
                it1 = out.iters ? (out.iters[qp] & 0xFFFF) : 0;
            } else if (wdual != nullptr && p.polish && first == 0) {
                double m = 0.0;
                for (int k = lane; k < p.nT; k += D::kNT) m = fmax(m, wdual[(T + 1) * nx + (T - 1) * p.nc + p.nc + k]);
                if (block_max<D>(m, S.red, lane) > 0.0) stage = 2;
            }
            for (;;) { // (one call site: the solve is a large inlined body)
                const int term_on = stage == 0 ? 0 : 1;
                int its = 0;
                S.term_on = term_on;
                status = ipm_solve<D, RS, RM, WARM>(p, S, R, rm, lane, term_on, its, tau, polished, weak, handed, tr ? tr + term_on * 64 * 8 : nullptr, wprim, wdual, stage == 2,
                                                    stage == 2 && warm.second == 1);
                if (stage == 2) {
                    if (status == HMPC_RETRY) { stage = warm.second == 1 ? 1 : 0; continue; }
                    if (warm.second == 1) it2 = its; else it1 = its;
                    second = true;
                    break;
                }
                if (term_on == 0) it1 = its; else it2 = its;
                if (term_on == 1) break;
                bool done = status == HMPC_INFEASIBLE;
                if (status == HMPC_OPTIMAL) done = terminal_violation<D>(p, S, lane, tau) < 0.0;
                if (done) break;
                second = true;
                stage = 1;
            }
        }
#ifdef HMPC_TEST_UNDECIDED // (test hook, HMPC_JIT_FLAGS: a compiled kernel that leaves every k-th node undecided, for the nets of hmpc_solve_batch_device)
        if (qp % HMPC_TEST_UNDECIDED == 1) status = HMPC_NUMERICAL;
#endif
        __syncthreads();
        {
            const int nslot = RS > 0 ? RS : p.Mpad / D::kNT;
            ROWS_BEGIN(k, rw)
                S.e[rw.e] = R.z(k, rw.e);
            ROWS_END
        }
        __syncthreads();
        write_record<D>(p, S, lane, status, tau, qp, out);
#ifdef HMPC_CHECK
        if (status == HMPC_OPTIMAL) { // an optimal record holds no NaN
            for (int o = lane; o < n; o += D::kNT) HMPC_CHK(S.w[o] == S.w[o], 4);
            for (int o = lane; o < (T + 1) * nx; o += D::kNT) HMPC_CHK(S.lam[o] == S.lam[o], 5);
            for (int o = lane; o < M; o += D::kNT) HMPC_CHK(S.e[o] == S.e[o], 6);
        }
        HMPC_CHK(status >= HMPC_OPTIMAL && status <= HMPC_NUMERICAL && tau > 0.0, 7);
#endif
        if (lane == 0 && out.iters) out.iters[qp] = (it1 + it2) | (polished ? HMPC_ITERS_POLISHED : 0) | (weak ? HMPC_ITERS_WEAK : 0) | (weak == 2 ? HMPC_ITERS_UNCERTIFIED : 0) | (handed ? HMPC_ITERS_HANDED : 0) |
                                                     ((WARM && second) ? HMPC_ITERS_TERMINAL : 0);
        if constexpr (WARM) { // second opinion: how many of the listed nodes this kernel leaves undecided as well (two words in front of the list; the one between them is this launch's work counter)
            if (warm.second == 2 && lane == 0 && status >= HMPC_MAXITER) atomicAdd(warm.pend - 2, 1);
        }
        if (lane == 0) S.flag[1] = (int)gridDim.x + atomicAdd(p.work_counter, 1);
        __syncthreads();
        slot = S.flag[1];
    }
}

// SEPARATE COMPILATION (the shipped build, csrc/Makefile): every instantiation of hmpc_qp_kernel hmpc_pick_kernel can
// return is compiled in a file of its own (instances/*.hip: HMPC_KERNEL_ONLY + one HMPC_INSTANCE line) -- eight at a time
// instead of 26 in a row, 5 min -> 1 min --, and the launching translation unit (hmpc_capi.hip, HMPC_EXTERN_INSTANCES)
// only declares them.  Without HMPC_EXTERN_INSTANCES (the diagnostic builds) they are instantiated where they are used.
// HMPC_INSTANCE_LIST must name exactly the argument lists of hmpc_pick_kernel below.
#define HMPC_INSTANCE_LIST(X)                                                                              \
    X(4, 7, 4, 10, 3, 2, 1) X(4, 7, 4, 5, 2, 1, 2) X(4, 7, 4, 10, 3, 1, 2) X(4, 7, 4, 3, 1, 1, 4) X(4, 7, 4, 5, 2, 1, 4) \
    X(4, 4, 2, 7, 2, 1, 2) X(4, 4, 2, 4, 1, 1, 4)                                                          \
    X(-1, 0, 0, 0, 0, 0, 1) X(-1, 0, 0, 0, 0, 0, 2) X(-1, 0, 0, 0, 0, 0, 4) X(0, 0, 0, 0, 0, 0, 1) X(0, 0, 0, 0, 0, 0, 2) X(0, 0, 0, 0, 0, 0, 4)
#define HMPC_KERNEL_SIG(NX, NU, NUB, F, Bn, Tn, NWv, W) \
    void hmpc_qp_kernel<NX, NU, NUB, F, Bn, Tn, NWv, W>(const DevProb, const double *, int, const int8_t *, int, const DevOut, double *, double *, const int32_t *, const DevWarm)
#define HMPC_INSTANCE(NX, NU, NUB, F, Bn, Tn, NWv) /* one line of an instance file: the cold and the hand-down kernel */ \
    template __global__ HMPC_KERNEL_SIG(NX, NU, NUB, F, Bn, Tn, NWv, false);                              \
    template __global__ HMPC_KERNEL_SIG(NX, NU, NUB, F, Bn, Tn, NWv, true);
#ifndef HMPC_KERNEL_ONLY // (the instance files and the code-generation probes compile hmpc_qp_kernel only)
// Processing order of a large frontier: counting sort of the nodes by their number of fixed binaries (one workgroup;
// the order inside a bucket is whatever the atomics give -- a record does not depend on when its node is solved).
// Launches with hand-down (warm.index set): a node whose PARENT needed the terminal-set rows goes first -- its own solve is
// the long kind (up to two full solves when the handed-down set does not verify, ~5.6 ms against ~2 ms), and a long node
// that starts late is the tail of the launch (measured on real trees: 11.8 ms per 4096 nodes with such nodes scattered).
// term_off: offset of the terminal-set multipliers in a dual row; nT of them.
__global__ void __launch_bounds__(1024) hmpc_order_kernel(const int8_t *__restrict__ fixg, int B, int nfix, int32_t *__restrict__ order,
                                                          const DevWarm warm, int term_off, int nT, int n_dual)
{
    __shared__ int bins[1025];
    const int nb = nfix + 1 < 1024 ? nfix + 1 : 1024;
    for (int i = threadIdx.x; i <= nb; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    const bool words = (nfix & 3) == 0 && ((size_t)fixg & 3) == 0;
    auto bucket = [&](int b) {
        if (warm.index != nullptr && nT > 0) {
            const int r = warm.index[b];
            if (r >= 0) {
                const double *mu = warm.dual + (size_t)r * n_dual + term_off;
                bool any = false;
                for (int k = 0; k < nT; k++) any = any || mu[k] > 0.0;
                if (any) return 0; // (shares the bucket of the root-like nodes: first out)
            }
        }
        const int8_t *f = fixg + (size_t)b * nfix;
        int d = 0;
        if (words) { // four entries per load: an entry is fixed iff its sign bit is clear
            const unsigned *w = (const unsigned *)f;
            for (int i = 0; i < nfix / 4; i++) d += __popc(~w[i] & 0x80808080u);
        } else {
            for (int i = 0; i < nfix; i++) d += f[i] >= 0;
        }
        return nfix + 1 <= 1024 ? d : (int)((long long)d * 1023 / nfix);
    };
    int mine[8]; // buckets of this thread's first 8 nodes (8192 nodes per launch), kept between the two passes
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int b = threadIdx.x + q * 1024;
        mine[q] = b < B ? bucket(b) : -1;
        if (mine[q] >= 0) atomicAdd(&bins[mine[q] + 1], 1);
    }
    for (int b = threadIdx.x + 8 * 1024; b < B; b += 1024) atomicAdd(&bins[bucket(b) + 1], 1);
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive scan by one wave: bins[k] = first slot of bucket k
        int carry = 0;
        for (int base = 0; base < nb; base += 64) {
            const int i = base + threadIdx.x + 1;
            int v = i <= nb ? bins[i] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(v, o);
                if ((int)threadIdx.x >= o) v += u;
            }
            if (i <= nb) bins[i] = v + carry;
            carry += __shfl(v, 63);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (mine[q] >= 0) order[atomicAdd(&bins[mine[q]], 1)] = threadIdx.x + q * 1024;
    for (int b = threadIdx.x + 8 * 1024; b < B; b += 1024) order[atomicAdd(&bins[bucket(b)], 1)] = b;
}

// Nodes of a launch that ended MAXITER / NUMERICAL, listed for the second opinion (hmpc_solve_batch_device): list[0] how many,
// list[1 ..] which (any order: a record does not depend on the workgroup that computes it).
__global__ void __launch_bounds__(256) hmpc_hard_kernel(const int32_t *__restrict__ status, int B, int32_t *__restrict__ list)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B && status[b] >= HMPC_MAXITER) list[1 + atomicAdd(list, 1)] = b;
}

// The nodes of the first-use check of a compiled kernel (hmpc_check_compiled): N - 2 nodes spread over the batch, the root
// relaxation of the batch's first initial state (every binary free) and its deepest node (every binary fixed to zero).
__global__ void __launch_bounds__(256) hmpc_check_set_kernel(const double *__restrict__ x0g, int x0_stride, const int8_t *__restrict__ fixg, int B, int nfix, int nx, int N,
                                                             double *__restrict__ x0c, int8_t *__restrict__ fixc)
{
    for (int k = blockIdx.x; k < N; k += gridDim.x) {
        const int src = k < N - 2 ? (int)((long long)k * B / (N - 2)) : 0;
        for (int i = threadIdx.x; i < nfix; i += 256) fixc[(size_t)k * nfix + i] = k == N - 2 ? (int8_t)-1 : k == N - 1 ? (int8_t)0 : fixg[(size_t)src * nfix + i];
        for (int i = threadIdx.x; i < nx; i += 256) x0c[(size_t)k * nx + i] = x0g[(size_t)src * x0_stride + i];
    }
}

#ifdef HMPC_EXTERN_INSTANCES
#define HMPC_EXTERN_INSTANCE(NX, NU, NUB, F, Bn, Tn, NWv)                                                  \
    extern template __global__ HMPC_KERNEL_SIG(NX, NU, NUB, F, Bn, Tn, NWv, false);                       \
    extern template __global__ HMPC_KERNEL_SIG(NX, NU, NUB, F, Bn, Tn, NWv, true);
HMPC_INSTANCE_LIST(HMPC_EXTERN_INSTANCE)
#endif
// Instantiations: the two cart-pole shapes of the reference (notebooks/cart_pole_with_walls: nx=4,
// nu=7, 4 binaries; warm_start_hmpc/test/cart_pole_with_wall.py: nx=4, nu=4, 2 binaries), with the
// row slots of their horizons and 1 / 2 / 4 waves per node, and the generic run-time-sized kernel.
typedef void (*hmpc_kernel_t)(const DevProb, const double *, int, const int8_t *, int, const DevOut, double *, double *, const int32_t *, const DevWarm);
struct hmpc_kernel_choice {
    hmpc_kernel_t fn;
    hmpc_kernel_t fn_warm; // the same kernel with the parent -> child hand-down compiled in (launches with hmpc_warm)
    int waves;
    int kc;  // entries per padded column of the kernel's LDS carve (Dims::kKC), 0: generic kernel
    int big; // generic kernel with lists and factor in global memory (Dims::kBig)
};
// Waves per node, measured on MI355X (cart-pole N=20, ms per batch with 1 / 2 / 4 waves):
//   8 nodes 1.81 / 1.50 / 1.41    77: 2.78 / 2.26 / 2.09    256: 2.88 / 2.47 / 2.30
//   512: 2.95 / 2.67 / 3.76       1024: 3.69 / 6.81 / 6.92  2048: 7.46 / 11.1 / 12.5
// A CU holds 4 waves of this kernel (one per SIMD, 512 registers each): a frontier that fills every
// SIMD with one node per wave is fastest with one wave per node; smaller frontiers finish sooner with
// the row-parallel phases spread over the otherwise idle SIMDs.  HMPC_WAVES overrides.
static int hmpc_waves_for(int B, int resident_nodes)
{
    if (const char *e = getenv("HMPC_WAVES")) {
        const int nw = atoi(e);
        if (nw == 1 || nw == 2 || nw == 4) return nw;
    }
    if (B <= resident_nodes / 4) return 4;
    if (B <= resident_nodes * 3 / 4) return 2; // (round 2, DPP broadcasts: 768 nodes 3.19 ms with 2 waves, 3.37 with 1)
    return 1;
}
// Slots the static row map needs for this problem with nw waves per node (see RowMapS).
static bool hmpc_static_slots(const DevProb &p, int nw, int &kf, int &kb, int &kt)
{
    const int nt = nw * WAVE;
    if (!p.static_rows || p.nc < 1 || p.nc > nt || p.nub < 1 || 2 * p.nub > nt) return false;
    const int sp = nt / p.nc, sb = nt / (2 * p.nub);
    kf = (p.T + sp - 1) / sp;
    kb = (p.T + sb - 1) / sb;
    kt = (p.nT + nt - 1) / nt;
    return true;
}
#define HMPC_TRY(NX, NU, NUB, F, Bn, Tn, NWv) \
    if (kf <= F && kb <= Bn && kt <= Tn && p.kcol <= Dims<NX, NU, NUB, NWv>::kKC)       \
        return {hmpc_qp_kernel<NX, NU, NUB, F, Bn, Tn, NWv>, hmpc_qp_kernel<NX, NU, NUB, F, Bn, Tn, NWv, true>, NWv, Dims<NX, NU, NUB, NWv>::kKC, 0};
// jit (optional): kernels compiled for this problem's shape at hmpc_create (hmpc_jit.h), by log2 of the waves per node;
// fn null where there is none.
static hmpc_kernel_choice hmpc_pick_kernel(const DevProb &p, int nw, const hmpc_kernel_choice *jit = nullptr)
{
#ifdef HMPC_DEV_BIG_ONLY
    // development build (`make dev`): only the streaming form of the generic kernel is instantiated (compiles in a
    // fraction of the time); never shipped
    (void)p; (void)nw;
    return {hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 4>, hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 4, true>, 4, 0, 1};
#else
    const bool generic = getenv("HMPC_FORCE_GENERIC") != nullptr || getenv("HMPC_FORCE_BIG") != nullptr;
    int kf = 0, kb = 0, kt = 0;
    if (!generic && p.ngram <= WAVE && p.nx == 4 && p.nu == 7 && p.nub == 4) {
        // smallest instantiation that holds the rows; fewer waves than asked for never fit more rows
        for (int w = nw; w <= 4; w *= 2) {
            if (!hmpc_static_slots(p, w, kf, kb, kt)) continue;
            if (w == 1) { HMPC_TRY(4, 7, 4, 10, 3, 2, 1) }
            if (w == 2) { HMPC_TRY(4, 7, 4, 5, 2, 1, 2) HMPC_TRY(4, 7, 4, 10, 3, 1, 2) }
            if (w == 4) { HMPC_TRY(4, 7, 4, 3, 1, 1, 4) HMPC_TRY(4, 7, 4, 5, 2, 1, 4) }
        }
    }
    if (!generic && p.ngram <= WAVE && p.nx == 4 && p.nu == 4 && p.nub == 2) {
        for (int w = nw < 2 ? 2 : nw; w <= 4; w *= 2) {
            if (!hmpc_static_slots(p, w, kf, kb, kt)) continue;
            if (w == 2) { HMPC_TRY(4, 4, 2, 7, 2, 1, 2) }
            if (w == 4) { HMPC_TRY(4, 4, 2, 4, 1, 1, 4) }
        }
    }
    if (!generic && jit) {
        for (int w = nw; w <= 4; w *= 2) {
            const hmpc_kernel_choice &j = jit[w == 1 ? 0 : w == 2 ? 1 : 2];
            if (j.fn) return j;
        }
    }
    // generic kernel; its streaming form when lists and factor do not fit one CU's LDS
    if (hmpc_lds_bytes(p, 0, 0) > 160 * 1024 || getenv("HMPC_FORCE_BIG")) {
        if (nw == 1) return {hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 1>, hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 1, true>, 1, 0, 1};
        if (nw == 2) return {hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 2>, hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 2, true>, 2, 0, 1};
        return {hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 4>, hmpc_qp_kernel<-1, 0, 0, 0, 0, 0, 4, true>, 4, 0, 1};
    }
    if (nw == 1) return {hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 1>, hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 1, true>, 1, 0, 0};
    if (nw == 2) return {hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 2>, hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 2, true>, 2, 0, 0};
    return {hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 4>, hmpc_qp_kernel<0, 0, 0, 0, 0, 0, 4, true>, 4, 0, 0};
#endif
}
#undef HMPC_TRY
#endif
