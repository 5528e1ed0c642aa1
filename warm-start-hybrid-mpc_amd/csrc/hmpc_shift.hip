// hmpc_shift.hip -- warm-start node shift of branch-and-bound leaves on the device.
//
// Restates, for flat dual rows (layout: include/hmpc.h), what the reference does per leaf with
// Python lists (warm_start_hmpc/controller.py:431-564 construct_warm_start, :566-613 retain rule,
// :615-666 _shift_dual_variables, :668-721 _pi_sum); the host mirror of the same arithmetic is
// warm_start_hmpc_amd/batched.py::construct_warm_start and controller.py::_pi_sum, against which
// tests/test_gpu_parity.py checks this kernel.
//
// One wavefront per leaf, SHIFT_WAVES per workgroup; the workgroup stages the node-independent maps in LDS once and then
// walks its share of the leaves.  The work per leaf is a strided copy of the row (everything moves one stage
// towards the present), two small matrix-vector products for the stage that enters at the end of
// the horizon, and a handful of dot products for the change of the dual objective: ~8 KB read and
// ~8 KB written per leaf, no reuse -- HBM bound.
//
// Two kernels: `hmpc_shift_row_kernel` (second half of this file; the one that runs wherever LDS has room for the rows of four
// waves: the row staged in LDS by `global_load_lds_dwordx4`, 4.5 TB/s at 65 536 leaves) and `hmpc_shift_kernel` (rows through
// registers, 3.8 TB/s; longer rows, HMPC_SHIFT_ROWS=0).  Same results to rounding; tests/test_gpu_parity.py runs both.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hmpc_device.h"

struct ShiftArgs {
    int B, K;
    const int32_t *owner;
    const double *x0, *u0, *e0;
    const int8_t *fix;
    const double *lb, *dual, *dobj;
    const int32_t *src; // row of dual / dobj that leaf b carries (children share their parent's row); null: row b
    int8_t *fix_out;
    double *lb_out, *dual_out, *dobj_out;
    uint8_t *flags;
};

static __device__ __forceinline__ double shift_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Waves per workgroup: the workgroup stages the maps once and every wave shifts one leaf at a time.
// The kernel is bound by the latency of a wave's own chain of memory round trips, so its rate is waves in flight divided
// by round trips per leaf.  Measured on MI355X (65 536 leaves):
//   8 waves per workgroup at the compiler's 163 registers -> one workgroup per CU                      2.12 TB/s
//   4 waves per workgroup held to 128 registers (4 waves per SIMD, 4 workgroups per CU), 36 spilled    2.46-2.55 TB/s
//   the same without spills (8 instead of 16 loads in flight per lane)                                 2.97 TB/s
//   no load or store under a lane predicate in the copy (see there), heads batched and prefetched      3.78 TB/s
//   (262 144 leaves: 4.44 TB/s = 55 % of the HBM peak); 3 waves per SIMD: 2.68, 5 and more spill: < 1.9.
#ifndef SHIFT_WAVES
#define SHIFT_WAVES 4
#endif
// loads in flight per lane in a copy batch (16 needs more than the 128 registers of 4 waves per SIMD: spills)
#ifndef SHIFT_CU
#define SHIFT_CU 8
#endif
// register slots (x 64 lanes) for the head of a row: old last stage of mu, identifier
#ifndef SHIFT_HEAD_SLOTS
#define SHIFT_HEAD_SLOTS 3
#endif
#ifndef HMPC_SHIFT_ATTR
#define HMPC_SHIFT_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

// LDS doubles the kernel needs (the host checks this against the CU's LDS; larger problems read the maps in place).
static inline size_t hmpc_shift_lds_doubles(const DevProb &p, bool staged)
{
    size_t d = (size_t)SHIFT_WAVES * (p.ncL + p.nqT + p.nx + p.nu);       // per wave: old last stage (mu_{T-1}, rho_T), x0, u0
    d += (size_t)p.nq * p.nx + (size_t)p.nr * p.nu + (size_t)p.nub * p.nu + (size_t)p.nq * p.nqT; // Q R V M_rho
    if (staged) d += (size_t)p.ncL * p.nc + (size_t)p.nc * (p.nx + p.nu + 1) + p.ncL; // M_mu' ; [F G | h] ; h_Tm1
    return d;
}

template <bool STAGED>
__global__ void __launch_bounds__(64 * SHIFT_WAVES) HMPC_SHIFT_ATTR hmpc_shift_kernel(const DevProb p, const ShiftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = p.nx, nu = p.nu, nub = p.nub, nuc = p.nuc, T = p.T, nc = p.nc, ncL = p.ncL, nq = p.nq, nr = p.nr, nqT = p.nqT;
    const int o_mu = (T + 1) * nx, o_lb = o_mu + (T - 1) * nc + ncL, o_ub = o_lb + T * nub, o_rho = o_ub + T * nub;
    const int o_sig = o_rho + T * nq + nqT;
    // LDS: per wave the old last stage; then (if they fit) the maps, M_mu transposed so that lanes read rows
    const int pw = ncL + nqT + nx + nu;
    double *muL = sm + (size_t)wave * pw, *rhoT = muL + ncL, *xs = rhoT + nqT, *us = xs + nx;
    double *Qs = sm + (size_t)SHIFT_WAVES * pw, *Rs = Qs + nq * nx, *Vs = Rs + nr * nu, *Mr = Vs + nub * nu;
    double *MT = Mr + nq * nqT, *FG = MT + (size_t)ncL * nc, *hT = FG + (size_t)nc * (nx + nu + 1);
    for (int i = tid; i < nq * nx; i += 64 * SHIFT_WAVES) Qs[i] = p.Q[i];
    for (int i = tid; i < nr * nu; i += 64 * SHIFT_WAVES) Rs[i] = p.R[i];
    for (int i = tid; i < nub * nu; i += 64 * SHIFT_WAVES) Vs[i] = p.shift_V[i];
    for (int i = tid; i < nq * nqT; i += 64 * SHIFT_WAVES) Mr[i] = p.shift_Mrho[i];
    if (STAGED) {
        for (int i = tid; i < ncL * nc; i += 64 * SHIFT_WAVES) {
            const int k = i / nc, r = i - k * nc;
            MT[i] = p.shift_Mmu[(size_t)r * ncL + k];
        }
        for (int i = tid; i < nc * (nx + nu + 1); i += 64 * SHIFT_WAVES) {
            const int r = i / (nx + nu + 1), j = i - r * (nx + nu + 1);
            FG[i] = j < nx ? p.F_raw[(size_t)r * nx + j] : j < nx + nu ? p.G_raw[(size_t)r * nu + (j - nx)] : p.h_raw[r];
        }
        for (int i = tid; i < ncL; i += 64 * SHIFT_WAVES) hT[i] = p.hT_raw[i];
    }
    __syncthreads();
    const int half = lane >= 32 ? 1 : 0, hl = lane & 31; // two half-waves split the long sums
    // the head of the NEXT leaf (owner, the binaries it fixes at time 0) is fetched while the current one is shifted
    const int stride = gridDim.x * SHIFT_WAVES;
    int b = blockIdx.x * SHIFT_WAVES + wave;
    int own_n = b < a.B ? a.owner[b] : 0;
    int fix_n = (b < a.B && lane < nub) ? a.fix[(size_t)b * T * nub + lane] : -1;
    for (; b < a.B; b += stride) {
        const int own = own_n, fix_lane = fix_n;
        if (b + stride < a.B) {
            own_n = a.owner[b + stride];
            fix_n = lane < nub ? a.fix[(size_t)(b + stride) * T * nub + lane] : -1;
        }
        const double *x0g = a.x0 + (size_t)own * nx, *u0g = a.u0 + (size_t)own * nu, *e0 = a.e0 + (size_t)own * nx;
        const int8_t *fix = a.fix + (size_t)b * T * nub;
        const size_t srow = a.src ? (size_t)a.src[b] : (size_t)b;
        const double *d = a.dual + srow * p.n_dual;
        double *o = a.dual_out + (size_t)b * p.n_dual;
        // retain rule (controller.py:566-613): the binaries the leaf fixes at time 0 are the applied ones
        int agree = 1;
        if (lane < nub) agree = fix_lane < 0 || fix_lane == (int)rint(u0g[nuc + lane]);
        agree = __all(agree);
        if (!agree) { // dropped: nothing else is defined for this leaf
            if (lane == 0) a.flags[b] = 0;
            continue;
        }
        // The head of the row: the old last stage goes to LDS (read by every lane below), the identifier moves one stage.
        // Same rule as for the copy below: a loop "load, use" with a run-time trip count waits for the memory once per
        // trip, and a load or store under a lane predicate makes every later one wait for everything before it -- so
        // batches of SHIFT_HEAD_SLOTS x 64 entries from clamped indices (lanes beyond the end repeat the last entry).
        {
            const double *dL = d + o_mu + (T - 1) * nc;
            // (the first 64 entries of rho_T, x0, u0 travel with the first batch)
            // (nqT == 0 -- no terminal cost -- is legal: the clamp would then address entry -1, the last multiplier of the old
            // last stage; the condition is wave uniform, so it adds no lane predicate)
            const double rv = nqT > 0 ? d[o_rho + T * nq + min(lane, nqT - 1)] : 0.0, xv = x0g[min(lane, nx - 1)], uv = u0g[min(lane, nu - 1)];
            for (int k0 = 0; k0 < ncL; k0 += 64 * SHIFT_HEAD_SLOTS) {
                double mh[SHIFT_HEAD_SLOTS];
#pragma unroll
                for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) mh[u] = dL[min(k0 + u * 64 + lane, ncL - 1)];
#pragma unroll
                for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) muL[min(k0 + u * 64 + lane, ncL - 1)] = mh[u];
            }
            if (nqT > 0) rhoT[min(lane, nqT - 1)] = rv;
            xs[min(lane, nx - 1)] = xv; us[min(lane, nu - 1)] = uv;
        }
        for (int k = lane + 64; k < nqT; k += 64) rhoT[k] = d[o_rho + T * nq + k];
        for (int j = lane + 64; j < nx; j += 64) xs[j] = x0g[j];
        for (int j = lane + 64; j < nu; j += 64) us[j] = u0g[j];
        const double *x0 = xs, *u0 = us; // LDS copies: every dot product below reads them
        // the time-0 entries the pi-sum needs are not part of the copy below: fetch them now, so that after the
        // copy nothing waits on global memory any more
        const double mu0_first = (half == 0 && hl < nc) ? d[o_mu + hl] : 0.0;                  // rows of the first chunk
        const double rho0 = lane < nq ? d[o_rho + lane] : 0.0, sig0 = lane < nr ? d[o_sig + lane] : 0.0;
        const double nulb0 = lane < nub ? d[o_lb + lane] : 0.0, nuub0 = lane < nub ? d[o_ub + lane] : 0.0;
        const double lam1 = lane < nx ? d[nx + lane] : 0.0, e0v = lane < nx ? e0[lane] : 0.0;
        const int fix0 = fix_lane;
        const double dobj_in = a.dobj[srow], lb_in = a.lb[b];
        // identifier: drop time 0, the stage that enters is free
        {
            int8_t *fo = a.fix_out + (size_t)b * T * nub;
            const int nfix = T * nub;
            for (int i0 = 0; i0 < nfix; i0 += 64 * SHIFT_HEAD_SLOTS) {
                int fb[SHIFT_HEAD_SLOTS];
#pragma unroll
                for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) fb[u] = fix[min(i0 + u * 64 + lane + nub, nfix - 1)];
#pragma unroll
                for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) {
                    const int i = min(i0 + u * 64 + lane, nfix - 1);
                    fo[i] = i < (T - 1) * nub ? (int8_t)fb[u] : (int8_t)-1;
                }
            }
        }
        // multipliers: everything moves one stage towards the present, the end is padded with zeros
        // (controller.py:615-666).  One pass over the output row, SHIFT_CU independent loads per lane in flight;
        // the two blocks that come from the maps (mu'_{T-2}, rho'_{T-1}) are written further down.
        const int e_mu1 = o_mu + (T - 2) * nc, e_mu2 = o_mu + (T - 1) * nc, e_rho1 = o_rho + (T - 1) * nq, e_rho2 = o_rho + T * nq;
        // A store under a lane predicate sits in a branch of its own; the wait counter then cannot tell how many memory
        // operations are outstanding and every one of the stores waits for ALL earlier ones, loads and stores, to
        // complete: a batch of eight stores paid eight write acknowledgements one after the other.  So the copy is free of
        // predicates: loads from a clamped index with the value selected afterwards, unconditional stores -- the two
        // mapped blocks receive zeros first and their values after the release fence below.
        constexpr int CU_ = SHIFT_CU;
        const int i_last = p.n_dual - 1;
        for (int i0 = 0; i0 < p.n_dual; i0 += 64 * CU_) {
            int src[CU_], dst[CU_];
            double v[CU_];
#pragma unroll
            for (int u = 0; u < CU_; u++) {
                // lanes beyond the end of the row repeat its last entry (same value to the same address: no predicate)
                const int i = min(i0 + u * 64 + lane, i_last);
                int sft, lim; // source = i + sft while i < lim, zero from lim to the end of the segment
                if (i < o_mu) { sft = nx; lim = T * nx; }
                else if (i < o_lb) { sft = nc; lim = e_mu1; }
                else if (i < o_ub) { sft = nub; lim = o_lb + (T - 1) * nub; }
                else if (i < o_rho) { sft = nub; lim = o_ub + (T - 1) * nub; }
                else if (i < o_sig) { sft = nq; lim = e_rho1; }
                else { sft = nr; lim = o_sig + (T - 1) * nr; }
                const bool mapped = (i >= e_mu1 && i < e_mu2) || (i >= e_rho1 && i < e_rho2);
                src[u] = (!mapped && i < lim) ? i + sft : -1;
                dst[u] = i;
            }
#pragma unroll
            for (int u = 0; u < CU_; u++) { const double t = d[max(src[u], 0)]; v[u] = src[u] >= 0 ? t : 0.0; }
#pragma unroll
            for (int u = 0; u < CU_; u++) o[dst[u]] = v[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the zeros of the mapped blocks are out before their values
        __builtin_amdgcn_wave_barrier(); // muL / rhoT written by this wave are read by all its lanes below
        double acc = 0.0; // every lane's share of the pi-sum (controller.py:668-721)
        // mu'_{T-2} = M_mu mu_{T-1}: rows in chunks of 32, each half-wave sums half of the columns
        const int kh = (ncL + 1) / 2, k0 = half * kh, k1 = (k0 + kh < ncL) ? k0 + kh : ncL;
        for (int r0 = 0; r0 < nc; r0 += 32) {
            const int r = r0 + hl;
            double m = 0.0;
            if (r < nc) {
                if (STAGED) {
                    for (int k = k0; k < k1; k++) m += MT[(size_t)k * nc + r] * muL[k];
                } else {
                    for (int k = k0; k < k1; k++) m += p.shift_Mmu[(size_t)r * ncL + k] * muL[k];
                }
            }
            m += __shfl_xor(m, 32);
            if (r < nc && half == 0) {
                o[o_mu + (T - 2) * nc + r] = m;
                // - mu'_{T-2} . h  - mu_0 . (F x0 + G u0 - h)
                double g;
                if (STAGED) {
                    const double *row = FG + (size_t)r * (nx + nu + 1);
                    g = -row[nx + nu];
                    acc += m * g;
                    for (int j = 0; j < nx; j++) g += row[j] * x0[j];
                    for (int j = 0; j < nu; j++) g += row[nx + j] * u0[j];
                } else {
                    g = -p.h_raw[r];
                    acc += m * g;
                    for (int j = 0; j < nx; j++) g += p.F_raw[(size_t)r * nx + j] * x0[j];
                    for (int j = 0; j < nu; j++) g += p.G_raw[(size_t)r * nu + j] * u0[j];
                }
                acc -= (r0 == 0 ? mu0_first : d[o_mu + r]) * g;
            }
        }
        for (int k = lane; k < ncL; k += 64) acc += muL[k] * (STAGED ? hT[k] : p.hT_raw[k]); // + mu_{T-1} . h_Tm1
        for (int r = lane; r < nq; r += 64) {
            double m = 0.0;
            for (int k = 0; k < nqT; k++) m += Mr[r * nqT + k] * rhoT[k];
            o[o_rho + (T - 1) * nq + r] = m;
            acc -= 0.25 * m * m; // - |rho'_{T-1}|^2 / 4
            double qx = 0.0;
            for (int j = 0; j < nx; j++) qx += Qs[r * nx + j] * x0[j];
            const double t0 = 0.5 * (r < 64 ? rho0 : d[o_rho + r]) - qx;
            acc += t0 * t0 - qx * qx; // |rho_0 / 2 - Q x0|^2 - |Q x0|^2
        }
        for (int k = lane; k < nqT; k += 64) acc += 0.25 * rhoT[k] * rhoT[k]; // + |rho_T|^2 / 4
        for (int r = lane; r < nr; r += 64) {
            double ru = 0.0;
            for (int j = 0; j < nu; j++) ru += Rs[r * nu + j] * u0[j];
            const double t0 = 0.5 * (r < 64 ? sig0 : d[o_sig + r]) - ru;
            acc += t0 * t0 - ru * ru;
        }
        for (int i = lane; i < nub; i += 64) {
            double vu = 0.0;
            for (int j = 0; j < nu; j++) vu += Vs[i * nu + j] * u0[j];
            const int f = i < 64 ? fix0 : fix[i];
            const double lo = f >= 0 ? (double)f : 0.0, hi = f >= 0 ? (double)f : 1.0;
            acc -= (lo - vu) * (i < 64 ? nulb0 : d[o_lb + i]) + (vu - hi) * (i < 64 ? nuub0 : d[o_ub + i]);
        }
        // model error against the new lam_0 (= old lam_1), controller.py:541-558
        acc -= lam1 * e0v;
        for (int j = lane + 64; j < nx; j += 64) acc -= d[nx + j] * e0[j];
        const double pi = shift_wave_sum(acc);
        if (lane == 0) {
            double obj = dobj_in + pi;
            obj = obj > 0.0 ? obj : 0.0;
            const double lb = lb_in;
            uint8_t flag = 1;
            double nlb;
            if (!isinf(lb)) nlb = obj;                      // a solved / bounded leaf: its bound is the shifted dual objective
            else if (obj <= 0.0) { nlb = 0.0; flag |= 2; } // the infeasibility proof did not survive: reopen
            else nlb = lb;                                  // still proved infeasible
            a.lb_out[b] = nlb;
            a.dobj_out[b] = obj;
            a.flags[b] = flag;
        }
        __builtin_amdgcn_wave_barrier(); // the next leaf overwrites muL / rhoT
    }
}

// ---- The same shift with the leaf's row staged in LDS by the memory pipeline itself (round 5) ------------------------------------
// The kernel above keeps a row's entries in registers between its loads and its stores, so the bytes a wave has in flight are
// bounded by its registers (8 loads of 512 B: half a row of the cart-pole problem), a leaf costs five to six dependent round
// trips (head, two copy batches, the release fence, the mapped blocks) and every leaf repeats the products with x0 and u0 that
// all leaves of a tree share.  Measured with phases knocked out (profiles/r05_shift_phases.txt): a leaf costs a wave 7 us, of
// which the memory round trip is 0.2 -- the instruction stream of the wave is the bound, not the memory.  So:
//  * `hmpc_shift_tree_kernel` computes once per TREE what depends on (x0, u0) only: F x0 + G u0 - h, Q x0, R u0, V u0;
//  * a wave fetches the WHOLE row with `global_load_lds_dwordx4` -- 1 KiB per instruction straight into LDS, no register holds
//    anything -- waits ONCE, and then everything the shift needs is an LDS read: the old last stage in place, the time-0
//    entries of the pi-sum, and the copy itself, which becomes `out[i] = buffer[map[i]]` with ONE map per workgroup (source
//    entry, the zero slot, or the slot of a mapped block -- computed before the copy, so every output entry is stored exactly
//    once and nothing needs a fence);
//  * the mapped block M_mu mu_{T-1} reads its matrix in pairs of columns (ds_read_b128), eight columns per batch, two chains.
// One workgroup per CU, as many waves as LDS has room for rows (cart-pole N = 20: 14 waves x 8 KiB in flight per CU); problems
// whose rows leave room for fewer than four waves take the kernel above.
static inline size_t hmpc_shift_row_pieces(const DevProb &p) { return ((size_t)p.n_dual + 127) / 128; }  // 1 KiB pieces of a row
// doubles per wave: the row (whole pieces), a zero, the two mapped blocks (even count: rows stay 16-byte aligned)
static inline size_t hmpc_shift_row_wave_doubles(const DevProb &p)
{
    const size_t d = hmpc_shift_row_pieces(p) * 128 + 1 + p.nc + p.nq;
    return (d + 1) / 2 * 2;
}
// doubles per workgroup: M_mu in pairs of columns, h, M_rho, h_Tm1, then the copy map (n_dual ints)
static inline size_t hmpc_shift_row_fixed_doubles(const DevProb &p)
{
    size_t d = ((size_t)p.ncL + 1) / 2 * 2 * p.nc + p.nc + (size_t)p.nq * p.nqT + p.ncL;
    d = (d + 1) / 2 * 2;
    return d + ((size_t)p.n_dual + 3) / 4 * 2;
}
// doubles per tree of the tree pass: F x0 + G u0 - h | Q x0 | R u0 | V u0
static inline size_t hmpc_shift_tree_doubles(const DevProb &p) { return (size_t)p.nc + p.nq + p.nr + p.nub; }

// One wave per tree; same order of operations as the per-leaf loops of the kernel above (and of batched.py).
__global__ void __launch_bounds__(64) hmpc_shift_tree_kernel(const DevProb p, int K, const double *x0s, const double *u0s, double *tv)
{
    const int k = blockIdx.x, lane = threadIdx.x, nx = p.nx, nu = p.nu, nc = p.nc, nq = p.nq, nr = p.nr, nub = p.nub;
    if (k >= K) return;
    const double *x0 = x0s + (size_t)k * nx, *u0 = u0s + (size_t)k * nu;
    double *o = tv + (size_t)k * (nc + nq + nr + nub);
    for (int r = lane; r < nc; r += 64) {
        double g = -p.h_raw[r];
        for (int j = 0; j < nx; j++) g += p.F_raw[(size_t)r * nx + j] * x0[j];
        for (int j = 0; j < nu; j++) g += p.G_raw[(size_t)r * nu + j] * u0[j];
        o[r] = g;
    }
    for (int r = lane; r < nq; r += 64) {
        double qx = 0.0;
        for (int j = 0; j < nx; j++) qx += p.Q[r * nx + j] * x0[j];
        o[nc + r] = qx;
    }
    for (int r = lane; r < nr; r += 64) {
        double ru = 0.0;
        for (int j = 0; j < nu; j++) ru += p.R[r * nu + j] * u0[j];
        o[nc + nq + r] = ru;
    }
    for (int i = lane; i < nub; i += 64) {
        double vu = 0.0;
        for (int j = 0; j < nu; j++) vu += p.shift_V[i * nu + j] * u0[j];
        o[nc + nq + nr + i] = vu;
    }
}

#ifndef SHIFT_KO
#define SHIFT_KO 0 // (diagnostic: phases knocked out for timing -- 1 row fetch, 2 mapped block, 4 copy)
#endif
__global__ void __launch_bounds__(1024) hmpc_shift_row_kernel(const DevProb p, const ShiftArgs a, const double *tv, const double2 *MT2g)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6, nthr = blockDim.x;
    const int nx = p.nx, nu = p.nu, nub = p.nub, nuc = p.nuc, T = p.T, nc = p.nc, ncL = p.ncL, nq = p.nq, nr = p.nr, nqT = p.nqT;
    const int nd = p.n_dual, ntv = nc + nq + nr + nub;
    const int o_mu = (T + 1) * nx, o_lb = o_mu + (T - 1) * nc + ncL, o_ub = o_lb + T * nub, o_rho = o_ub + T * nub;
    const int o_sig = o_rho + T * nq + nqT;
    const int e_mu1 = o_mu + (T - 2) * nc, e_mu2 = o_mu + (T - 1) * nc, e_rho1 = o_rho + (T - 1) * nq, e_rho2 = o_rho + T * nq;
    const int np = (nd + 127) / 128, zslot = np * 128, mslot = zslot + 1;
    const int ncL2 = (ncL + 1) / 2;                              // pairs of columns of M_mu (an odd last column is paired with zeros)
    double2 *MT2 = (double2 *)sm;                                // [pair of columns][row]
    double *hs = sm + (size_t)2 * ncL2 * nc, *Mr = hs + nc, *hT = Mr + nq * nqT;
    const size_t maps = ((size_t)(hT + ncL - sm) + 1) / 2 * 2;
    int *map = (int *)(sm + maps);
    const size_t pw = ((size_t)zslot + 1 + nc + nq + 1) / 2 * 2;
    double *row = sm + maps + ((size_t)nd + 3) / 4 * 2 + (size_t)wave * pw;
    double *mapv = row + mslot;
    for (int i = tid; i < ncL2 * nc; i += nthr) MT2[i] = MT2g[i];   // (laid out by hmpc_set_shift_maps)
    for (int i = tid; i < nc; i += nthr) hs[i] = p.h_raw[i];
    for (int i = tid; i < nq * nqT; i += nthr) Mr[i] = p.shift_Mrho[i];
    for (int i = tid; i < ncL; i += nthr) hT[i] = p.hT_raw[i];
    // where output entry i comes from in a wave's buffer (controller.py:615-666: everything moves one stage towards the
    // present, the end of every segment is padded with zeros, the two blocks of the entering stage come from the maps)
    for (int i = tid; i < nd; i += nthr) {
        int sft, lim;
        if (i < o_mu) { sft = nx; lim = T * nx; }
        else if (i < o_lb) { sft = nc; lim = e_mu1; }
        else if (i < o_ub) { sft = nub; lim = o_lb + (T - 1) * nub; }
        else if (i < o_rho) { sft = nub; lim = o_ub + (T - 1) * nub; }
        else if (i < o_sig) { sft = nq; lim = e_rho1; }
        else { sft = nr; lim = o_sig + (T - 1) * nr; }
        map[i] = (i >= e_mu1 && i < e_mu2) ? mslot + (i - e_mu1) : (i >= e_rho1 && i < e_rho2) ? mslot + nc + (i - e_rho1) : i < lim ? i + sft : zslot;
    }
    if (lane == 0) row[zslot] = 0.0;
    __syncthreads();
    const int half = lane >= 32 ? 1 : 0, hl = lane & 31;
    const int stride = gridDim.x * W;
    const int lb_ = min(lane, nub - 1);            // (clamped: no load under a lane predicate)
    // each half-wave sums half of the columns of M_mu (an even count: the pairs stay whole)
    const int kh = (ncL2 + 1) / 2 * 2, k0 = half * kh, k1 = min(k0 + kh, ncL);
    // owner two leaves ahead, identifier head / applied binaries / row index one leaf ahead: the retain rule of a leaf is
    // decided from registers, and the addresses of its row and of its tree's vectors are known when its turn comes
    int b = blockIdx.x * W + wave;
    int own_c = b < a.B ? a.owner[b] : 0, own_n = b + stride < a.B ? a.owner[b + stride] : 0;
    int fix_c = b < a.B ? a.fix[(size_t)b * T * nub + lb_] : -1;
    int src_c = b < a.B ? (a.src ? a.src[b] : b) : 0;
    double ub_c = a.u0[(size_t)own_c * nu + nuc + lb_];
    const __attribute__((address_space(3))) char *row3 = (const __attribute__((address_space(3))) char *)row;
    for (; b < a.B; b += stride) {
        const int bn = b + stride, bnn = b + 2 * stride;
        const int own_nn = bnn < a.B ? a.owner[bnn] : 0;
        const int fix_n = bn < a.B ? a.fix[(size_t)bn * T * nub + lb_] : -1;
        const int src_n = bn < a.B ? (a.src ? a.src[bn] : bn) : 0;
        const double ub_n = a.u0[(size_t)own_n * nu + nuc + lb_];
        const int own = own_c, fix_lane = lane < nub ? fix_c : -1;
        // retain rule (controller.py:566-613): the binaries the leaf fixes at time 0 are the applied ones
        const int agree = __all(fix_lane < 0 || fix_lane == (int)rint(ub_c));
        if (agree) {
            const double *d = a.dual + (size_t)src_c * nd;
            double *o = a.dual_out + (size_t)b * nd;
            // the row: np pieces of 1 KiB, lane l of piece k brings entries 128 k + 2 l and the next one; lanes beyond the end
            // repeat the last pair (it lands in the padding of the last piece); an odd last entry is fetched by itself
            for (int k = 0; k < ((SHIFT_KO & 1) ? 0 : np); k++) {
                const int i = min(k * 128 + 2 * lane, nd - 2);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(d + i),
                                                 (__attribute__((address_space(3))) void *)(row3 + (size_t)k * 1024), 16, 0, 0);
            }
            const double last = d[nd - 1];
            // the tree's vectors: the lanes that use an entry fetch it (first chunk of 32 rows; further chunks below)
            const double *tvo = tv + (size_t)own * ntv;
            const double g_first = tvo[min(hl, nc - 1)], qx_l = tvo[nc + min(lane, nq - 1)], ru_l = tvo[nc + nq + min(lane, nr - 1)];
            const double vu_l = tvo[nc + nq + nr + lb_], e0v = a.e0[(size_t)own * nx + min(lane, nx - 1)];
            const double dobj_in = a.dobj[src_c], lb_in = a.lb[b];
            const int8_t *fix = a.fix + (size_t)b * T * nub;
            // identifier: drop time 0, the stage that enters is free (loaded with the row, stored after the wait: a store before it
            // would put a write acknowledgement into the wait)
            int fb[SHIFT_HEAD_SLOTS];
#pragma unroll
            for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) fb[u] = fix[min(u * 64 + lane + nub, T * nub - 1)];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row has landed (the compiler does not count these loads)
            if (lane == 0) row[nd - 1] = last;
            {
                int8_t *fo = a.fix_out + (size_t)b * T * nub;
                const int nfix = T * nub;
#pragma unroll
                for (int u = 0; u < SHIFT_HEAD_SLOTS; u++) {
                    const int i = min(u * 64 + lane, nfix - 1);
                    fo[i] = i < (T - 1) * nub ? (int8_t)fb[u] : (int8_t)-1;
                }
                for (int i0 = 64 * SHIFT_HEAD_SLOTS; i0 < nfix; i0 += 64) {   // (identifiers beyond 192 binaries)
                    const int i = min(i0 + lane, nfix - 1);
                    const int8_t f = fix[min(i0 + lane + nub, nfix - 1)];
                    fo[i] = i < (T - 1) * nub ? f : (int8_t)-1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double *muL = row + o_mu + (T - 1) * nc, *rhoT = row + o_rho + T * nq;
            double acc = 0.0; // every lane's share of the pi-sum (controller.py:668-721)
            // mu'_{T-2} = M_mu mu_{T-1}: rows in chunks of 32, each half-wave sums half of the columns
            for (int r0 = 0; r0 < nc; r0 += 32) {
                const int r = r0 + hl, rc = min(r, nc - 1);
                double m = 0.0, m2 = 0.0;
                int k = k0;
                for (; k + 8 <= ((SHIFT_KO & 2) ? k0 : k1); k += 8) {   // eight columns at a time, two chains
                    double2 mt[4];
                    double mv[8];
#pragma unroll
                    for (int u = 0; u < 4; u++) mt[u] = MT2[(size_t)((k >> 1) + u) * nc + rc];
#pragma unroll
                    for (int u = 0; u < 8; u++) mv[u] = muL[k + u];
#pragma unroll
                    for (int u = 0; u < 4; u++) { m += mt[u].x * mv[2 * u]; m2 += mt[u].y * mv[2 * u + 1]; }
                }
                for (; k < ((SHIFT_KO & 2) ? k0 : k1); k += 2) {         // (the rest of the half: whole pairs, the last one may hold a zero column)
                    const double2 t = MT2[(size_t)(k >> 1) * nc + rc];
                    m += t.x * muL[k];
                    m2 += t.y * muL[min(k + 1, ncL - 1)];
                }
                m = r < nc ? m + m2 : 0.0;
                m += __shfl_xor(m, 32);
                if (r < nc && half == 0) {
                    mapv[r] = m;
                    // - mu'_{T-2} . h  - mu_0 . (F x0 + G u0 - h)
                    acc -= m * hs[r];
                    acc -= row[o_mu + r] * (r0 == 0 ? g_first : tvo[r]);
                }
            }
            for (int k = lane; k < ncL; k += 64) acc += muL[k] * hT[k]; // + mu_{T-1} . h_Tm1
            for (int r = lane; r < nq; r += 64) {
                double m = 0.0;
                for (int k = 0; k < nqT; k++) m += Mr[r * nqT + k] * rhoT[k];
                mapv[nc + r] = m;
                acc -= 0.25 * m * m; // - |rho'_{T-1}|^2 / 4
                const double qx = r < 64 ? qx_l : tvo[nc + r];
                const double t0 = 0.5 * row[o_rho + r] - qx;
                acc += t0 * t0 - qx * qx; // |rho_0 / 2 - Q x0|^2 - |Q x0|^2
            }
            for (int k = lane; k < nqT; k += 64) acc += 0.25 * rhoT[k] * rhoT[k]; // + |rho_T|^2 / 4
            for (int r = lane; r < nr; r += 64) {
                const double ru = r < 64 ? ru_l : tvo[nc + nq + r];
                const double t0 = 0.5 * row[o_sig + r] - ru;
                acc += t0 * t0 - ru * ru;
            }
            for (int i = lane; i < nub; i += 64) {
                const double vu = i < 64 ? vu_l : tvo[nc + nq + nr + i];
                const int f = i < 64 ? fix_lane : fix[i];
                const double lo = f >= 0 ? (double)f : 0.0, hi = f >= 0 ? (double)f : 1.0;
                acc -= (lo - vu) * row[o_lb + i] + (vu - hi) * row[o_ub + i];
            }
            // model error against the new lam_0 (= old lam_1), controller.py:541-558
            if (lane < nx) acc -= row[nx + lane] * e0v;
            for (int j = lane + 64; j < nx; j += 64) acc -= row[nx + j] * a.e0[(size_t)own * nx + j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // the mapped blocks are in the buffer
            // the copy: every output entry once, batches of SHIFT_CU independent LDS reads and stores, no predicate (lanes beyond
            // the end of the row repeat its last entry); rows of an even length -- every row then starts on a 16-byte boundary --
            // go out two entries per lane.  (Measured and without effect, 4.4 - 4.5 TB/s at 65 536 leaves either way: 16-byte against
            // 8-byte stores, the row fetch and / or the stores non-temporal -- profiles/r05_shift_phases.txt.)
            constexpr int CU_ = SHIFT_CU;
            if ((nd & 1) == 0) {
                const int np2 = nd >> 1;
                const int2 *map2 = (const int2 *)map;
                double2 *o2 = (double2 *)o;
                for (int j0 = 0; j0 < ((SHIFT_KO & 4) ? 0 : np2); j0 += 64 * CU_) {
                    double2 v[CU_];
                    int dst[CU_];
#pragma unroll
                    for (int u = 0; u < CU_; u++) {
                        dst[u] = min(j0 + u * 64 + lane, np2 - 1);
                        const int2 s = map2[dst[u]];
                        v[u] = make_double2(row[s.x], row[s.y]);
                    }
#pragma unroll
                    for (int u = 0; u < CU_; u++) o2[dst[u]] = v[u];
                }
            } else {
                for (int i0 = 0; i0 < ((SHIFT_KO & 4) ? 0 : nd); i0 += 64 * CU_) {
                    double v[CU_];
                    int dst[CU_];
#pragma unroll
                    for (int u = 0; u < CU_; u++) { dst[u] = min(i0 + u * 64 + lane, nd - 1); v[u] = row[map[dst[u]]]; }
#pragma unroll
                    for (int u = 0; u < CU_; u++) o[dst[u]] = v[u];
                }
            }
            const double pi = shift_wave_sum(acc);
            if (lane == 0) {
                double obj = dobj_in + pi;
                obj = obj > 0.0 ? obj : 0.0;
                uint8_t flag = 1;
                double nlb;
                if (!isinf(lb_in)) nlb = obj;                   // a solved / bounded leaf: its bound is the shifted dual objective
                else if (obj <= 0.0) { nlb = 0.0; flag |= 2; } // the infeasibility proof did not survive: reopen
                else nlb = lb_in;                               // still proved infeasible
                a.lb_out[b] = nlb;
                a.dobj_out[b] = obj;
                a.flags[b] = flag;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // the next leaf overwrites the buffer
        } else if (lane == 0) {
            a.flags[b] = 0; // dropped: nothing else is defined for this leaf
        }
        own_c = own_n; own_n = own_nn; fix_c = fix_n; src_c = src_n; ub_c = ub_n;
    }
}
