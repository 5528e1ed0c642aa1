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
