// Batched dense LPs of the offline terminal ingredients -- included by hmpc_capi.hip.
//
//     maximise c_k'x   subject to   A x <= b_k,   x in R^n free,     k = 0 .. B-1,  one matrix A for the batch
//
// is the shape of every LP the reference solves, one Gurobi call at a time, when a controller is built:
//   warm_start_hmpc/mcais.py:103-118       per horizon t, one LP per facet:  max (D A^t)_i x  s.t.  D_inf x <= e_inf
//   warm_start_hmpc/mcais.py:169-182       one LP per facet:                 max E_i x  s.t.  E x <= f + unit_i
//   warm_start_hmpc/controller.py:205-226  one LP per row r_i of [F_Tm1 G_Tm1]:  min h'mu s.t. [F G]'mu = r_i, mu >= 0
//                                          = the multipliers of  max r_i'y  s.t.  [F G] y <= h.
// One workgroup (4 wavefronts) per LP; every sweep of the reference's loops is one launch.
//
// Layout.  A arrives transposed and row-normalised ([n][m], rows of A have unit 2-norm): lane l works on rows
// l, l + 64, ... so that every read of a column of A, from LDS or from L2, is contiguous across the wavefront and
// free of bank conflicts.  Per LP in LDS: eleven row-indexed vectors [m], the n x n normal matrix, nine n-vectors,
// and A itself when it fits beside them (it is the same for all LPs: otherwise it stays in L2).
// Algorithm (identical to oracle/dense_lp.c, the test-side CPU restatement): homogeneous self-dual embedding,
// Mehrotra predictor-corrector on the normal equations A'DA (LDL' with frozen lost pivots), then the purification
// walk to the vertex and the weighted least-norm correction of the multipliers.  Row products and the n(n+1)/2
// entries of A'DA are wave-wide reductions over rows; the two triangular sweeps of a solve run in the registers
// of wavefront 0 (lane i holds component i, v_readlane broadcasts), no barrier per pivot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LP_THREADS 256
#define LP_WAVES 4
#define LP_DELTA 1e-12
#define LP_PROX_STEPS 4

struct LpArgs {
    int n, m, B, c_stride, b_stride, max_iter, a_in_lds;
    double tol;
    const double *At;   // [n][m], rows of A scaled to unit norm
    const double *rs;   // [m] the row scales
    const double *c, *b;
    const int32_t *relax;
    double *obj, *x, *z;
    int32_t *status, *iters;
};

namespace lp {

__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }

struct Ctx {
    int n, m, tid, lane, wave;
    const double *At;
    double *red;   // [4 * LP_WAVES]
};

// up to four sums over the workgroup in one pass (fixed order: reproducible)
__device__ void block_sum4(const Ctx &c, double &a, double &b, double &d, double &e)
{
    a = wave_sum(a); b = wave_sum(b); d = wave_sum(d); e = wave_sum(e);
    __syncthreads();
    if (c.lane == 0) { c.red[c.wave] = a; c.red[4 + c.wave] = b; c.red[8 + c.wave] = d; c.red[12 + c.wave] = e; }
    __syncthreads();
    a = c.red[0] + c.red[1] + c.red[2] + c.red[3];
    b = c.red[4] + c.red[5] + c.red[6] + c.red[7];
    d = c.red[8] + c.red[9] + c.red[10] + c.red[11];
    e = c.red[12] + c.red[13] + c.red[14] + c.red[15];
}
__device__ void block_max4(const Ctx &c, double &a, double &b, double &d, double &e)
{
    a = wave_max(a); b = wave_max(b); d = wave_max(d); e = wave_max(e);
    __syncthreads();
    if (c.lane == 0) { c.red[c.wave] = a; c.red[4 + c.wave] = b; c.red[8 + c.wave] = d; c.red[12 + c.wave] = e; }
    __syncthreads();
    a = fmax(fmax(c.red[0], c.red[1]), fmax(c.red[2], c.red[3]));
    b = fmax(fmax(c.red[4], c.red[5]), fmax(c.red[6], c.red[7]));
    d = fmax(fmax(c.red[8], c.red[9]), fmax(c.red[10], c.red[11]));
    e = fmax(fmax(c.red[12], c.red[13]), fmax(c.red[14], c.red[15]));
}
__device__ double block_min(const Ctx &c, double a)
{
    a = wave_min(a);
    __syncthreads();
    if (c.lane == 0) c.red[c.wave] = a;
    __syncthreads();
    return fmin(fmin(c.red[0], c.red[1]), fmin(c.red[2], c.red[3]));
}

// a_r . x for row r (x in LDS)
__device__ __forceinline__ double row_dot(const Ctx &c, int r, const double *x)
{
    double a = 0;
    for (int j = 0; j < c.n; j++) a += c.At[j * c.m + r] * x[j];
    return a;
}

// out = A'u: column j by wavefront j mod 4, lanes over rows.  Ends with a barrier.
__device__ void at_times(const Ctx &c, const double *u, double *out)
{
    for (int j = c.wave; j < c.n; j += LP_WAVES) {
        double a = 0;
        for (int r = c.lane; r < c.m; r += 64) a += c.At[j * c.m + r] * u[r];
        a = wave_sum(a);
        if (c.lane == 0) out[j] = a;
    }
    __syncthreads();
}

// N = A'diag(D)A + reg I, then L diag(d) L' in place (unit lower L below the diagonal, d on it); a pivot lost to
// cancellation freezes its component (oracle/dense_lp.c normal_factor).  Returns nonzero on a NaN pivot.
__device__ int normal_factor(const Ctx &c, const double *D, double reg, double *N, double *diag, double *col)
{
    const int n = c.n, m = c.m;
    int cnt = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++, cnt++) {
            if ((cnt & (LP_WAVES - 1)) != c.wave) continue;
            double a = 0;
            for (int r = c.lane; r < m; r += 64) a += c.At[i * m + r] * D[r] * c.At[j * m + r];
            a = wave_sum(a);
            if (c.lane == 0) { a += i == j ? reg : 0.0; N[i * n + j] = a; if (i == j) diag[i] = a; }
        }
    __syncthreads();
    int bad = 0;
    for (int k = 0; k < n; k++) {
        double d = N[k * n + k];
        if (d != d) bad = 1;
        const bool frozen = !(d > 1e-13 * diag[k]);
        if (frozen) d = 1e64;
        __syncthreads();                    // everyone has read the pivot
        if (c.tid == 0 && frozen) N[k * n + k] = d;
        for (int i = k + 1 + c.tid; i < n; i += LP_THREADS) { const double l = N[i * n + k] / d; N[i * n + k] = l; col[i] = l; }
        __syncthreads();
        const int W = n - k - 1;
        for (int t = c.tid; t < W * W; t += LP_THREADS) {
            const int i = k + 1 + t / W, j = k + 1 + t % W;
            if (j <= i) N[i * n + j] -= col[i] * (col[j] * d);
        }
        __syncthreads();
    }
    return bad;
}

// v <- N^-1 v with the factor in LDS; the sweeps run in wavefront 0's registers (n <= 64).  Ends with a barrier.
__device__ void normal_solve(const Ctx &c, const double *N, double *v)
{
    const int n = c.n;
    if (c.wave == 0) {
        const int i = c.lane;
        double vi = i < n ? v[i] : 0.0;
        for (int k = 0; k < n; k++) { const double vk = __shfl(vi, k); if (i > k && i < n) vi -= N[i * n + k] * vk; }
        if (i < n) vi /= N[i * n + i];
        for (int k = n - 1; k >= 0; k--) { const double vk = __shfl(vi, k); if (i < k) vi -= N[k * n + i] * vk; }
        if (i < n) v[i] = vi;
    }
    __syncthreads();
}

} // namespace lp

extern "C" __global__ __launch_bounds__(LP_THREADS) void hmpc_lp_kernel(LpArgs a)
{
    using namespace lp;
    extern __shared__ double lp_lds[];
    const int n = a.n, m = a.m, tid = threadIdx.x;
    double *p = lp_lds;
    double *s = p; p += m; double *z = p; p += m; double *D = p; p += m; double *rc = p; p += m; double *rhs = p; p += m;
    double *z1 = p; p += m; double *z2 = p; p += m; double *dsa = p; p += m; double *dza = p; p += m; double *ds = p; p += m;
    double *bb = p; p += m;
    double *N = p; p += n * n;
    double *x = p; p += n; double *rd = p; p += n; double *x1 = p; p += n; double *x2 = p; p += n; double *t = p; p += n;
    double *q = p; p += n; double *xp = p; p += n; double *diag = p; p += n; double *col = p; p += n;
    double *red = p; p += 4 * LP_WAVES;
    Ctx c; c.n = n; c.m = m; c.tid = tid; c.lane = tid & 63; c.wave = tid >> 6; c.red = red; c.At = a.At;
    if (a.a_in_lds) {
        double *Al = p;
        for (int i = tid; i < n * m; i += LP_THREADS) Al[i] = a.At[i];
        c.At = Al;
    }

    for (int k = blockIdx.x; k < a.B; k += gridDim.x) {
        const double *ck = a.c + (size_t)k * a.c_stride, *bk = a.b + (size_t)k * a.b_stride;
        const int rel = a.relax ? a.relax[k] : -1;
        double cinf = 0, binf = 0, u0 = 0, u1 = 0;
        __syncthreads();
        for (int j = tid; j < n; j += LP_THREADS) { cinf = fmax(cinf, fabs(ck[j])); x[j] = 0; }
        for (int r = tid; r < m; r += LP_THREADS) {
            const double v = (bk[r] + (r == rel ? 1.0 : 0.0)) * a.rs[r];
            bb[r] = v; binf = fmax(binf, fabs(v)); s[r] = 1; z[r] = 1;
        }
        block_max4(c, cinf, binf, u0, u1);
        if (!(cinf > 0)) cinf = 1;
        for (int j = tid; j < n; j += LP_THREADS) q[j] = -ck[j] / cinf;
        __syncthreads();

        double tau = 1, kap = 1;
        int status = 2 /* MAXITER */, it;
        for (it = 0; it <= a.max_iter; it++) {
            at_times(c, z, rd);
            double qx = 0, bz = 0, sz = 0, zero = 0, rcinf = 0, rdinf = 0, xinf = 0, zinf = 0, tinf = 0, worst = 0;
            for (int j = tid; j < n; j += LP_THREADS) {
                tinf = fmax(tinf, fabs(rd[j]));
                const double v = rd[j] + q[j] * tau; rd[j] = v;
                rdinf = fmax(rdinf, fabs(v)); xinf = fmax(xinf, fabs(x[j])); qx += q[j] * x[j];
            }
            for (int r = tid; r < m; r += LP_THREADS) {
                const double ax = row_dot(c, r, x), v = ax + s[r] - bb[r] * tau;
                rc[r] = v; rcinf = fmax(rcinf, fabs(v)); zinf = fmax(zinf, fabs(z[r])); worst = fmax(worst, ax);
                bz += bb[r] * z[r]; sz += s[r] * z[r];
            }
            block_sum4(c, qx, bz, sz, zero);
            block_max4(c, rcinf, rdinf, xinf, zinf);
            block_max4(c, tinf, worst, u0, u1);
            const double rg = qx + bz + kap, mu = (sz + tau * kap) / (m + 1);
            xinf /= tau; zinf /= tau;
            const double pobj = qx / tau, dobj = -bz / tau;
            if (rcinf / tau <= a.tol * (1 + xinf + binf) && rdinf / tau <= a.tol * (1 + zinf) &&
                fabs(pobj - dobj) <= a.tol * (1 + fmin(fabs(pobj), fabs(dobj)))) { status = 0; break; }
            if (bz < 0 && (tinf <= 1e-7 * (-bz) || (tau <= 1e-8 * kap && tinf <= 1e-3 * (-bz)))) { status = 1; break; }
            if (qx < 0 && (worst <= 1e-7 * (-qx) || (tau <= 1e-8 * kap && worst <= 1e-3 * (-qx)))) { status = 4; break; }
            if (it == a.max_iter) break;

            for (int r = tid; r < m; r += LP_THREADS) { const double d = z[r] / s[r]; D[r] = d; rhs[r] = d * bb[r]; }
            __syncthreads();
            if (normal_factor(c, D, 1e-14, N, diag, col)) { status = 3; break; }
            // constant direction: N x1 = A'D b - q, z1 = D (A x1 - b)
            at_times(c, rhs, x1);
            for (int j = tid; j < n; j += LP_THREADS) x1[j] -= q[j];
            __syncthreads();
            normal_solve(c, N, x1);
            double qx1 = 0, bz1 = 0, e2 = 0, e3 = 0;
            for (int j = tid; j < n; j += LP_THREADS) qx1 += q[j] * x1[j];
            for (int r = tid; r < m; r += LP_THREADS) { const double v = D[r] * (row_dot(c, r, x1) - bb[r]); z1[r] = v; bz1 += bb[r] * v; }
            block_sum4(c, qx1, bz1, e2, e3);
            const double den = kap / tau - qx1 - bz1;

            double sigma = 0, dtau_a = 0, dkap_a = 0, alpha = 0, dtau = 0, dkap = 0;
            for (int pass = 0; pass < 2; pass++) {
                const double lin = pass == 0 ? 1.0 : 1.0 - sigma;
                const double dkap_rhs = tau * kap + (pass ? dtau_a * dkap_a - sigma * mu : 0.0);
                for (int r = tid; r < m; r += LP_THREADS) {
                    const double dsr = s[r] * z[r] + (pass ? dsa[r] * dza[r] - sigma * mu : 0.0);
                    const double v = lin * rc[r] - dsr / z[r];
                    rhs[r] = v; ds[r] = D[r] * v;
                }
                __syncthreads();
                at_times(c, ds, x2);
                for (int j = tid; j < n; j += LP_THREADS) x2[j] = -lin * rd[j] - x2[j];
                __syncthreads();
                normal_solve(c, N, x2);
                double qx2 = 0, bz2 = 0;
                e2 = 0; e3 = 0;
                for (int j = tid; j < n; j += LP_THREADS) qx2 += q[j] * x2[j];
                for (int r = tid; r < m; r += LP_THREADS) { const double v = D[r] * (row_dot(c, r, x2) + rhs[r]); z2[r] = v; bz2 += bb[r] * v; }
                block_sum4(c, qx2, bz2, e2, e3);
                dtau = (lin * rg + qx2 + bz2 - dkap_rhs / tau) / den;
                dkap = -(dkap_rhs + kap * dtau) / tau;
                double amax = 1e30;
                if (dtau < 0) amax = fmin(amax, -tau / dtau);
                if (dkap < 0) amax = fmin(amax, -kap / dkap);
                for (int r = tid; r < m; r += LP_THREADS) {
                    const double dz = z2[r] + dtau * z1[r];
                    const double dsr = s[r] * z[r] + (pass ? dsa[r] * dza[r] - sigma * mu : 0.0);
                    const double dsv = -(dsr + s[r] * dz) / z[r];
                    if (dz < 0) amax = fmin(amax, -z[r] / dz);
                    if (dsv < 0) amax = fmin(amax, -s[r] / dsv);
                    z2[r] = dz;
                    if (pass == 0) { dza[r] = dz; dsa[r] = dsv; } else ds[r] = dsv;
                }
                amax = block_min(c, amax);
                if (pass == 0) { const double aa = fmin(1.0, amax); sigma = (1 - aa) * (1 - aa) * (1 - aa); dtau_a = dtau; dkap_a = dkap; }
                else alpha = fmin(1.0, 0.99 * amax);
            }
            for (int j = tid; j < n; j += LP_THREADS) x[j] += alpha * (x2[j] + dtau * x1[j]);
            for (int r = tid; r < m; r += LP_THREADS) { z[r] += alpha * z2[r]; s[r] += alpha * ds[r]; }
            tau += alpha * dtau; kap += alpha * dkap;
            __syncthreads();
            if (!(tau > 0) || !(kap >= 0)) { status = 3; break; }
        }

        if (status == 0) {
            for (int j = tid; j < n; j += LP_THREADS) { x[j] /= tau; xp[j] = x[j]; }
            int nact = 0;
            for (int r = tid; r < m; r += LP_THREADS) { z[r] /= tau; s[r] /= tau; const bool on = z[r] > s[r]; dsa[r] = on ? 1.0 : 0.0; nact += on; }
            double na = nact, e1 = 0, e2 = 0, e3 = 0;
            block_sum4(c, na, e1, e2, e3);
            // purification: projection on the active rows, walk along what is left of the cost to the next row
            bool ok = na > 0;
            for (int round = 0; ok && round <= n; round++) {
                if (normal_factor(c, dsa, LP_DELTA, N, diag, col)) { ok = false; break; }
                for (int kk = 0; kk < LP_PROX_STEPS; kk++) {
                    for (int r = tid; r < m; r += LP_THREADS) rhs[r] = dsa[r] * (bb[r] - row_dot(c, r, xp));
                    __syncthreads();
                    at_times(c, rhs, t);
                    normal_solve(c, N, t);
                    for (int j = tid; j < n; j += LP_THREADS) xp[j] += t[j];
                    __syncthreads();
                }
                for (int j = tid; j < n; j += LP_THREADS) x2[j] = -q[j];
                __syncthreads();
                for (int kk = 0; kk < 2; kk++) {
                    normal_solve(c, N, x2);
                    for (int j = tid; j < n; j += LP_THREADS) x2[j] *= LP_DELTA;
                    __syncthreads();
                }
                double dinf = 0;
                e1 = e2 = e3 = 0;
                for (int j = tid; j < n; j += LP_THREADS) dinf = fmax(dinf, fabs(x2[j]));
                block_max4(c, dinf, e1, e2, e3);
                if (dinf <= 1e-10 || round == n) break;
                double step = 1e300;
                for (int r = tid; r < m; r += LP_THREADS) {
                    double cand = 1e300;
                    if (!(dsa[r] > 0)) {
                        const double ad = row_dot(c, r, x2);
                        if (ad > 1e-13) cand = fmax(bb[r] - row_dot(c, r, xp), 0.0) / ad;
                    }
                    rhs[r] = cand; step = fmin(step, cand);
                }
                step = block_min(c, step);
                if (!(step < 1e300)) break;
                double first = 1e300;   // the lowest-numbered row among those that block at this step
                for (int r = tid; r < m; r += LP_THREADS) if (rhs[r] == step) first = fmin(first, (double)r);
                first = block_min(c, first);
                for (int j = tid; j < n; j += LP_THREADS) xp[j] += step * x2[j];
                if (tid == 0) dsa[(int)first] = 1.0;
                __syncthreads();
            }
            if (ok) {
                double viol = 0, move = 0, xin = 0, gain = 0, px = 0;
                e1 = e2 = 0;
                for (int r = tid; r < m; r += LP_THREADS) viol = fmax(viol, row_dot(c, r, xp) - bb[r]);
                for (int j = tid; j < n; j += LP_THREADS) {
                    move = fmax(move, fabs(xp[j] - x[j])); xin = fmax(xin, fabs(x[j]));
                    gain += -q[j] * (xp[j] - x[j]); px += q[j] * x[j];
                }
                e3 = 0;
                block_max4(c, viol, move, xin, e3);
                block_sum4(c, gain, px, e1, e2);
                if (viol <= 1e-11 * (1 + binf) && gain >= -1e-8 * (1 + fabs(px)) && move <= 1e-2 * (1 + xin))
                    for (int j = tid; j < n; j += LP_THREADS) x[j] = xp[j];
                __syncthreads();
            }
            // multipliers: inactive rows to zero, weighted least-norm correction so that A'z = c to rounding
            for (int r = tid; r < m; r += LP_THREADS) { D[r] = z[r] / s[r]; if (!(z[r] > s[r])) z[r] = 0; }
            __syncthreads();
            if (normal_factor(c, D, 1e-14, N, diag, col) == 0) {
                for (int kk = 0; kk < 2; kk++) {
                    at_times(c, z, t);
                    for (int j = tid; j < n; j += LP_THREADS) t[j] = -q[j] - t[j];
                    __syncthreads();
                    normal_solve(c, N, t);
                    for (int r = tid; r < m; r += LP_THREADS) { const double zr = z[r] + D[r] * row_dot(c, r, t); z[r] = zr > 0 ? zr : 0; }
                    __syncthreads();
                }
            }
        } else if (status == 1 || status == 4) {   // rays, largest entry 1
            double big = 0, e1 = 0, e2 = 0, e3 = 0;
            if (status == 1) for (int r = tid; r < m; r += LP_THREADS) big = fmax(big, fabs(z[r]));
            else for (int j = tid; j < n; j += LP_THREADS) big = fmax(big, fabs(x[j]));
            block_max4(c, big, e1, e2, e3);
            if (status == 1) for (int r = tid; r < m; r += LP_THREADS) z[r] /= big;
            else for (int j = tid; j < n; j += LP_THREADS) x[j] /= big;
            __syncthreads();
        }

        double v = 0, e1 = 0, e2 = 0, e3 = 0;
        for (int j = tid; j < n; j += LP_THREADS) { a.x[(size_t)k * n + j] = x[j]; v += ck[j] * x[j]; }
        block_sum4(c, v, e1, e2, e3);
        if (tid == 0) { a.obj[k] = status == 0 ? v : __builtin_nan(""); a.status[k] = status; a.iters[k] = it; }
        if (a.z) for (int r = tid; r < m; r += LP_THREADS) a.z[(size_t)k * m + r] = z[r] * a.rs[r] * (status == 0 ? cinf : 1.0);
    }
}

static size_t hmpc_lp_lds_bytes(int n, int m, int a_in_lds)
{
    return sizeof(double) * ((size_t)11 * m + (size_t)n * n + (size_t)9 * n + 4 * LP_WAVES + (a_in_lds ? (size_t)n * m : 0));
}

// ---- C ABI (declared in include/hmpc.h) ----
namespace {
struct LpBuffers {   // freed on every exit path
    std::vector<void *> ptrs;
    ~LpBuffers() { for (void *q : ptrs) if (q) (void)hipFree(q); }
    template <class T> hipError_t get(T **out, size_t count)
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(q);
        *out = (T *)q;
        return e;
    }
};
} // namespace

extern "C" int hmpc_lp_solve_batch(int32_t device, int32_t n, int32_t m, const double *A, const double *c, int32_t c_stride,
                                   const double *b, int32_t b_stride, const int32_t *relax, int32_t B, double tol,
                                   int32_t max_iter, double *obj, double *x, double *z, int32_t *status, int32_t *iters)
{
    g_err.clear();
    if (n < 1 || n > 64 || m < 1 || B < 0 || !A || !c || !b || !obj || !x || !status || !iters)
        return fail(HMPC_EINVAL, "lp: bad argument (1 <= n <= 64, m >= 1, non-null arrays)");
    if ((c_stride != 0 && c_stride != n) || (b_stride != 0 && b_stride != m)) return fail(HMPC_EINVAL, "lp: strides are 0 (shared) or the row length");
    if (relax) for (int k = 0; k < B; k++) if (relax[k] < -1 || relax[k] >= m) return fail(HMPC_EINVAL, "lp: relaxed row out of range");
    if (B == 0) return HMPC_OK;
    if (tol <= 0) tol = 1e-9;
    if (max_iter <= 0) max_iter = 100;
    if (device >= 0) HIPCHK(hipSetDevice(device));
    int dev = 0, cus = 0, lds_max = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIPCHK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    if (lds_max < 160 * 1024) lds_max = 64 * 1024;
    if (hmpc_lp_lds_bytes(n, m, 0) > (size_t)lds_max) return fail(HMPC_ETOOBIG, "lp: the row vectors of one LP exceed one CU's LDS");

    // rows to unit 2-norm, transposed: the kernel reads columns of A along the rows
    std::vector<double> At((size_t)n * m), rs(m);
    for (int r = 0; r < m; r++) {
        double a = 0;
        for (int j = 0; j < n; j++) a += A[(size_t)r * n + j] * A[(size_t)r * n + j];
        rs[r] = a > 0 ? 1.0 / std::sqrt(a) : 1.0;
        for (int j = 0; j < n; j++) At[(size_t)j * m + r] = A[(size_t)r * n + j] * rs[r];
    }
    LpBuffers buf;
    LpArgs a{};
    a.n = n; a.m = m; a.B = B; a.c_stride = c_stride; a.b_stride = b_stride; a.max_iter = max_iter; a.tol = tol;
    a.a_in_lds = hmpc_lp_lds_bytes(n, m, 1) <= (size_t)lds_max;
    double *dAt, *drs, *dc, *db;
    int32_t *drelax = nullptr;
    const size_t nc = c_stride ? (size_t)B * n : n, nb = b_stride ? (size_t)B * m : m;
    HIPCHK(buf.get(&dAt, At.size())); HIPCHK(buf.get(&drs, m)); HIPCHK(buf.get(&dc, nc)); HIPCHK(buf.get(&db, nb));
    HIPCHK(buf.get(&a.obj, B)); HIPCHK(buf.get(&a.x, (size_t)B * n)); HIPCHK(buf.get(&a.status, B)); HIPCHK(buf.get(&a.iters, B));
    if (z) HIPCHK(buf.get(&a.z, (size_t)B * m));
    if (relax) { HIPCHK(buf.get(&drelax, B)); HIPCHK(hipMemcpy(drelax, relax, sizeof(int32_t) * B, hipMemcpyHostToDevice)); }
    HIPCHK(hipMemcpy(dAt, At.data(), sizeof(double) * At.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(drs, rs.data(), sizeof(double) * m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc, c, sizeof(double) * nc, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, sizeof(double) * nb, hipMemcpyHostToDevice));
    a.At = dAt; a.rs = drs; a.c = dc; a.b = db; a.relax = drelax;

    const size_t lds = hmpc_lp_lds_bytes(n, m, a.a_in_lds);
    HIPCHK(hipFuncSetAttribute((const void *)hmpc_lp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)lds_max / lds));
    const int grid = std::min(B, cus * per_cu);
    hipLaunchKernelGGL(hmpc_lp_kernel, dim3(grid), dim3(LP_THREADS), lds, 0, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(obj, a.obj, sizeof(double) * B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(x, a.x, sizeof(double) * (size_t)B * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, a.status, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(iters, a.iters, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (z) HIPCHK(hipMemcpy(z, a.z, sizeof(double) * (size_t)B * m, hipMemcpyDeviceToHost));
    return HMPC_OK;
}
