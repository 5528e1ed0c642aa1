/* CPU oracle for the facet LPs of the offline terminal ingredients -- TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file; the
 * product path (csrc/hmpc_lp.hip behind hmpc_lp_solve_batch) never does.
 *
 * What it restates (reference = /root/reference):
 *   warm_start_hmpc/mcais.py:103-118   one LP per facet and horizon:  max J_i x  s.t.  D_inf x <= e_inf
 *   warm_start_hmpc/mcais.py:169-182   one LP per facet:              max E_i x  s.t.  E x <= f + unit_i
 *   warm_start_hmpc/controller.py:205-226  one LP per row of [F_Tm1 G_Tm1]:  min h'mu s.t. [F G]'mu = r_i, mu >= 0,
 *       solved here in its dual form  max r_i'y s.t. [F G] y <= h  (mu = the multipliers z of the rows)
 * all of the shape
 *       maximise c_k'x   subject to   A x <= b_k,   x free,       k = 0 .. B-1, one matrix A for the batch.
 * The reference solves them with Gurobi (absent here); the algorithm below is the published homogeneous self-dual
 * embedding with Mehrotra's predictor-corrector (Xu, Hung, Ye 1996; Andersen & Andersen 2000) on the normal
 * equations A'DA, followed by a projection on the active rows that makes the optimal value exact.
 * Pinned against an independent solver (HiGHS through scipy) and against the committed terminal sets, which HiGHS
 * produced: tests/test_terminal_lp.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LP_OPTIMAL 0
#define LP_INFEASIBLE 1
#define LP_MAXITER 2
#define LP_NUMERICAL 3
#define LP_UNBOUNDED 4
#define LP_DELTA 1e-12   /* proximal weight of the purification */
#define LP_PROX_STEPS 4

typedef struct {
    int n, m;
    double *s, *z, *D, *rc, *rhs, *z1, *z2, *dsa, *dza, *ds;   /* m */
    double *x, *rd, *x1, *x2, *t, *q, *xp;                    /* n */
    double *N, *L;                                            /* n x n */
} lpw_t;

static double *dal(size_t k) { return (double *)calloc(k ? k : 1, sizeof(double)); }

static lpw_t *lpw_create(int n, int m)
{
    lpw_t *w = (lpw_t *)calloc(1, sizeof(lpw_t));
    w->n = n; w->m = m;
    w->s = dal(m); w->z = dal(m); w->D = dal(m); w->rc = dal(m); w->rhs = dal(m); w->z1 = dal(m); w->z2 = dal(m);
    w->dsa = dal(m); w->dza = dal(m); w->ds = dal(m);
    w->x = dal(n); w->rd = dal(n); w->x1 = dal(n); w->x2 = dal(n); w->t = dal(n); w->q = dal(n); w->xp = dal(n);
    w->N = dal((size_t)n * n); w->L = dal((size_t)n * n);
    return w;
}

static void lpw_free(lpw_t *w)
{
    free(w->s); free(w->z); free(w->D); free(w->rc); free(w->rhs); free(w->z1); free(w->z2); free(w->dsa); free(w->dza); free(w->ds);
    free(w->x); free(w->rd); free(w->x1); free(w->x2); free(w->t); free(w->q); free(w->xp); free(w->N); free(w->L); free(w);
}

/* N = A' diag(D) A + reg I, factorised as L diag(d) L' in place (L unit lower, d on the diagonal of w->L). */
static int normal_factor(lpw_t *w, const double *A, const double *D, double reg)
{
    const int n = w->n, m = w->m;
    double *N = w->L;
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
        double acc = i == j ? reg : 0.0;
        for (int r = 0; r < m; r++) acc += A[r * n + i] * D[r] * A[r * n + j];
        N[i * n + j] = acc;
    }
    for (int k = 0; k < n; k++) w->t[k] = N[k * n + k];
    for (int k = 0; k < n; k++) {
        double d = N[k * n + k];
        if (d != d) return 1;
        /* a pivot lost to cancellation (weights span > 16 orders at a degenerate vertex): freeze the component */
        if (!(d > 1e-13 * w->t[k])) { d = 1e64; N[k * n + k] = d; }
        for (int i = k + 1; i < n; i++) N[i * n + k] /= d;
        for (int j = k + 1; j < n; j++) {
            double ljk = N[j * n + k] * d;
            for (int i = j; i < n; i++) N[i * n + j] -= N[i * n + k] * ljk;
        }
    }
    return 0;
}

static void normal_solve(const lpw_t *w, double *v)
{
    const int n = w->n; const double *L = w->L;
    for (int i = 0; i < n; i++) { double a = v[i]; for (int j = 0; j < i; j++) a -= L[i * n + j] * v[j]; v[i] = a; }
    for (int i = 0; i < n; i++) v[i] /= L[i * n + i];
    for (int i = n - 1; i >= 0; i--) { double a = v[i]; for (int j = i + 1; j < n; j++) a -= L[j * n + i] * v[j]; v[i] = a; }
}

/* v = A' u */
static void at_times(const lpw_t *w, const double *A, const double *u, double *v)
{
    for (int j = 0; j < w->n; j++) v[j] = 0;
    for (int r = 0; r < w->m; r++) { double ur = u[r]; if (ur != 0.0) for (int j = 0; j < w->n; j++) v[j] += A[r * w->n + j] * ur; }
}

static double row_dot(const lpw_t *w, const double *A, int r, const double *x)
{
    double a = 0; for (int j = 0; j < w->n; j++) a += A[r * w->n + j] * x[j]; return a;
}

static double vinf(const double *v, int k) { double a = 0; for (int i = 0; i < k; i++) if (fabs(v[i]) > a) a = fabs(v[i]); return a; }

static double pobj_final(const lpw_t *w, const double *q) { double a = 0; for (int j = 0; j < w->n; j++) a += q[j] * w->x[j]; return a; }

/* One LP in scaled data: rows of A have unit 2-norm, q = -c / |c|_inf.  Returns status; x, z in scaled units, tau = 1. */
static int lp_one(lpw_t *w, const double *A, const double *q, const double *b, double tol, int max_iter, int *iters)
{
    const int n = w->n, m = w->m;
    double tau = 1, kap = 1;
    for (int j = 0; j < n; j++) w->x[j] = 0;
    for (int r = 0; r < m; r++) { w->s[r] = 1; w->z[r] = 1; }
    const double binf = vinf(b, m);
    int status = LP_MAXITER, it;
    for (it = 0; it <= max_iter; it++) {
        /* residuals of the embedding: rd = A'z + q tau, rc = A x + s - b tau, rg = q'x + b'z + kappa */
        at_times(w, A, w->z, w->rd);
        double qx = 0, bz = 0, sz = 0;
        for (int j = 0; j < n; j++) { w->rd[j] += q[j] * tau; qx += q[j] * w->x[j]; }
        for (int r = 0; r < m; r++) { w->rc[r] = row_dot(w, A, r, w->x) + w->s[r] - b[r] * tau; bz += b[r] * w->z[r]; sz += w->s[r] * w->z[r]; }
        const double rg = qx + bz + kap, mu = (sz + tau * kap) / (m + 1);
        const double xinf = vinf(w->x, n) / tau, zinf = vinf(w->z, m) / tau;
        const double pobj = qx / tau, dobj = -bz / tau;
        if (vinf(w->rc, m) / tau <= tol * (1 + xinf + binf) && vinf(w->rd, n) / tau <= tol * (1 + zinf) &&
            fabs(pobj - dobj) <= tol * (1 + fmin(fabs(pobj), fabs(dobj)))) { status = LP_OPTIMAL; break; }
        /* certificates: empty set (A'z = 0, b'z < 0) / unbounded (A x <= 0, q'x < 0) */
        if (bz < 0) { at_times(w, A, w->z, w->t); if (vinf(w->t, n) <= 1e-7 * (-bz) || (tau <= 1e-8 * kap && vinf(w->t, n) <= 1e-3 * (-bz))) { status = LP_INFEASIBLE; break; } }
        if (qx < 0) {
            double worst = 0;
            for (int r = 0; r < m; r++) { double a = row_dot(w, A, r, w->x); if (a > worst) worst = a; }
            if (worst <= 1e-7 * (-qx) || (tau <= 1e-8 * kap && worst <= 1e-3 * (-qx))) { status = LP_UNBOUNDED; break; }
        }
        if (it == max_iter) break;

        for (int r = 0; r < m; r++) w->D[r] = w->z[r] / w->s[r];
        if (normal_factor(w, A, w->D, 1e-14)) { status = LP_NUMERICAL; break; }
        /* constant direction: N x1 = A'D b - q, z1 = D (A x1 - b) */
        for (int r = 0; r < m; r++) w->rhs[r] = w->D[r] * b[r];
        at_times(w, A, w->rhs, w->x1);
        for (int j = 0; j < n; j++) w->x1[j] -= q[j];
        normal_solve(w, w->x1);
        double qx1 = 0, bz1 = 0;
        for (int j = 0; j < n; j++) qx1 += q[j] * w->x1[j];
        for (int r = 0; r < m; r++) { w->z1[r] = w->D[r] * (row_dot(w, A, r, w->x1) - b[r]); bz1 += b[r] * w->z1[r]; }
        const double den = kap / tau - qx1 - bz1;

        double sigma = 0, dtau_a = 0, dkap_a = 0, alpha = 0, dtau = 0, dkap = 0;
        for (int pass = 0; pass < 2; pass++) {
            const double lin = pass == 0 ? 1.0 : 1.0 - sigma;
            const double dkap_rhs = tau * kap + (pass ? dtau_a * dkap_a - sigma * mu : 0.0);
            /* N x2 = -lin rd - A'D (lin rc - dsr / z),  z2 = D (A x2 + lin rc - dsr / z) */
            for (int r = 0; r < m; r++) {
                const double dsr = w->s[r] * w->z[r] + (pass ? w->dsa[r] * w->dza[r] - sigma * mu : 0.0);
                w->rhs[r] = lin * w->rc[r] - dsr / w->z[r];
                w->ds[r] = w->D[r] * w->rhs[r];
            }
            at_times(w, A, w->ds, w->x2);
            for (int j = 0; j < n; j++) w->x2[j] = -lin * w->rd[j] - w->x2[j];
            normal_solve(w, w->x2);
            double qx2 = 0, bz2 = 0;
            for (int j = 0; j < n; j++) qx2 += q[j] * w->x2[j];
            for (int r = 0; r < m; r++) { w->z2[r] = w->D[r] * (row_dot(w, A, r, w->x2) + w->rhs[r]); bz2 += b[r] * w->z2[r]; }
            dtau = (lin * rg + qx2 + bz2 - dkap_rhs / tau) / den;
            dkap = -(dkap_rhs + kap * dtau) / tau;
            double amax = 1e30;
            if (dtau < 0) amax = fmin(amax, -tau / dtau);
            if (dkap < 0) amax = fmin(amax, -kap / dkap);
            for (int r = 0; r < m; r++) {
                const double dz = w->z2[r] + dtau * w->z1[r];
                const double dsr = w->s[r] * w->z[r] + (pass ? w->dsa[r] * w->dza[r] - sigma * mu : 0.0);
                const double ds = -(dsr + w->s[r] * dz) / w->z[r];
                if (dz < 0) amax = fmin(amax, -w->z[r] / dz);
                if (ds < 0) amax = fmin(amax, -w->s[r] / ds);
                w->z2[r] = dz;
                if (pass == 0) { w->dza[r] = dz; w->dsa[r] = ds; } else w->ds[r] = ds;
            }
            if (pass == 0) { const double aa = fmin(1.0, amax); sigma = (1 - aa) * (1 - aa) * (1 - aa); dtau_a = dtau; dkap_a = dkap; }
            else alpha = fmin(1.0, 0.99 * amax);
        }
        for (int j = 0; j < n; j++) w->x[j] += alpha * (w->x2[j] + dtau * w->x1[j]);
        for (int r = 0; r < m; r++) { w->z[r] += alpha * w->z2[r]; w->s[r] += alpha * w->ds[r]; }
        tau += alpha * dtau; kap += alpha * dkap;
        if (!(tau > 0) || !(kap >= 0)) { status = LP_NUMERICAL; break; }
    }
    *iters = it;
    if (status == LP_OPTIMAL) {
        for (int j = 0; j < n; j++) w->x[j] /= tau;
        for (int r = 0; r < m; r++) { w->z[r] /= tau; w->s[r] /= tau; }
        /* Purification: projection on the active rows (z > s) by proximal least squares -- LP_PROX_STEPS steps of
         * x += (A_a'A_a + delta I)^-1 A_a'(b_a - A_a x) --, then, while the cost still has a component d along the
         * face those rows leave free (weakly active rows the iterate has not reached: z s ~ mu on them), a walk
         * along d to the first blocking row, which joins the set.  Ends at the point a vertex solver returns whenever
         * the optimal value is attained at one; kept only if every row holds.  The decisions of mcais.py:128 and
         * :181 compare the optimal value with 0 and 1e-7. */
        int nact = 0;
        for (int r = 0; r < m; r++) { w->dsa[r] = w->z[r] > w->s[r] ? 1.0 : 0.0; nact += w->dsa[r] > 0; }
        memcpy(w->xp, w->x, sizeof(double) * n);
        int ok = nact > 0;
        for (int round = 0; ok && round <= n; round++) {
            if (normal_factor(w, A, w->dsa, LP_DELTA)) { ok = 0; break; }
            for (int k = 0; k < LP_PROX_STEPS; k++) {
                for (int r = 0; r < m; r++) w->rhs[r] = w->dsa[r] * (b[r] - row_dot(w, A, r, w->xp));
                at_times(w, A, w->rhs, w->t);
                normal_solve(w, w->t);
                for (int j = 0; j < n; j++) w->xp[j] += w->t[j];
            }
            for (int j = 0; j < n; j++) w->x2[j] = -q[j];
            for (int k = 0; k < 2; k++) {   /* delta N^-1 twice: the row-space part of the cost shrinks to delta^2 */
                normal_solve(w, w->x2);
                for (int j = 0; j < n; j++) w->x2[j] *= LP_DELTA;
            }
            if (vinf(w->x2, n) <= 1e-10 || round == n) break;
            double step = 1e300; int blk = -1;
            for (int r = 0; r < m; r++) {
                if (w->dsa[r] > 0) continue;
                const double ad = row_dot(w, A, r, w->x2);
                if (ad > 1e-13) { const double sl = fmax(b[r] - row_dot(w, A, r, w->xp), 0.0); if (sl / ad < step) { step = sl / ad; blk = r; } }
            }
            if (blk < 0) break;
            for (int j = 0; j < n; j++) w->xp[j] += step * w->x2[j];
            w->dsa[blk] = 1.0;
        }
        if (ok) {
            double viol = 0, move = 0, gain = 0;
            for (int r = 0; r < m; r++) { double a = row_dot(w, A, r, w->xp) - b[r]; if (a > viol) viol = a; }
            for (int j = 0; j < n; j++) { move = fmax(move, fabs(w->xp[j] - w->x[j])); gain += -q[j] * (w->xp[j] - w->x[j]); }
            if (viol <= 1e-11 * (1 + binf) && gain >= -1e-8 * (1 + fabs(pobj_final(w, q))) && move <= 1e-2 * (1 + vinf(w->x, n))) memcpy(w->x, w->xp, sizeof(double) * n);
        }
        /* Multipliers: inactive rows to zero, then the weighted least-norm correction dz = D A (A'DA)^-1 (-q - A'z)
         * with the weights of the last iterate, so that A'z = c holds to rounding (controller.py:205-226 uses z as
         * the column of M with [F G]'M = [F_Tm1 G_Tm1]'). */
        for (int r = 0; r < m; r++) { w->D[r] = w->z[r] / w->s[r]; if (!(w->z[r] > w->s[r])) w->z[r] = 0; }
        if (normal_factor(w, A, w->D, 1e-14) == 0) {
            for (int k = 0; k < 2; k++) {
                at_times(w, A, w->z, w->t);
                for (int j = 0; j < n; j++) w->t[j] = -q[j] - w->t[j];
                normal_solve(w, w->t);
                for (int r = 0; r < m; r++) { double zr = w->z[r] + w->D[r] * row_dot(w, A, r, w->t); w->z[r] = zr > 0 ? zr : 0; }
            }
        }
    } else if (status == LP_INFEASIBLE) {
        double big = vinf(w->z, m); for (int r = 0; r < m; r++) w->z[r] /= big;
    } else if (status == LP_UNBOUNDED) {
        double big = vinf(w->x, n); for (int j = 0; j < n; j++) w->x[j] /= big;
    }
    return status;
}

/* maximise c_k'x s.t. A x <= b_k (+1 on row relax[k] when relax is given); strides 0 share one vector over the batch.
 * obj = c'x (NaN unless optimal), x[B][n], z[B][m] >= 0 with A'z = c at an optimum (a ray otherwise). */
int oracle_lp_batch(int n, int m, const double *A, const double *c, int c_stride, const double *b, int b_stride,
                    const int32_t *relax, int B, double tol, int max_iter, int threads,
                    double *obj, double *x, double *z, int32_t *status, int32_t *iters)
{
    if (n < 1 || m < 1 || B < 0) return -1;
    double *As = dal((size_t)m * n), *rs = dal(m);
    for (int r = 0; r < m; r++) {
        double a = 0; for (int j = 0; j < n; j++) a += A[r * n + j] * A[r * n + j];
        rs[r] = a > 0 ? 1.0 / sqrt(a) : 1.0;
        for (int j = 0; j < n; j++) As[r * n + j] = A[r * n + j] * rs[r];
    }
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int k = 0; k < B; k++) {
        lpw_t *w = lpw_create(n, m);
        double *bs = dal(m);
        const double *ck = c + (size_t)k * c_stride, *bk = b + (size_t)k * b_stride;
        double cinf = vinf(ck, n); if (!(cinf > 0)) cinf = 1;
        for (int j = 0; j < n; j++) w->q[j] = -ck[j] / cinf;
        for (int r = 0; r < m; r++) bs[r] = (bk[r] + (relax && relax[k] == r ? 1.0 : 0.0)) * rs[r];
        int it = 0;
        const int st = lp_one(w, As, w->q, bs, tol, max_iter, &it);
        status[k] = st; iters[k] = it;
        double v = 0;
        for (int j = 0; j < n; j++) { x[(size_t)k * n + j] = w->x[j]; v += ck[j] * w->x[j]; }
        obj[k] = st == LP_OPTIMAL ? v : NAN;
        if (z) for (int r = 0; r < m; r++) z[(size_t)k * m + r] = w->z[r] * rs[r] * (st == LP_OPTIMAL ? cinf : 1.0);
        free(bs); lpw_free(w);
    }
    free(As); free(rs);
    return 0;
}
