/*
 * oracle/hsde_qp.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C float64 CPU restatement of the QP relaxation that the reference
 * solves at every branch-and-bound node, with the reference's sign, status
 * and infeasibility-certificate conventions.  Only tests/, smoke() and the
 * cpu_baseline leg of bench.py may call this file.
 *
 * What is restated (paths under /root/reference):
 *   - the QP itself:             warm_start_hmpc/controller.py:119-184
 *       min sum_t |Q x_t|^2 + |R u_t|^2 + |Q_T x_T|^2
 *       lam_0   : x_0 == x0
 *       nu_lb_t : -ub_t <= -lb_t          nu_ub_t : ub_t <= ub_t^max
 *       lam_t+1 : x_{t+1} == A x_t + B u_t
 *       mu_t    : F x_t + G u_t <= h      (t = T-1: F_Tm1, G_Tm1, h_Tm1)
 *   - node -> bounds:            controller.py:273-327 (fixed v => lb = ub = v)
 *   - status / sign conventions: warm_start_hmpc/bounded_qp.py:200-332
 *       optimal   : multipliers of "<=" rows are >= 0 (the reference returns -Pi)
 *       infeasible: objective +inf, primal absent, multipliers = Farkas proof,
 *                   dual objective = -sum(rhs * farkas) > 0
 *   - output record:             warm_start_hmpc/subproblem_solution.py:68-168
 *       rho_t = 2 Q x_t, rho_T = 2 Q_T x_T, sigma_t = 2 R u_t (zeros if infeasible)
 *
 * The arithmetic of the reference lives in Gurobi (gurobipy, un-pinned in
 * setup.py:16-20, proprietary, absent from this image), so it cannot be
 * restated; it is replaced by a published algorithm: a Mehrotra
 * predictor-corrector interior point method on the homogeneous embedding of
 * the QP (Andersen & Ye 1999; the QP form used by Goulart & Chen, "Clarabel",
 * 2024), with the KKT systems solved by a Riccati recursion over the horizon.
 * Optimal points and infeasibility proofs are self-certifying (KKT residuals
 * / Farkas conditions) and the tests check exactly those, so the oracle is
 * pinned by the reference's own known-answer and property tests
 * (SURVEY.md 8c); no Gurobi output exists to compare digits with.
 *
 * Inequality rows of stage t, in this order:  [F G] rows, then -ub_i <= -lb_i
 * (i < nub), then ub_i <= ub_i^max.  A fixed binary is not an inequality pair
 * here: it is the equality ub_i = v with one free multiplier nu = nu_ub-nu_lb,
 * split afterwards as nu_ub = max(nu,0), nu_lb = max(-nu,0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ST_OPTIMAL 0
#define ST_INFEASIBLE 1
#define ST_MAXITER 2
#define ST_NUMERICAL 3
/* iterative refinement pays only near the end, where D = z/s spans many orders of magnitude */
#define REFINE_MU 1e-3
#define REFINE_MU2 1e-7

typedef struct {
    int nx, nu, nub, nuc, nz, T, nc, ncL, mreg, mlast, M, nq, nr, nqT;
    const double *A, *B, *Q, *R, *QT;
    double *Creg, *hreg, *Clast, *hlast; /* row-scaled, bound rows included */
    double *sreg, *slast;                /* row scales of the general rows  */
    double *P, *PT;                      /* 2*cs*(Q'Q (+) R'R), 2*cs*QT'QT   */
    double cs;                           /* cost scale                       */
    int polish_l1;                       /* 1: curvature of the scaled cost of order one -- the polish starts and ends at its second penalty level (hmpc_device.h) */
    int *roff;                           /* first row of stage t            */
} prob_t;

static const double *Ct(const prob_t *p, int t) { return t < p->T - 1 ? p->Creg : p->Clast; }
static const double *ht(const prob_t *p, int t) { return t < p->T - 1 ? p->hreg : p->hlast; }
static int mt(const prob_t *p, int t) { return t < p->T - 1 ? p->mreg : p->mlast; }

static void build_stage(const prob_t *p, const double *F, const double *G, const double *h, int nrow,
                        double *C, double *hh, double *scale)
{
    int nx = p->nx, nu = p->nu, nz = p->nz, nub = p->nub, nuc = p->nuc;
    memset(C, 0, sizeof(double) * (size_t)(nrow + 2 * nub) * nz);
    for (int r = 0; r < nrow; r++) {
        double n2 = 0;
        for (int j = 0; j < nx; j++) n2 += F[r * nx + j] * F[r * nx + j];
        for (int j = 0; j < nu; j++) n2 += G[r * nu + j] * G[r * nu + j];
        double sc = n2 > 0 ? 1.0 / sqrt(n2) : 1.0;
        scale[r] = sc;
        for (int j = 0; j < nx; j++) C[r * nz + j] = sc * F[r * nx + j];
        for (int j = 0; j < nu; j++) C[r * nz + nx + j] = sc * G[r * nu + j];
        hh[r] = sc * h[r];
    }
    for (int i = 0; i < nub; i++) {
        C[(nrow + i) * nz + nx + nuc + i] = -1.0; hh[nrow + i] = 0.0;       /* -ub <= 0 */
        C[(nrow + nub + i) * nz + nx + nuc + i] = 1.0; hh[nrow + nub + i] = 1.0; /* ub <= 1 */
    }
}

static prob_t *prob_create(int nx, int nu, int nub, int T, int nc, int ncL, int nq, int nr, int nqT,
                           const double *A, const double *B, const double *F, const double *G, const double *h,
                           const double *FL, const double *GL, const double *hL,
                           const double *Q, const double *R, const double *QT)
{
    prob_t *p = (prob_t *)calloc(1, sizeof(prob_t));
    p->nx = nx; p->nu = nu; p->nub = nub; p->nuc = nu - nub; p->nz = nx + nu; p->T = T;
    p->nc = nc; p->ncL = ncL; p->mreg = nc + 2 * nub; p->mlast = ncL + 2 * nub;
    p->M = (T - 1) * p->mreg + p->mlast; p->nq = nq; p->nr = nr; p->nqT = nqT;
    p->A = A; p->B = B; p->Q = Q; p->R = R; p->QT = QT;
    int nz = p->nz;
    p->Creg = (double *)malloc(sizeof(double) * p->mreg * nz); p->hreg = (double *)malloc(sizeof(double) * p->mreg);
    p->Clast = (double *)malloc(sizeof(double) * p->mlast * nz); p->hlast = (double *)malloc(sizeof(double) * p->mlast);
    p->sreg = (double *)malloc(sizeof(double) * (nc + 1)); p->slast = (double *)malloc(sizeof(double) * (ncL + 1));
    build_stage(p, F, G, h, nc, p->Creg, p->hreg, p->sreg);
    build_stage(p, FL, GL, hL, ncL, p->Clast, p->hlast, p->slast);
    p->P = (double *)calloc((size_t)nz * nz, sizeof(double)); p->PT = (double *)calloc((size_t)nx * nx, sizeof(double));
    /* cost scale: bring the largest Hessian entry to 1 */
    double big = 0;
    for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) {
        double a = 0, b = 0;
        for (int k = 0; k < nq; k++) a += Q[k * nx + i] * Q[k * nx + j];
        for (int k = 0; k < nqT; k++) b += QT[k * nx + i] * QT[k * nx + j];
        p->P[i * nz + j] = 2 * a; p->PT[i * nx + j] = 2 * b;
        if (fabs(2 * a) > big) big = fabs(2 * a);
        if (fabs(2 * b) > big) big = fabs(2 * b);
    }
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) {
        double a = 0;
        for (int k = 0; k < nr; k++) a += R[k * nu + i] * R[k * nu + j];
        p->P[(nx + i) * nz + nx + j] = 2 * a;
        if (fabs(2 * a) > big) big = fabs(2 * a);
    }
    p->cs = big > 0 ? 1.0 / big : 1.0;
    for (int i = 0; i < nz * nz; i++) p->P[i] *= p->cs;
    for (int i = 0; i < nx * nx; i++) p->PT[i] *= p->cs;
    {   /* curvature of the scaled cost: smallest positive diagonal entry (as hmpc_create) */
        double cmin = 1.0;
        for (int i = 0; i < nz; i++) if (p->P[i * nz + i] > 0 && p->P[i * nz + i] < cmin) cmin = p->P[i * nz + i];
        for (int i = 0; i < nx; i++) if (p->PT[i * nx + i] > 0 && p->PT[i * nx + i] < cmin) cmin = p->PT[i * nx + i];
        p->polish_l1 = cmin >= 1e-2 ? 1 : 0;
    }
    p->roff = (int *)malloc(sizeof(int) * (T + 1));
    for (int t = 0; t <= T; t++) p->roff[t] = t < T ? t * p->mreg : p->M;
    return p;
}

static void prob_free(prob_t *p)
{
    free(p->Creg); free(p->hreg); free(p->Clast); free(p->hlast); free(p->sreg); free(p->slast);
    free(p->P); free(p->PT); free(p->roff); free(p);
}

/* per-solve workspace */
typedef struct {
    double *w, *lam, *nuf, *s, *z;           /* iterate: w = [w_0..w_{T-1} | x_T]              */
    double *D, *Phi, *L, *Kx, *Pr, *mb;      /* factorisation                                  */
    double *rd, *rdyn, *rc;                  /* residuals                                      */
    double *g, *rhs_c, *pvec, *lu;           /* solve scratch                                  */
    double *w1, *lam1, *nuf1, *z1;           /* constant direction                             */
    double *w2, *lam2, *nuf2, *z2;           /* residual direction (affine, then corrector)    */
    double *dsa, *dza;                       /* affine slack / multiplier steps                */
    double *Pw, *sd, *sc, *ed, *edyn, *ec;
    unsigned char *act;
} work_t;

static work_t *work_create(const prob_t *p)
{
    work_t *k = (work_t *)calloc(1, sizeof(work_t));
    int nz = p->nz, nx = p->nx, nu = p->nu, T = p->T, M = p->M, n = T * nz + nx;
#define AL(name, cnt) k->name = (double *)calloc((size_t)(cnt), sizeof(double))
    AL(w, n); AL(lam, (T + 1) * nx); AL(nuf, T * p->nub); AL(s, M); AL(z, M);
    AL(D, M); AL(Phi, (size_t)T * nz * nz); AL(L, (size_t)T * nu * nu); AL(Kx, (size_t)T * nu * nx);
    AL(Pr, (size_t)(T + 1) * nx * nx); AL(mb, (size_t)T * nz);
    AL(rd, n); AL(rdyn, T * nx); AL(rc, M); AL(g, n); AL(rhs_c, M); AL(pvec, (T + 1) * nx); AL(lu, T * nu);
    AL(w1, n); AL(lam1, (T + 1) * nx); AL(nuf1, T * p->nub); AL(z1, M);
    AL(w2, n); AL(lam2, (T + 1) * nx); AL(nuf2, T * p->nub); AL(z2, M);
    AL(dsa, M); AL(dza, M); AL(Pw, n); AL(sd, n); AL(sc, T * nx); AL(ed, n); AL(edyn, T * nx); AL(ec, M);
#undef AL
    k->act = (unsigned char *)calloc((size_t)M, 1);
    return k;
}

static void work_free(work_t *k)
{
    free(k->w); free(k->lam); free(k->nuf); free(k->s); free(k->z); free(k->D); free(k->Phi); free(k->L);
    free(k->Kx); free(k->Pr); free(k->mb); free(k->rd); free(k->rdyn); free(k->rc); free(k->g); free(k->rhs_c);
    free(k->pvec); free(k->lu); free(k->w1); free(k->lam1); free(k->nuf1); free(k->z1); free(k->w2);
    free(k->lam2); free(k->nuf2); free(k->z2); free(k->dsa); free(k->dza); free(k->Pw); free(k->sd); free(k->sc); free(k->ed); free(k->edyn); free(k->ec); free(k->act); free(k);
}

/* Riccati factorisation of  Phi_t = P + C_t' D C_t  along the horizon.
 * fix[t*nub+i] in {-1 free, 0, 1}; a fixed binary is a prescribed variable. */
static int factor(const prob_t *p, work_t *k, const int8_t *fix)
{
    int nx = p->nx, nu = p->nu, nz = p->nz, T = p->T, nuc = p->nuc, nub = p->nub;
    double Mm[64 * 64], PA[32 * 64];
    memcpy(k->Pr + (size_t)T * nx * nx, p->PT, sizeof(double) * nx * nx);
    for (int t = T - 1; t >= 0; t--) {
        const double *C = Ct(p, t); int m = mt(p, t); const double *D = k->D + p->roff[t];
        const double *Pn = k->Pr + (size_t)(t + 1) * nx * nx;
        /* M = P + C' D C */
        for (int i = 0; i < nz; i++) for (int j = 0; j <= i; j++) {
            double a = p->P[i * nz + j];
            for (int r = 0; r < m; r++) { double ci = C[r * nz + i]; if (ci != 0.0) a += ci * D[r] * C[r * nz + j]; }
            Mm[i * nz + j] = a; Mm[j * nz + i] = a;
        }
        /* PA = Pn [A B] (nx x nz) ;  M += [A B]' PA */
        for (int i = 0; i < nx; i++) for (int j = 0; j < nz; j++) {
            double a = 0;
            for (int l = 0; l < nx; l++) a += Pn[i * nx + l] * (j < nx ? p->A[l * nx + j] : p->B[l * nu + (j - nx)]);
            PA[i * nz + j] = a;
        }
        for (int i = 0; i < nz; i++) for (int j = 0; j <= i; j++) {
            double a = 0;
            for (int l = 0; l < nx; l++) a += (i < nx ? p->A[l * nx + i] : p->B[l * nu + (i - nx)]) * PA[l * nz + j];
            Mm[i * nz + j] += a; if (i != j) Mm[j * nz + i] += a;
        }
        /* prescribed (fixed) binaries */
        double *mb = k->mb + (size_t)t * nz;
        for (int i = 0; i < nz; i++) mb[i] = 0;
        for (int b = 0; b < nub; b++) if (fix[t * nub + b] == 1) {
            int c = nx + nuc + b;
            for (int i = 0; i < nz; i++) mb[i] += Mm[i * nz + c];
        }
        for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) {
            int c = nx + nuc + b;
            for (int i = 0; i < nz; i++) { Mm[i * nz + c] = 0; Mm[c * nz + i] = 0; }
            Mm[c * nz + c] = 1.0;
        }
        /* Cholesky of M_uu */
        double *L = k->L + (size_t)t * nu * nu, *Kx = k->Kx + (size_t)t * nu * nx;
        for (int j = 0; j < nu; j++) {
            double d = Mm[(nx + j) * nz + nx + j];
            for (int l = 0; l < j; l++) d -= L[j * nu + l] * L[j * nu + l];
            if (!(d > 0)) return -1;
            d = sqrt(d); L[j * nu + j] = d;
            for (int i = j + 1; i < nu; i++) {
                double a = Mm[(nx + i) * nz + nx + j];
                for (int l = 0; l < j; l++) a -= L[i * nu + l] * L[j * nu + l];
                L[i * nu + j] = a / d;
            }
        }
        /* Kx = L^{-1} M_ux */
        for (int c = 0; c < nx; c++) for (int i = 0; i < nu; i++) {
            double a = Mm[(nx + i) * nz + c];
            for (int l = 0; l < i; l++) a -= L[i * nu + l] * Kx[l * nx + c];
            Kx[i * nx + c] = a / L[i * nu + i];
        }
        double *Pt = k->Pr + (size_t)t * nx * nx;
        for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) {
            double a = Mm[i * nz + j];
            for (int l = 0; l < nu; l++) a -= Kx[l * nx + i] * Kx[l * nx + j];
            Pt[i * nx + j] = a;
        }
    }
    return 0;
}

/* Solves K v = rhs for one right-hand side.
 *   rhs_d : n (stage gradients), may be NULL (= 0)
 *   a     : prescribed x_0 (nx), may be NULL
 *   cdyn  : T*nx dynamics offsets, may be NULL
 *   useb  : prescribed value of fixed binary = its v (1) or 0 (0)
 *   rhs_c : M
 * outputs dw (n), dlam ((T+1)nx), dnuf (T nub), dz (M)                          */
static void kkt_solve(const prob_t *p, work_t *k, const int8_t *fix, const double *rhs_d, const double *a,
                      const double *cdyn, int useb, const double *rhs_c,
                      double *dw, double *dlam, double *dnuf, double *dz)
{
    int nx = p->nx, nu = p->nu, nz = p->nz, T = p->T, nuc = p->nuc, nub = p->nub;
    double *g = k->g, *pv = k->pvec;
    double q[64], mv[128];
    /* g = rhs_d + C' (D rhs_c) */
    for (int t = 0; t < T; t++) {
        const double *C = Ct(p, t); int m = mt(p, t); int ro = p->roff[t];
        for (int j = 0; j < nz; j++) g[t * nz + j] = rhs_d ? rhs_d[t * nz + j] : 0.0;
        for (int r = 0; r < m; r++) if (k->act[ro + r]) {
            double e = k->D[ro + r] * rhs_c[ro + r];
            for (int j = 0; j < nz; j++) g[t * nz + j] += C[r * nz + j] * e;
        }
    }
    for (int j = 0; j < nx; j++) { g[T * nz + j] = rhs_d ? rhs_d[T * nz + j] : 0.0; pv[T * nx + j] = -g[T * nz + j]; }
    /* backward */
    for (int t = T - 1; t >= 0; t--) {
        const double *Pn = k->Pr + (size_t)(t + 1) * nx * nx;
        const double *L = k->L + (size_t)t * nu * nu, *Kx = k->Kx + (size_t)t * nu * nx;
        for (int i = 0; i < nx; i++) {
            double s = pv[(t + 1) * nx + i];
            if (cdyn) for (int l = 0; l < nx; l++) s += Pn[i * nx + l] * cdyn[t * nx + l];
            q[i] = s;
        }
        for (int j = 0; j < nz; j++) {
            double s = -g[t * nz + j];
            for (int l = 0; l < nx; l++) s += (j < nx ? p->A[l * nx + j] : p->B[l * nu + (j - nx)]) * q[l];
            if (useb) s += k->mb[t * nz + j];
            mv[j] = s;
        }
        for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0)
            mv[nx + nuc + b] = (useb && fix[t * nub + b] == 1) ? -1.0 : 0.0;
        double *lu = k->lu + (size_t)t * nu;
        for (int i = 0; i < nu; i++) {
            double s = mv[nx + i];
            for (int l = 0; l < i; l++) s -= L[i * nu + l] * lu[l];
            lu[i] = s / L[i * nu + i];
        }
        for (int i = 0; i < nx; i++) {
            double s = mv[i];
            for (int l = 0; l < nu; l++) s -= Kx[l * nx + i] * lu[l];
            pv[t * nx + i] = s;
        }
    }
    /* forward */
    for (int i = 0; i < nx; i++) dw[i] = a ? a[i] : 0.0;
    for (int t = 0; t < T; t++) {
        const double *L = k->L + (size_t)t * nu * nu, *Kx = k->Kx + (size_t)t * nu * nx;
        const double *lu = k->lu + (size_t)t * nu;
        double *x = dw + t * nz, *u = dw + t * nz + nx, *xn = dw + (t + 1) * nz;
        double v[64];
        for (int i = 0; i < nu; i++) {
            double s = lu[i];
            for (int l = 0; l < nx; l++) s += Kx[i * nx + l] * x[l];
            v[i] = -s;
        }
        for (int i = nu - 1; i >= 0; i--) {
            double s = v[i];
            for (int l = i + 1; l < nu; l++) s -= L[l * nu + i] * u[l];
            u[i] = s / L[i * nu + i];
        }
        for (int i = 0; i < nx; i++) {
            double s = cdyn ? cdyn[t * nx + i] : 0.0;
            for (int l = 0; l < nx; l++) s += p->A[i * nx + l] * x[l];
            for (int l = 0; l < nu; l++) s += p->B[i * nu + l] * u[l];
            xn[i] = s;
        }
    }
    /* multipliers of the equalities: lam_t = -(P_t x_t + p_t) */
    for (int t = 0; t <= T; t++) {
        const double *Pt = k->Pr + (size_t)t * nx * nx; const double *x = dw + t * nz;
        for (int i = 0; i < nx; i++) {
            double s = pv[t * nx + i];
            for (int l = 0; l < nx; l++) s += Pt[i * nx + l] * x[l];
            dlam[t * nx + i] = -s;
        }
    }
    /* dz = D (C dw - rhs_c) */
    for (int t = 0; t < T; t++) {
        const double *C = Ct(p, t); int m = mt(p, t); int ro = p->roff[t];
        for (int r = 0; r < m; r++) {
            if (!k->act[ro + r]) { dz[ro + r] = 0; continue; }
            double s = -rhs_c[ro + r];
            for (int j = 0; j < nz; j++) s += C[r * nz + j] * dw[t * nz + j];
            dz[ro + r] = k->D[ro + r] * s;
        }
    }
    /* lam_0 from the stationarity row of x_0 (x_0 is prescribed: the row defines its multiplier, as the rows of the fixed
     * binaries define theirs below).  Through the recursion -- lam_0 = -(P_0 x_0 + p_0) -- it carries the rounding of the
     * cost-to-go of stage 0, eps |P_0| with |P_0| ~ max D: on deep nodes of BASELINE configs[4] that row of the dual
     * residual stood at 2e-6 in the last iterations (every other row: 1e-15) and the refinement, which takes the
     * prescribed components as met, never saw it (round 4). */
    {
        const double *C = Ct(p, 0); int m = mt(p, 0); int ro = p->roff[0];
        for (int i = 0; i < nx; i++) {
            double s = rhs_d ? rhs_d[i] : 0.0;
            for (int j = 0; j < nz; j++) s -= p->P[i * nz + j] * dw[j];
            for (int r = 0; r < m; r++) s -= C[r * nz + i] * dz[ro + r];
            for (int l = 0; l < nx; l++) s += p->A[l * nx + i] * dlam[nx + l];
            dlam[i] = s;
        }
    }
    /* multipliers of the fixed binaries from stationarity of their component */
    for (int t = 0; t < T; t++) {
        const double *C = Ct(p, t); int m = mt(p, t); int ro = p->roff[t];
        for (int b = 0; b < nub; b++) {
            if (fix[t * nub + b] < 0) { dnuf[t * nub + b] = 0; continue; }
            int c = nx + nuc + b;
            double s = rhs_d ? rhs_d[t * nz + c] : 0.0;
            for (int j = 0; j < nz; j++) s -= p->P[c * nz + j] * dw[t * nz + j];
            for (int r = 0; r < m; r++) s -= C[r * nz + c] * dz[ro + r];
            for (int l = 0; l < nx; l++) s += p->B[l * nu + (c - nx)] * dlam[(t + 1) * nx + l];
            dnuf[t * nub + b] = s;
        }
    }
}

/* Residual of the linear blocks  K d = (rhs_d ; rhs_e ; rhs_c) + dtau (0 ; f ; h)  at a computed d.
 * The prescribed components (x_0, fixed binaries) are met exactly by construction. */
static void kkt_residual(const prob_t *p, const work_t *k, const int8_t *fix, const double *rhs_d, const double *cdyn,
                         const double *rhs_c, double dtau, const double *dw, const double *dlam, const double *dnuf,
                         const double *dz, double *ed, double *edyn, double *ec)
{
    int nx = p->nx, nu = p->nu, nz = p->nz, T = p->T, nuc = p->nuc, nub = p->nub;
    for (int t = 0; t < T; t++) {
        const double *C = Ct(p, t); const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
        double *e = ed + t * nz;
        for (int i = 0; i < nz; i++) {
            double a = rhs_d[t * nz + i];
            for (int j = 0; j < nz; j++) a -= p->P[i * nz + j] * dw[t * nz + j];
            e[i] = a;
        }
        for (int j = 0; j < nx; j++) {
            double a = dlam[t * nx + j];
            for (int l = 0; l < nx; l++) a -= p->A[l * nx + j] * dlam[(t + 1) * nx + l];
            e[j] -= a;
        }
        for (int j = 0; j < nu; j++) {
            double a = 0;
            for (int l = 0; l < nx; l++) a -= p->B[l * nu + j] * dlam[(t + 1) * nx + l];
            e[nx + j] -= a;
        }
        for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) e[nx + nuc + b] -= dnuf[t * nub + b];
        for (int r = 0; r < m; r++) {
            if (!k->act[ro + r]) { ec[ro + r] = 0; continue; }
            double a = rhs_c[ro + r] + dtau * hh[r] + dz[ro + r] * k->s[ro + r] / k->z[ro + r];
            for (int j = 0; j < nz; j++) { double c = C[r * nz + j]; if (c != 0.0) { e[j] -= c * dz[ro + r]; a -= c * dw[t * nz + j]; } }
            ec[ro + r] = a;
        }
        /* x_0 and fixed binaries are prescribed: their stationarity rows define dlam_0 / dnuf */
        for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) e[nx + nuc + b] = 0;
        for (int i = 0; i < nx; i++) {
            double a = cdyn[t * nx + i] - dw[(t + 1) * nz + i];
            for (int l = 0; l < nx; l++) a += p->A[i * nx + l] * dw[t * nz + l];
            for (int l = 0; l < nu; l++) a += p->B[i * nu + l] * dw[t * nz + nx + l];
            edyn[t * nx + i] = a;
        }
    }
    for (int j = 0; j < nx; j++) {
        double a = rhs_d[T * nz + j] - dlam[T * nx + j];
        for (int l = 0; l < nx; l++) a -= p->PT[j * nx + l] * dw[T * nz + l];
        ed[T * nz + j] = a;
    }
    for (int j = 0; j < nx; j++) ed[j] = 0;   /* x_0 row defines dlam_0 */
}

static double vmaxabs(const double *v, int n) { double m = 0; for (int i = 0; i < n; i++) if (fabs(v[i]) > m) m = fabs(v[i]); return m; }


/* ---- Active-set polish ---------------------------------------------------------------------------
 * An interior-point iterate that meets the stopping test still carries, on nodes without an interior
 * (big-M rows that collapse to equalities), an error of (dual residual) / (curvature of the stage cost)
 * ~ 1e-5 in the trajectory.  What a simplex / crossover code -- the reference's Gurobi -- returns is the
 * vertex solution of the optimal active set; this step computes that point from the iterate.
 *
 * Rows with z > s (or whose slack shrinks faster than their multiplier) are taken as active.  The equality-constrained QP on them is solved by the method of
 * multipliers with the same Riccati machinery (one factorisation, a few solves):
 *     w+ = argmin 1/2 w'Pw + z'(C_A w - h_A) + rho/2 |C_A w - h_A|^2 + delta/2 |C_I (w - w0)|^2   s.t. dynamics, x_0, fixed binaries
 *     z+ = z + rho (C_A w+ - h_A)
 * i.e. one KKT solve with D = rho on the active rows, D = delta on the inactive ones (a proximal term in
 * the metric of the inactive rows: it makes the step unique where the cost sees no input, and leaves the
 * iterate's component there untouched), right-hand side rhs_c = h - z/rho (active), C w0 (inactive).
 * Dependent active rows (the pairs f <= 0, -f <= 0) are harmless: the multipliers converge on the range of C_A.
 * The result is exactly complementary (s_A = 0, z_I = 0) and exactly stationary; it is accepted only if
 * z_A >= 0 and C_I w <= h_I hold to the tolerances below and the active rows are met to 1e-11 -- otherwise
 * the interior-point iteration simply goes on (nothing of its state is touched).
 * Outputs (tau = 1 units): k->w1, lam1, nuf1 and the row multipliers in k->dza.
 */
#ifndef REFINE_FROM_IT
#define REFINE_FROM_IT 12
#endif
#define POLISH_RHO 1e5      /* penalty of the active rows (scaled problem: unit rows, largest Hessian entry 1) */
#define POLISH_RHO2 1e7     /* second level, for active sets whose multiplier steps do not settle at the first        */
#define POLISH_DELTA 1e-10 /* proximal weight of the inactive rows; must stay above eps * rho            */
#define POLISH_ITERS 5      /* multiplier steps per active set                                            */
#define POLISH_ROUNDS 6        /* active sets tried per attempt ...                                        */
#define POLISH_ROUNDS_LAST 10  /* ... and by the last attempt, on the iterate the solve would return (round 3, with the rule that every row with a negative multiplier leaves) */
#ifndef POLISH_ROUNDS_WARM
#define POLISH_ROUNDS_WARM 3     /* active sets tried when the set is handed down by the parent node */
#endif
#define POLISH_WARM_BMOVE 0.5 /* a hand-down is not tried where a fixed binary lies further than this from the parent's value */
#ifndef POLISH_WARM_VMAX
#define POLISH_WARM_VMAX 1e-2
#endif
#define ESC_MAX 2           /* tolerance escalations after a failed last attempt (solve_one) ... */
#define ESC_ITERS 6         /* ... and iterations an escalated solve may spend without meeting its tolerance */
#define POLISH_ATTEMPTS 3   /* attempts per solve (a node whose active set resists is left to the interior-point iterate) */
/* zwarm (optional, M entries, solver units, tau = 1): multipliers handed down by the parent node -- the active set is
 * read from them (z > 0: the parent's record is a polished vertex, exactly complementary) instead of from the iterate,
 * and k->w holds the parent's primal point with the child's prescribed components written over it. */
static int polish(const prob_t *p, work_t *k, const double *x0, const int8_t *fix, double tau, double winf, double zinf, double last_alpha, const double *zwarm, int last)
{
    int nz = p->nz, T = p->T;
    double rho = p->polish_l1 ? POLISH_RHO2 : POLISH_RHO;
    int level = p->polish_l1; /* 0: first penalty level; 1: second; 2: back at the first for the last digits */
    double *zk = k->dza, *cw = k->dsa, *cw0 = k->ec;
    for (int t = 0; t < T; t++) {
        const double *C = Ct(p, t); int m = mt(p, t), ro = p->roff[t];
        for (int r = 0; r < m; r++) {
            int q = ro + r;
            if (!k->act[q]) { k->D[q] = 0; zk[q] = 0; cw0[q] = 0; continue; }
            double a = 0;
            for (int j = 0; j < nz; j++) a += C[r * nz + j] * k->w[t * nz + j];
            cw0[q] = a / tau;
            /* active: z > s, or -- Tapia indicators over the last step (dz in z2, ds in rhs_c) -- the slack shrinks
             * faster than the multiplier: s+/s < z+/z.  The second test reads weakly active rows (both small) right
             * far more often than the first alone: 2.2 instead of 3.3 active sets per polish, no failures on the cart-pole. */
            int active = zwarm ? zwarm[q] > 0 : k->z[q] > k->s[q];
            if (!zwarm && last_alpha > 0) {
                const double sp = k->s[q] - last_alpha * k->rhs_c[q], zp = k->z[q] - last_alpha * k->z2[q];
                if (sp > 0 && zp > 0 && k->z[q] * sp > k->s[q] * zp) active = 1;
            }
            if (active) { k->D[q] = rho; zk[q] = (zwarm ? zwarm[q] : k->z[q]) / tau; }
            else { k->D[q] = POLISH_DELTA; zk[q] = 0; }
        }
    }
    /* A violated inactive row costs its violation in x; a negative multiplier that is clipped costs |z| / (curvature
     * of the stage cost) -- 1e-9 / 1.4e-4 would already be 7e-6 in the trajectory.  So the sign test is absolute and
     * tight (scaled problem: unit rows, largest Hessian entry 1): a row with z < -1e-11 changes sides, and ends inactive
     * with a violation below es if it was active with a zero multiplier. */
    const double ez = 1e-11, es = 1e-9 * (1 + winf);
    (void)zinf;
    /* (a hand-down that does not verify on the parent's set or after one exchange of rows is dropped: the child's
     * optimum is elsewhere -- typically the child is infeasible -- and every further round costs an iteration's worth) */
    /* (last: the attempt on the iterate the solve would return -- the one that gets the longer sequence of active sets) */
    const int max_rounds = zwarm ? POLISH_ROUNDS_WARM : last ? POLISH_ROUNDS_LAST : POLISH_ROUNDS;
    for (int round = 0; round < max_rounds; round++) {
        if (factor(p, k, fix) != 0) { return 0; }
        memcpy(cw, cw0, sizeof(double) * p->M); /* the proximal centre starts at the interior-point iterate */
        double pinf = 0, pmove = 0;
        for (int it = 0; it < (level == 1 ? 2 * POLISH_ITERS : POLISH_ITERS); it++) { /* (the second level gets twice the steps: it is there for the slow sets) */
            for (int t = 0; t < T; t++) {
                const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
                for (int r = 0; r < m; r++) { int q = ro + r; k->rhs_c[q] = k->D[q] >= 1.0 ? hh[r] - zk[q] / rho : cw[q]; }
            }
            kkt_solve(p, k, fix, NULL, x0, NULL, 1, k->rhs_c, k->w1, k->lam1, k->nuf1, k->z1);
            pinf = 0; pmove = 0;
            for (int t = 0; t < T; t++) {
                const double *C = Ct(p, t); int m = mt(p, t), ro = p->roff[t];
                for (int r = 0; r < m; r++) {
                    int q = ro + r;
                    if (k->D[q] >= 1.0) {
                        double d = fabs(k->z1[q] - zk[q]) / rho;
                        if (d > pinf) pinf = d;
                        zk[q] = k->z1[q];
                    } else if (k->act[q]) { /* the proximal centre follows the iterate */
                        double a = 0;
                        for (int j = 0; j < nz; j++) a += C[r * nz + j] * k->w1[t * nz + j];
                        if (fabs(a - cw[q]) > pmove) pmove = fabs(a - cw[q]);
                        cw[q] = a;
                    }
                }
            }
            if (it >= 1 && pinf <= 1e-12 * (1 + winf) && POLISH_DELTA * pmove <= 1e-13) break;
        }
        /* the active rows must be met -- to 1e-12 at the first level (the residual enters the cost with the
         * multipliers, ~1e3: a set that does not get there in POLISH_ITERS steps goes to the second level), to 1e-10
         * beyond --, and the proximal term (dropped from the multipliers) must have died out: it is the stationarity
         * residual of the result */
        if (getenv("ORACLE_QP_TRACE")) fprintf(stderr, "      polish round %d pinf %.3e pmove %.3e\n", round, pinf, pmove);
        if (!(pinf <= (level == 0 ? 1e-12 : 1e-10) * (1 + winf)) || !(POLISH_DELTA * pmove <= 1e-12 * (1 + zinf))) {
            /* The multiplier steps contract by 1 / (1 + rho lambda), lambda the eigenvalues of C_A Phi^-1 C_A': active
             * rows that nearly depend on each other (a terminal-set facet next to the state bound it was pushed
             * through the dynamics from) do not settle at rho = 1e5.  They do at 1e7 -- not the first choice: its
             * rounding (eps rho = 1e-9 in the multipliers, over the curvature of the stage cost ~1e-6 in the
             * trajectory) is no longer below the proximal weight -- so the second level is used only where the first
             * fails, once per attempt, and its result goes through the first level once more (level 2 below). */
            if (level != 0 || !(pinf == pinf)) return 0;
            level = 1; rho = POLISH_RHO2;
            for (int q = 0; q < p->M; q++) if (k->D[q] >= 1.0) k->D[q] = rho;
            continue;
        }
        /* Sign of the multipliers, slack of the inactive rows.  Rows on the wrong side change sides: of the violated
         * inactive rows only those within a factor two of the worst violation (a missing active row drags others across
         * their bounds; the next round shows which of them are real); of the active rows EVERY one with a negative
         * multiplier leaves.  (Until round 3 the multipliers followed the factor-two rule as well: on relaxations with
         * many weakly active rows -- the random MLD of BASELINE configs[4], binaries outside the cost -- the most negative
         * multiplier then only halved from round to round and 28 % of the optimal nodes never verified; with this rule and
         * ten rounds in the last attempt 3 % do not, in fewer factorisations.  Nothing changes on the cart-pole systems: same active sets,
         * same number of rounds.) */
        double vmax = 0, zmin = 0;
        for (int t = 0; t < T; t++) {
            const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
            for (int r = 0; r < m; r++) {
                int q = ro + r;
                if (!k->act[q]) continue;
                if (k->D[q] >= 1.0) { if (zk[q] < zmin) zmin = zk[q]; }
                else if (cw[q] - hh[r] > vmax) vmax = cw[q] - hh[r];
            }
        }
        if (getenv("ORACLE_QP_TRACE")) fprintf(stderr, "      polish round %d vmax %.3e zmin %.3e\n", round, vmax, zmin);
        /* a handed-down set whose point misses an inactive row by this much is not near the child's optimum (the child is
         * infeasible, or fixing the binary moved the solution): drop it after this one factorisation */
        if (zwarm && vmax > POLISH_WARM_VMAX * (1 + winf)) return 0;
        if (vmax <= es && zmin >= -ez) {
            if (level == 1 && !p->polish_l1 && round + 1 < max_rounds) {
                /* verified at the second level: the same active set once more at the first, from these multipliers --
                 * what is left of them to settle are the components that matter (C_A' dz of the size of the rounding of
                 * the second level); the ones that were slow are already in place */
                level = 2; rho = POLISH_RHO;
                for (int q = 0; q < p->M; q++) if (k->D[q] >= 1.0) k->D[q] = rho;
                continue;
            }
            for (int q = 0; q < p->M; q++) if (k->D[q] >= 1.0 && zk[q] < 0) zk[q] = 0;
            return 1;
        }
        for (int t = 0; t < T; t++) {
            const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
            for (int r = 0; r < m; r++) {
                int q = ro + r;
                if (!k->act[q]) continue;
                if (k->D[q] < 1.0) { if (vmax > es && cw[q] - hh[r] > 0.5 * vmax) { k->D[q] = rho; zk[q] = 0; } }
                else if (zmin < -ez && zk[q] < 0.0) { k->D[q] = POLISH_DELTA; zk[q] = 0; }
            }
        }
    }
    return 0;
}

/* One QP.  Outputs are in the ORIGINAL (unscaled) problem. */
static int solve_one(const prob_t *p, work_t *k, const double *x0, const int8_t *fix, int term_on, int refine, int do_polish, double ptol, double tol, double tol_inf, int max_iter,
                     const double *wprimal, const double *wdual, int attempt_only,
                     double *obj, double *dobj, int *iters, double *primal, double *dual, double *term_viol, int *polished_out)
{
    int nx = p->nx, nu = p->nu, nz = p->nz, T = p->T, nuc = p->nuc, nub = p->nub, M = p->M, n = T * nz + nx;
    int mact = 0;
    for (int t = 0; t < T; t++) {
        int m = mt(p, t), mg = m - 2 * nub, ro = p->roff[t];
        for (int r = 0; r < m; r++) {
            int on = 1;
            if (r >= mg) on = fix[t * nub + ((r - mg) % nub)] < 0;
            else if (!term_on && t == T - 1 && r >= p->nc) on = 0;   /* terminal-set rows masked */
            k->act[ro + r] = (unsigned char)on; mact += on;
            k->s[ro + r] = 1.0; k->z[ro + r] = on ? 1.0 : 0.0;
        }
    }
    memset(k->w, 0, sizeof(double) * n); memset(k->lam, 0, sizeof(double) * (T + 1) * nx);
    memset(k->nuf, 0, sizeof(double) * T * nub);
    double tau = 1.0, kap = 1.0;
    /* prescribed components follow tau exactly */
    for (int i = 0; i < nx; i++) k->w[i] = x0[i] * tau;
    for (int t = 0; t < T; t++) for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) k->w[t * nz + nx + nuc + b] = fix[t * nub + b] * tau;

    int status = ST_MAXITER, it = 0, extra_done = 0, polished = 0, npol = 0, weak = 0, tried_it = -1, leave = 0;
    double last_alpha = 0, last_dtau = 0, last_dkap = 0;
    /* ---- Parent -> child hand-down (the reference hands the parent's simplex basis to the child, controller.py:260-264,
     * subproblem_solution.py:37-43).  wprimal / wdual: the PARENT's record (output conventions).  Its active set -- the rows
     * with a positive multiplier, minus the bound rows of binaries the child fixes -- is tried by the polish before the
     * first interior-point iteration, multipliers and proximal centre from the parent; a child whose optimum lies on the
     * same active set (the branch that fixes a binary where the relaxation had it) verifies after one factorisation and a
     * few solves.  Otherwise nothing has happened: the cold start below is untouched. */
    if (wprimal && wdual && do_polish) {
        double *zw = k->z2;
        const double *mu_w = wdual + (T + 1) * nx; const int nmu_w = (T - 1) * p->nc + p->ncL;
        const double *nlb_w = mu_w + nmu_w, *nub_w = nlb_w + T * nub;
        double winf0 = 0, zinf0 = 0;
        for (int t = 0; t <= T; t++) for (int i = 0; i < nx; i++) k->w[t * nz + i] = wprimal[t * nx + i];
        for (int t = 0; t < T; t++) for (int i = 0; i < nu; i++) k->w[t * nz + nx + i] = wprimal[(T + 1) * nx + t * nu + i];
        for (int i = 0; i < nx; i++) k->w[i] = x0[i];
        /* a binary this node fixes far from where the parent's relaxation had it (the 1-branch of a binary relaxed to 0:
         * most infeasible children): the parent's set is not near this node's optimum, nothing is tried */
        double bmove = 0;
        for (int t = 0; t < T; t++) for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) {
            const double d = fabs(k->w[t * nz + nx + nuc + b] - fix[t * nub + b]);
            if (d > bmove) bmove = d;
            k->w[t * nz + nx + nuc + b] = fix[t * nub + b];
        }
        for (int i = 0; i < n; i++) if (fabs(k->w[i]) > winf0) winf0 = fabs(k->w[i]);
        for (int t = 0; t < T; t++) {
            int m = mt(p, t), mg = m - 2 * nub, ro = p->roff[t]; const double *sc = t < T - 1 ? p->sreg : p->slast;
            for (int r = 0; r < m; r++) {
                double v = 0;
                if (k->act[ro + r]) {
                    if (r < mg) v = mu_w[t * p->nc + r] * p->cs / sc[r];
                    else if (r < mg + nub) v = nlb_w[t * nub + (r - mg)] * p->cs;
                    else v = nub_w[t * nub + (r - mg - nub)] * p->cs;
                }
                zw[ro + r] = v; if (v > zinf0) zinf0 = v;
            }
        }
        for (int i = 0; i < (T + 1) * nx; i++) if (fabs(wdual[i]) * p->cs > zinf0) zinf0 = fabs(wdual[i]) * p->cs;
        if (bmove <= POLISH_WARM_BMOVE && winf0 == winf0 && zinf0 == zinf0 && polish(p, k, x0, fix, 1.0, winf0, zinf0, 0.0, zw, 0)) { status = ST_OPTIMAL; polished = 64; /* (marks a handed-down set that verified) */ it = 0; tau = 1.0; goto output; }
        if (attempt_only) return -1; /* (the caller goes on with the regular sequence of solves) */
        memset(k->w, 0, sizeof(double) * n);
        for (int i = 0; i < nx; i++) k->w[i] = x0[i] * tau;
        for (int t = 0; t < T; t++) for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) k->w[t * nz + nx + nuc + b] = fix[t * nub + b] * tau;
    }
    double hinf = fmax(vmaxabs(p->hreg, p->mreg), vmaxabs(p->hlast, p->mlast));
    double x0inf = vmaxabs(x0, nx);
    (void)hinf;
    for (it = 0; it <= max_iter; it++) {
        /* ---- residuals ---- */
        double wPw = 0, fy = 0, hz = 0, sz = 0;
        for (int t = 0; t <= T; t++) {
            int dim = t < T ? nz : nx; const double *PP = t < T ? p->P : p->PT;
            for (int i = 0; i < dim; i++) {
                double s = 0;
                for (int j = 0; j < dim; j++) s += PP[i * dim + j] * k->w[t * nz + j];
                k->Pw[t * nz + i] = s; wPw += s * k->w[t * nz + i];
            }
        }
        for (int t = 0; t < T; t++) {
            const double *C = Ct(p, t); const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
            double *rd = k->rd + t * nz;
            for (int j = 0; j < nz; j++) rd[j] = k->Pw[t * nz + j];
            for (int j = 0; j < nx; j++) {
                double s = k->lam[t * nx + j];
                for (int l = 0; l < nx; l++) s -= p->A[l * nx + j] * k->lam[(t + 1) * nx + l];
                rd[j] += s;
            }
            for (int j = 0; j < nu; j++) {
                double s = 0;
                for (int l = 0; l < nx; l++) s -= p->B[l * nu + j] * k->lam[(t + 1) * nx + l];
                rd[nx + j] += s;
            }
            for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) { rd[nx + nuc + b] += k->nuf[t * nub + b]; fy += fix[t * nub + b] * k->nuf[t * nub + b]; }
            for (int r = 0; r < m; r++) {
                if (!k->act[ro + r]) { k->rc[ro + r] = 0; continue; }
                double zr = k->z[ro + r], s = k->s[ro + r] - hh[r] * tau;
                for (int j = 0; j < nz; j++) { double c = C[r * nz + j]; if (c != 0.0) { rd[j] += c * zr; s += c * k->w[t * nz + j]; } }
                k->rc[ro + r] = s; hz += hh[r] * zr; sz += k->s[ro + r] * zr;
            }
            for (int i = 0; i < nx; i++) {
                double s = k->w[(t + 1) * nz + i];
                for (int l = 0; l < nx; l++) s -= p->A[i * nx + l] * k->w[t * nz + l];
                for (int l = 0; l < nu; l++) s -= p->B[i * nu + l] * k->w[t * nz + nx + l];
                k->rdyn[t * nx + i] = s;
            }
        }
        for (int j = 0; j < nx; j++) { k->rd[T * nz + j] = k->Pw[T * nz + j] + k->lam[T * nx + j]; fy += x0[j] * k->lam[j]; }
        double rg = wPw / tau + fy + hz + kap;
        double mu = (sz + tau * kap) / (mact + 1);

        /* ---- termination ---- */
        double rdinf = vmaxabs(k->rd, n), rcinf = fmax(vmaxabs(k->rc, M), vmaxabs(k->rdyn, T * nx));
        double winf = vmaxabs(k->w, n) / tau, zinf = fmax(vmaxabs(k->z, M), fmax(vmaxabs(k->lam, (T + 1) * nx), vmaxabs(k->nuf, T * nub))) / tau;
        double pobj = 0.5 * wPw / (tau * tau), dob = -0.5 * wPw / (tau * tau) - (fy + hz) / tau;
        double gap = fabs(pobj - dob);
        if (getenv("ORACLE_QP_TRACE")) fprintf(stderr, "it %3d tau %.3e kap %.3e mu %.3e rp %.3e rd %.3e gap %.3e pobj %.6e eta %.3e\n", it, tau, kap, mu, rcinf / tau, rdinf / tau, gap, pobj, -(fy + hz));
        /* Two levels (as in Ipopt's acceptable / desired tolerances).  ACCEPTABLE: scaled residuals and
         * gap <= tol.  DESIRED: acceptable and gap, dual residual <= 1e-2 tol -- they bound the suboptimality, and
         * with the curvature of this cost a 1e-8 gap still leaves ~2e-5 in the trajectory; two
         * implementations that stop an iteration apart must agree to 1e-5.  Once an acceptable iterate
         * exists, up to 3 more iterations are spent on the desired level; if one of them is worse
         * (precision floor of the linear algebra) it is undone and the acceptable iterate returned. */
        /* TOLERANCE ESCALATION (round 4).  Relaxations without strict complementarity (the random MLD of BASELINE
         * configs[4]: ~100 rows whose slack AND multiplier both vanish, binaries outside the cost) let the active-set
         * exchange of the polish cycle when the set is read from an iterate of gap 1e-8 (2 - 7 % of the optimal nodes of the
         * dive frontiers); the interior-point iterate such a solve returned instead is ~sqrt(gap) = 1e-4 off in the
         * trajectory.  From a tighter iterate the set is read right: when the LAST attempt has failed on the iterate the
         * solve would return, the stopping tolerance drops by 100 (twice at most: 1e-10, 1e-12), the iteration goes on and
         * the next iterate that meets it gets a last attempt of its own.  Measured on those frontiers: every optimal node
         * polishes at the first or second level, +0.2 iterations per node on average.  Nothing changes for a node whose
         * polish verifies -- every optimal node of the cart-pole systems.  The level is carried by npol (no state of its
         * own: the kernel's loop is short of registers): npol = POLISH_ATTEMPTS + 2 esc (+ 1 once the level's attempt is spent). */
        for (;;) {
            const int esc = npol > POLISH_ATTEMPTS ? (npol - POLISH_ATTEMPTS) / 2 : 0;
            const double tole = esc == 0 ? tol : esc == 1 ? 1e-2 * tol : 1e-4 * tol;
            const double gtol0 = tol * (1 + fmin(fabs(pobj), fabs(dob))), gtol = tole * (1 + fmin(fabs(pobj), fabs(dob)));
            const int acceptable = rcinf / tau <= tole * (1 + winf + x0inf) && rdinf / tau <= tole * (1 + zinf) && gap <= gtol;
            /* acceptable at the tolerance the caller asked for (what an escalated solve falls back on) */
            const int acc0 = rcinf / tau <= tol * (1 + winf + x0inf) && rdinf / tau <= tol * (1 + zinf) && gap <= gtol0;
            /* The polish is tried as soon as the iterate is good enough to read the active set from (ptol), once per
             * iterate, and not on an iterate that is about to be undone. */
            const double gptol = ptol * (1 + fmin(fabs(pobj), fabs(dob)));
            /* the barrier parameter is exhausted and the point is optimal to 100 x tol (1e-6, a simplex code's
             * default): nothing more can be gained from the interior-point iteration on an interior-free node */
            const int exh0 = mu < 1e-11 && rcinf / tau <= 100 * tol * (1 + winf + x0inf) && rdinf / tau <= 100 * tol * (1 + zinf) && gap <= 100 * gtol0;
            const int exhausted = status != ST_OPTIMAL && exh0;
            /* an escalated solve that no longer gets closer (ESC_ITERS iterations without meeting the tighter tolerance)
             * returns what it has, as long as that is acceptable to the caller's tolerance */
            const int stalled = esc > 0 && status == ST_OPTIMAL && !acceptable && (acc0 || exh0) && extra_done >= ESC_ITERS;
            /* (the iterate this solve would return: one last attempt from it even when the regular ones are used up --
             * they were spent on immature iterates; 5 of 4 000 optimal nodes of the one-wall system at N=40 ended that
             * way, one of them 3e-5 off) */
            const int desired = gap <= 1e-2 * gtol && rdinf / tau <= 1e-2 * tole * (1 + zinf);
            const int final_exit = (acceptable && (desired || extra_done >= 3 || it == max_iter)) || exhausted || stalled;
            const int ready = do_polish && tried_it != it &&
                              (npol < POLISH_ATTEMPTS ? (acceptable || exhausted || (status != ST_OPTIMAL && rcinf / tau <= ptol * (1 + winf + x0inf) &&
                                                                                       rdinf / tau <= ptol * (1 + zinf) && gap <= gptol))
                                                      : (((npol - POLISH_ATTEMPTS) & 1) == 0 && final_exit));
            if (ready) { npol++; tried_it = it; if (polish(p, k, x0, fix, tau, winf, zinf, last_alpha, NULL, npol > POLISH_ATTEMPTS)) { status = ST_OPTIMAL; polished = npol; break; } }
            if (do_polish && tried_it == it && npol > POLISH_ATTEMPTS && ((npol - POLISH_ATTEMPTS) & 1) == 1 && final_exit && esc < ESC_MAX && it < max_iter) {
                /* the last attempt of this level has failed on the iterate the solve would return: next level */
                npol++; extra_done = 0; status = ST_OPTIMAL;
                continue; /* (the exits below are decided at the new tolerance) */
            }
            if (acceptable) {
                status = ST_OPTIMAL;
                if (desired || extra_done >= 3 || it == max_iter) { leave = 1; break; }
                extra_done++;
            } else if (exhausted) {
                status = ST_OPTIMAL;
                leave = 1;
            } else if (status == ST_OPTIMAL) {
                if (esc > 0 && (acc0 || exh0)) { /* on the way to a tighter tolerance */
                    if (extra_done >= ESC_ITERS) leave = 1;
                    else extra_done++;
                    break;
                }
                for (int i = 0; i < n; i++) k->w[i] -= last_alpha * k->w2[i];
                for (int i = 0; i < (T + 1) * nx; i++) k->lam[i] -= last_alpha * k->lam2[i];
                for (int i = 0; i < T * nub; i++) k->nuf[i] -= last_alpha * k->nuf2[i];
                for (int r = 0; r < M; r++) if (k->act[r]) { k->z[r] -= last_alpha * k->z2[r]; k->s[r] -= last_alpha * k->rhs_c[r]; }
                tau -= last_alpha * last_dtau; kap -= last_alpha * last_dkap;
                for (int i = 0; i < nx; i++) k->w[i] = x0[i] * tau;
                for (int t = 0; t < T; t++) for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) k->w[t * nz + nx + nuc + b] = fix[t * nub + b] * tau;
                leave = 1;
            }
            break;
        }
        if (polished || leave) break;
        {   /* Farkas: E'y + C'z = rd - Pw, -(f'y + h'z) > 0 */
            double eta = -(fy + hz), cert = 0;
            for (int i = 0; i < n; i++) { double v = fabs(k->rd[i] - k->Pw[i]); if (v > cert) cert = v; }
            if (getenv("ORACLE_QP_TRACE")) fprintf(stderr, "      cert %.3e eta %.3e zinf %.3e raw rd %.3e rc %.3e\n", cert, eta, zinf * tau, rdinf, rcinf);
            /* (a) clean proof; (b) tau has collapsed against kappa: in exact arithmetic that alone
             * means "no optimum", and the proof is as accurate as double precision allows */
            /* Every exit carries a bound on the certificate residual: a ray that is not a proof must not
             * prune a subtree (a node whose tau vanishes without one ends MAXITER / NUMERICAL and is surfaced). */
            if (eta > 0 && (cert <= tol_inf * eta || (tau <= 1e-8 * kap && cert <= 1e-3 * eta))) {
                status = ST_INFEASIBLE; break;
            }
            /* (c) tau has collapsed but the ray is no proof to tolerance: the node is infeasible by about the accuracy
             * of the linear algebra (the least-violated point misses its rows by ~1e-6; measured on the published
             * sd = 0.01 runs).  The embedding's conclusion -- tau -> 0 with kappa > 0 and a cost bounded below:
             * no feasible point -- is taken (the node is pruned, as a simplex code with a 1e-6 feasibility tolerance
             * would), but the ray is flagged WEAK: it prunes this node and nothing else -- the warm-start shift never
             * carries it to the next step (the leaf is reopened). */
            /* (d) (round 4) tau has collapsed four more decades and still no ray verifies: a node on the very boundary between
             * feasible and infeasible -- f'y + h'z and E'y + C'z both go to zero, eta changes sign from one iteration to the
             * next (tests/golden/hard_node_sd003.npz, from the replay of the published sd = .003 runs with cold searches
             * only: the kernel ran it to 100 iterations, this code left at 37 through (b) by the luck of its rounding).
             * Same conclusion and flag as (c). */
            if ((eta > 0 && tau <= 1e-8 * kap && cert <= 0.5 * eta) || tau <= 1e-12 * kap) {
                /* (2: through (d) alone, no ray meets even the bound of (c): flagged UNCERTIFIED for the drivers to count) */
                status = ST_INFEASIBLE; weak = (eta > 0 && tau <= 1e-8 * kap && cert <= 0.5 * eta) ? 1 : 2; break;
            }
        }
        if (it == max_iter) break;

        /* ---- factorisation ---- */
        for (int r = 0; r < M; r++) k->D[r] = k->act[r] ? k->z[r] / k->s[r] : 0.0;
        if (factor(p, k, fix) != 0) { if (status != ST_OPTIMAL) status = ST_NUMERICAL; break; }

        /* ---- constant direction: rhs = (0; f; h) ---- */
        for (int t = 0; t < T; t++) { const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t]; for (int r = 0; r < m; r++) k->rhs_c[ro + r] = hh[r]; }
        kkt_solve(p, k, fix, NULL, x0, NULL, 1, k->rhs_c, k->w1, k->lam1, k->nuf1, k->z1);
        /* Denominator of the tau step:  kap/tau + w'Pw/tau^2 - 2 w'P w1/tau - (f'y1 + h'z1).  The constant direction solves
         * P w1 + E'y1 + C'z1 = 0, E w1 = f, C w1 - z1/D = h, hence -(f'y1 + h'z1) = w1'P w1 + z1'D^-1 z1 and
         *     den = kap/tau + (w/tau - w1)' P (w/tau - w1) + sum_i z1_i^2 / D_i :
         * a sum of nonnegative terms.  As the difference of the four O(1 .. 100) terms it was written as until round 4 it
         * loses everything in the last iterations -- the two quadratic terms go to zero like kap/tau -- as soon as the
         * solve of the constant direction is only good to 1e-8: on a deep node of BASELINE configs[4] the kernel's den came
         * out ten times too small at mu = 3e-11, tau dropped from 0.20 to 0.046 in one step and the dual residual went from
         * 5e-8 to 3e-5 (profiles/r04_den_cancellation.txt). */
        double q1 = 0, q2 = 0;
        for (int t = 0; t <= T; t++) {
            int dim = t < T ? nz : nx; const double *PP = t < T ? p->P : p->PT;
            for (int i = 0; i < dim; i++) {
                double s = 0;
                for (int j = 0; j < dim; j++) s += PP[i * dim + j] * (k->w[t * nz + j] / tau - k->w1[t * nz + j]);
                q1 += s * (k->w[t * nz + i] / tau - k->w1[t * nz + i]);
            }
        }
        for (int r = 0; r < M; r++) if (k->act[r]) q2 += k->z1[r] * k->z1[r] / k->D[r];
        double den = kap / tau + q1 + q2;

        double dtau_a = 0, dkap_a = 0, sigma = 0, alpha = 0;
        for (int pass = 0; pass < 2; pass++) {
            double lin = pass == 0 ? 1.0 : 1.0 - sigma;
            double dkap_rhs = tau * kap + (pass ? dtau_a * dkap_a - sigma * mu : 0.0);
            /* rhs_d = -lin rd ; cdyn = -lin rdyn ; rhs_c = -lin rc + ds/z */
            double *sd = k->sd, *sc = k->sc;
            for (int i = 0; i < n; i++) sd[i] = -lin * k->rd[i];
            for (int i = 0; i < T * nx; i++) sc[i] = -lin * k->rdyn[i];
            for (int r = 0; r < M; r++) {
                if (!k->act[r]) { k->rhs_c[r] = 0; continue; }
                double ds = k->s[r] * k->z[r] + (pass ? k->dsa[r] * k->dza[r] - sigma * mu : 0.0);
                k->rhs_c[r] = -lin * k->rc[r] + ds / k->z[r];
            }
            kkt_solve(p, k, fix, sd, NULL, sc, 0, k->rhs_c, k->w2, k->lam2, k->nuf2, k->z2);
            double g2 = 0, fy2 = 0, hz2 = 0;
            for (int i = 0; i < n; i++) g2 += k->Pw[i] * k->w2[i];
            g2 *= 2.0 / tau;
            for (int j = 0; j < nx; j++) fy2 += x0[j] * k->lam2[j];
            for (int t = 0; t < T; t++) {
                const double *hh = ht(p, t); int m = mt(p, t), ro = p->roff[t];
                for (int b = 0; b < nub; b++) if (fix[t * nub + b] == 1) fy2 += k->nuf2[t * nub + b];
                for (int r = 0; r < m; r++) if (k->act[ro + r]) hz2 += hh[r] * k->z2[ro + r];
            }
            double dtau = (lin * rg - dkap_rhs / tau + g2 + fy2 + hz2) / den;
            double dkap = -(dkap_rhs + kap * dtau) / tau;
            /* combined direction d = v2 + dtau v1 (kept in the "2" arrays) */
            for (int i = 0; i < n; i++) k->w2[i] += dtau * k->w1[i];
            for (int i = 0; i < (T + 1) * nx; i++) k->lam2[i] += dtau * k->lam1[i];
            for (int i = 0; i < T * nub; i++) k->nuf2[i] += dtau * k->nuf1[i];
            for (int r = 0; r < M; r++) k->z2[r] = k->act[r] ? k->z2[r] + dtau * k->z1[r] : 0.0;
            /* iterative refinement against the three linear blocks of the Newton system at this dtau,
             * K d = rhs2 + dtau rhs1: one step once mu < 1e-3, two once mu < 1e-7 (one step squares the
             * relative error of a solve, and the stage cost's small curvature needs the dual residual
             * well below the stopping tolerance for the trajectory to be accurate to 1e-5) */
            const int nref = (pass == 1 && refine && (it >= REFINE_FROM_IT || npol > 0 || !do_polish)) ? (mu < REFINE_MU2 ? 2 : mu < REFINE_MU ? 1 : 0) : 0;
            for (int rf = 0; rf < nref; rf++) {
                kkt_residual(p, k, fix, sd, sc, k->rhs_c, dtau, k->w2, k->lam2, k->nuf2, k->z2, k->ed, k->edyn, k->ec);
                kkt_solve(p, k, fix, k->ed, NULL, k->edyn, 0, k->ec, k->w1, k->lam1, k->nuf1, k->z1); /* v1 no longer needed */
                for (int i = 0; i < n; i++) k->w2[i] += k->w1[i];
                for (int i = 0; i < (T + 1) * nx; i++) k->lam2[i] += k->lam1[i];
                for (int i = 0; i < T * nub; i++) k->nuf2[i] += k->nuf1[i];
                for (int r = 0; r < M; r++) if (k->act[r]) k->z2[r] += k->z1[r];
            }
            /* slack step from the complementarity row, step length to the boundary */
            double amax = 1e30;
            if (dtau < 0) amax = fmin(amax, -tau / dtau);
            if (dkap < 0) amax = fmin(amax, -kap / dkap);
            for (int r = 0; r < M; r++) {
                if (!k->act[r]) { k->dsa[r] = 0; continue; }
                double dz = k->z2[r];
                double dsr = k->s[r] * k->z[r] + (pass ? k->dsa[r] * k->dza[r] - sigma * mu : 0.0);
                double ds = -(dsr + k->s[r] * dz) / k->z[r];
                if (dz < 0) amax = fmin(amax, -k->z[r] / dz);
                if (ds < 0) amax = fmin(amax, -k->s[r] / ds);
                if (pass == 0) { k->dza[r] = dz; k->dsa[r] = ds; } else k->rhs_c[r] = ds; /* rhs_c reused as ds */
            }
            if (pass == 0) {
                double aa = fmin(1.0, amax);
                sigma = (1 - aa) * (1 - aa) * (1 - aa);
                dtau_a = dtau; dkap_a = dkap;
            } else {
                /* Fraction to the boundary: 0.99 -- except in the END GAME of an infeasible node (round 5).  Once tau has fallen
                 * below kappa and keeps falling the Newton step wants tau -> 0 exactly; the certificate's residual goes with tau,
                 * so 0.99 buys two decades per iteration and three or four iterations pass between "eta > 0" and "residual <=
                 * 1e-6 eta".  There the fraction follows the barrier, 1 - max(mu, 1e-5) (as Ipopt's tau = max(tau_min, 1 - mu)):
                 * one iteration less on every infeasible node (12.3 -> 11.4 on the headline tree), the same statuses, no ray
                 * left WEAK (without the floor of 1e-5 tau outruns the certificate: two WEAK exits on BASELINE configs[4]).
                 * Optimal nodes never get here (tau stays O(1)): their records are bit for bit those of the fixed fraction. */
                alpha = fmin(1.0, ((dtau < 0 && tau < kap) ? fmax(0.99, 1.0 - fmax(mu, 1e-5)) : 0.99) * amax);
                if (getenv("ORACLE_QP_TRACE")) fprintf(stderr, "      sigma %.3e alpha %.3e dtau %.3e\n", sigma, alpha, dtau);
                for (int i = 0; i < n; i++) k->w[i] += alpha * k->w2[i];
                for (int i = 0; i < (T + 1) * nx; i++) k->lam[i] += alpha * k->lam2[i];
                for (int i = 0; i < T * nub; i++) k->nuf[i] += alpha * k->nuf2[i];
                for (int r = 0; r < M; r++) if (k->act[r]) { k->z[r] += alpha * k->z2[r]; k->s[r] += alpha * k->rhs_c[r]; }
                tau += alpha * dtau; kap += alpha * dkap;
                last_alpha = alpha; last_dtau = dtau; last_dkap = dkap;
                for (int i = 0; i < nx; i++) k->w[i] = x0[i] * tau;
                for (int t = 0; t < T; t++) for (int b = 0; b < nub; b++) if (fix[t * nub + b] >= 0) k->w[t * nz + nx + nuc + b] = fix[t * nub + b] * tau;
            }
        }
        if (!(tau > 0) || !(kap >= 0) || tau != tau) { status = ST_NUMERICAL; break; }
        if (status == ST_OPTIMAL && it + 1 > max_iter) break;
    }
output:
    *iters = it;
    if (polished) { /* the polished point replaces the iterate (tau = 1 units; exactly complementary) */
        memcpy(k->w, k->w1, sizeof(double) * n); memcpy(k->lam, k->lam1, sizeof(double) * (T + 1) * nx);
        memcpy(k->nuf, k->nuf1, sizeof(double) * T * nub);
        for (int r = 0; r < M; r++) k->z[r] = k->D[r] >= 1.0 ? k->dza[r] : 0.0;
        tau = 1.0;
    }
    if (polished_out) *polished_out = polished | ((weak ? 1 : 0) << 8) | ((weak == 2 ? 1 : 0) << 10);

    /* ---- outputs in the reference's conventions ---- */
    int nmu = (T - 1) * p->nc + p->ncL;
    double *lam_o = dual, *mu_o = dual + (T + 1) * nx, *nlb_o = mu_o + nmu, *nub_o = nlb_o + T * nub;
    double *rho_o = nub_o + T * nub, *sig_o = rho_o + T * p->nq + p->nqT;
    int nd = (T + 1) * nx + nmu + 2 * T * nub + T * p->nq + p->nqT + T * p->nr;
    double *x_o = primal, *u_o = primal + (T + 1) * nx;
    double scale;
    if (status == ST_INFEASIBLE) {
        double big = fmax(vmaxabs(k->z, M), fmax(vmaxabs(k->lam, (T + 1) * nx), vmaxabs(k->nuf, T * nub)));
        scale = 1.0 / big;   /* Farkas ray: scale is arbitrary, normalise the largest scaled multiplier to 1 */
    } else scale = 1.0 / (tau * p->cs);
    for (int i = 0; i < nd; i++) dual[i] = 0;
    for (int i = 0; i < (T + 1) * nx; i++) lam_o[i] = k->lam[i] * scale;
    double farkas = 0;
    for (int j = 0; j < nx; j++) farkas -= x0[j] * lam_o[j];
    for (int t = 0; t < T; t++) {
        int m = mt(p, t), mg = m - 2 * nub, ro = p->roff[t]; const double *sc = t < T - 1 ? p->sreg : p->slast;
        const double *hh = ht(p, t);
        for (int r = 0; r < mg; r++) { double v = k->z[ro + r] * scale * sc[r]; mu_o[t * p->nc + r] = v; farkas -= (hh[r] / sc[r]) * v; }
        for (int b = 0; b < nub; b++) {
            double lo, hi;
            if (fix[t * nub + b] < 0) { lo = k->z[ro + mg + b] * scale; hi = k->z[ro + mg + nub + b] * scale; farkas -= hi; }
            else { double v = k->nuf[t * nub + b] * scale; hi = v > 0 ? v : 0; lo = v < 0 ? -v : 0; farkas -= fix[t * nub + b] * (hi - lo); }
            nlb_o[t * nub + b] = lo; nub_o[t * nub + b] = hi;
        }
    }
    *term_viol = 0;
    if (status == ST_INFEASIBLE) {
        for (int i = 0; i < (T + 1) * nx + T * nu; i++) primal[i] = NAN;
        *obj = INFINITY; *dobj = farkas;
    } else {
        double cost = 0;
        for (int t = 0; t <= T; t++) for (int i = 0; i < nx; i++) x_o[t * nx + i] = k->w[t * nz + i] / tau;
        for (int t = 0; t < T; t++) for (int i = 0; i < nu; i++) u_o[t * nu + i] = k->w[t * nz + nx + i] / tau;
        for (int t = 0; t <= T; t++) {
            const double *QQ = t < T ? p->Q : p->QT; int rows = t < T ? p->nq : p->nqT;
            double *rho = rho_o + t * p->nq;
            for (int r = 0; r < rows; r++) {
                double s = 0;
                for (int j = 0; j < nx; j++) s += QQ[r * nx + j] * x_o[t * nx + j];
                rho[r] = 2 * s; cost += s * s;
            }
        }
        for (int t = 0; t < T; t++) for (int r = 0; r < p->nr; r++) {
            double s = 0;
            for (int j = 0; j < nu; j++) s += p->R[r * nu + j] * u_o[t * nu + j];
            sig_o[t * p->nr + r] = 2 * s; cost += s * s;
        }
        *obj = cost;
        /* largest value of (scaled terminal row) - h over the terminal-set rows */
        double tv = -INFINITY;
        for (int r = p->nc; r < p->ncL; r++) {
            double a = -p->hlast[r] * tau;
            for (int j = 0; j < nz; j++) a += p->Clast[r * nz + j] * k->w[(T - 1) * nz + j];
            if (a / tau > tv) tv = a / tau;
        }
        *term_viol = tv;
        /* dual objective of SURVEY.md Appendix A.3 from the multipliers themselves */
        double dq = 0;
        for (int i = 0; i < T * p->nq + p->nqT; i++) dq += rho_o[i] * rho_o[i];
        for (int i = 0; i < T * p->nr; i++) dq += sig_o[i] * sig_o[i];
        *dobj = -0.25 * dq + farkas;
    }
    return status;
}

/* ---- C entry point used through ctypes ---- */
int oracle_solve_batch(int nx, int nu, int nub, int T, int nc, int ncL, int nq, int nr, int nqT,
                       const double *A, const double *B, const double *F, const double *G, const double *h,
                       const double *FL, const double *GL, const double *hL,
                       const double *Q, const double *R, const double *QT,
                       const double *x0, int x0_stride, int nbatch, const int8_t *fix,
                       double tol, double tol_inf, int max_iter, int nthreads, int lazy_terminal, int refine, int do_polish, double ptol,
                       const double *warm_primal, const double *warm_dual, const int32_t *warm_index,
                       double *obj, double *dobj, int *status, int *iters, double *primal, double *dual, int *polished)
{
    if (nx + nu > 64 || nx > 32) return -1;
    prob_t *p = prob_create(nx, nu, nub, T, nc, ncL, nq, nr, nqT, A, B, F, G, h, FL, GL, hL, Q, R, QT);
    int np_ = (T + 1) * nx + T * nu;
    int nd = (T + 1) * nx + (T - 1) * nc + ncL + 2 * T * nub + T * nq + nqT + T * nr;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
#pragma omp parallel
    {
        work_t *k = work_create(p);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < nbatch; b++) {
            /* Lazy terminal set: first without the terminal-set rows.  An infeasibility proof found
             * there is a proof for the node and carries no terminal multipliers (it survives the
             * warm-start shift); an optimum that satisfies the masked rows strictly is the optimum
             * of the node.  Otherwise solve again with every row. */
            const double *xb = x0 + (size_t)b * x0_stride; const int8_t *fb = fix + (size_t)b * T * nub;
            double *pb = primal + (size_t)b * np_, *db = dual + (size_t)b * nd, tv = 0; int it1 = 0, it2 = 0, st, pol = 0, second = 0;
            /* the parent's record, if one is handed down (rows of warm_primal / warm_dual; may be the output arrays of an
             * earlier call, never rows of this call's outputs) */
            const int wi = (warm_index && warm_primal && warm_dual) ? warm_index[b] : -1;
            const double *wp = wi >= 0 ? warm_primal + (size_t)wi * np_ : NULL, *wd = wi >= 0 ? warm_dual + (size_t)wi * nd : NULL;
            if (ncL > nc && lazy_terminal) {
                /* A parent whose optimum lies on terminal-set rows hands those down too: its set is first tried WITH the
                 * terminal rows (with them masked the point is far from the parent's and the hand-down drops out at
                 * once, leaving a full first solve before the second one verifies).  Only an attempt: if it does not
                 * verify, the regular sequence -- masked first, so that an infeasible node's ray carries no terminal
                 * multipliers -- runs as without it. */
                int parent_term = 0;
                st = -1;
                if (wd && do_polish) {
                    const double *mu_last = wd + (T + 1) * nx + (size_t)(T - 1) * nc;
                    for (int r = nc; r < ncL; r++) if (mu_last[r] > 0.0) parent_term = 1;
                }
                if (parent_term) {
                    st = solve_one(p, k, xb, fb, 1, refine, do_polish, ptol, tol, tol_inf, max_iter, wp, wd, 1, obj + b, dobj + b, &it1, pb, db, &tv, &pol);
                    if (st >= 0) second = 1;
                }
                if (st < 0) {
                    st = solve_one(p, k, xb, fb, 0, refine, do_polish, ptol, tol, tol_inf, max_iter, wp, wd, 0, obj + b, dobj + b, &it1, pb, db, &tv, &pol);
                    if (!(st == ST_INFEASIBLE || (st == ST_OPTIMAL && tv < 0.0)) && (second = 1))
                        st = solve_one(p, k, xb, fb, 1, refine, do_polish, ptol, tol, tol_inf, max_iter, wp, wd, 0, obj + b, dobj + b, &it2, pb, db, &tv, &pol);
                }
            } else st = solve_one(p, k, xb, fb, 1, refine, do_polish, ptol, tol, tol_inf, max_iter, wp, wd, 0, obj + b, dobj + b, &it1, pb, db, &tv, &pol);
            status[b] = st; iters[b] = it1 + it2; if (polished) polished[b] = pol | (second ? 0x200 : 0);
        }
        work_free(k);
    }
    prob_free(p);
    return 0;
}
