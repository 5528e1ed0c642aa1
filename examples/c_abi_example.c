/* A non-Python caller of libhmpc.so: plain C99 against include/hmpc.h.
 *
 *   gcc -std=c99 -I include examples/c_abi_example.c -L warm-start-hybrid-mpc_amd -lhmpc -Wl,-rpath,$PWD/warm-start-hybrid-mpc_amd -lm -o c_abi_example
 *
 * The problem is a toy mixed logical dynamical system small enough to state inline -- a double integrator whose
 * second input is a binary "boost" that adds to the force (x+ = A x + B [u; b], |x| <= 5, |u| <= 1, 0 <= b <= 1,
 * cost |x|^2 + |u|^2 per stage, horizon 6) -- and the calls are the ones a binding in another language makes:
 * hmpc_create, hmpc_record_sizes, hmpc_solve_batch on the root and its two children (the role of one round of
 * branch_and_bound.py:462-493), hmpc_lp_solve_batch on the facet LPs of the unit box, hmpc_destroy.
 * Exit code 0 iff every answer is the known one.  tests/test_capi.py compiles it everywhere and runs it on the GPU box. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "hmpc.h"

#define NX 2
#define NU 2
#define NUB 1
#define HORIZON 6
#define NC 8

static int fail(const char *what)
{
    fprintf(stderr, "c_abi_example: %s: %s\n", what, hmpc_last_error());
    return 1;
}

int main(void)
{
    const double dt = 0.5;
    const double A[NX * NX] = {1, dt, 0, 1};
    const double B[NX * NU] = {0, 0, dt, 2 * dt};                 /* the binary input doubles the force */
    /* rows of [F G | h]:  +-x1 <= 5, +-x2 <= 5, +-u <= 1, b <= 1, -b <= 0 */
    const double F[NC * NX] = {1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0};
    const double G[NC * NU] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, -1, 0, 0, 1, 0, -1};
    const double h[NC] = {5, 5, 5, 5, 1, 1, 1, 0};
    const double Q[NX * NX] = {1, 0, 0, 1}, R[1 * NU] = {1, 0};
    hmpc_problem p;
    p.nx = NX; p.nu = NU; p.nub = NUB; p.T = HORIZON; p.nc = NC; p.ncT = NC; p.nq = NX; p.nr = 1; p.nqT = NX;
    p.A = A; p.B = B; p.F = F; p.G = G; p.h = h; p.F_Tm1 = F; p.G_Tm1 = G; p.h_Tm1 = h; p.Q = Q; p.R = R; p.Q_T = Q;

    hmpc_handle *qp = NULL;
    if (hmpc_create(&p, NULL, &qp) != HMPC_OK) return fail("hmpc_create");
    int32_t n_primal = 0, n_dual = 0;
    if (hmpc_record_sizes(qp, &n_primal, &n_dual) != HMPC_OK) return fail("hmpc_record_sizes");
    if (n_primal != (HORIZON + 1) * NX + HORIZON * NU) { fprintf(stderr, "unexpected record size %d\n", n_primal); return 1; }

    /* three nodes: the root relaxation, and the children that fix the first binary to 0 and to 1 */
    enum { BATCH = 3 };
    const double x0[NX] = {4.0, 0.0};
    int8_t fix[BATCH * HORIZON * NUB];
    for (int i = 0; i < BATCH * HORIZON * NUB; i++) fix[i] = -1;
    fix[1 * HORIZON * NUB + 0] = 0;
    fix[2 * HORIZON * NUB + 0] = 1;
    double obj[BATCH], dual_obj[BATCH];
    int32_t status[BATCH], iters[BATCH];
    double *primal = (double *)malloc(sizeof(double) * BATCH * n_primal);
    double *dual = (double *)malloc(sizeof(double) * BATCH * n_dual);
    hmpc_result out;
    out.obj = obj; out.dual_obj = dual_obj; out.status = status; out.iters = iters; out.primal = primal; out.dual = dual;
    if (hmpc_solve_batch(qp, x0, 0, fix, BATCH, NULL, &out) != HMPC_OK) return fail("hmpc_solve_batch");
    int bad = 0;
    for (int k = 0; k < BATCH; k++) {
        printf("node %d: status %d, objective %.9f (dual %.9f), %d iterations%s, u_0 = (%.6f, %.6f)\n", k, status[k], obj[k], dual_obj[k],
               iters[k] & 0xFFFF, (iters[k] & HMPC_ITERS_POLISHED) ? ", polished" : "", primal[k * n_primal + (HORIZON + 1) * NX],
               primal[k * n_primal + (HORIZON + 1) * NX + 1]);
        if (status[k] != HMPC_OPTIMAL || fabs(obj[k] - dual_obj[k]) > 1e-6 * (1 + fabs(obj[k]))) bad = 1;   /* strong duality */
    }
    /* a child is a restriction of its parent: its bound cannot be lower; the better child is the parent's bound or above */
    if (obj[1] < obj[0] - 1e-7 || obj[2] < obj[0] - 1e-7) bad = 1;
    /* the fixed binaries come back as fixed */
    if (fabs(primal[1 * n_primal + (HORIZON + 1) * NX + 1] - 0.0) > 1e-9 || fabs(primal[2 * n_primal + (HORIZON + 1) * NX + 1] - 1.0) > 1e-9) bad = 1;
    /* the two children once more, each handed the record of its parent (node 0): same optimum, and the child whose
     * optimum lies on the parent's active set needs no interior-point iteration (controller.py:260-264 hands down the
     * simplex basis; hmpc_warm the parent's record) */
    if ((iters[0] & HMPC_ITERS_POLISHED) != 0) {
        const int32_t parent[2] = {0, 0};
        double wobj[2], wdobj[2];
        int32_t wstatus[2], witers[2];
        hmpc_warm warm;
        warm.primal = primal; warm.dual = dual; warm.index = parent; warm.rows = 1;
        hmpc_result wout;
        wout.obj = wobj; wout.dual_obj = wdobj; wout.status = wstatus; wout.iters = witers; wout.primal = NULL; wout.dual = NULL;
        if (hmpc_solve_batch(qp, x0, 0, fix + HORIZON * NUB, 2, &warm, &wout) != HMPC_OK) return fail("hmpc_solve_batch (hand-down)");
        for (int k = 0; k < 2; k++) {
            printf("node %d, parent's record handed down: status %d, objective %.9f, %d iterations%s\n", k + 1, wstatus[k], wobj[k],
                   witers[k] & 0xFFFF, (witers[k] & HMPC_ITERS_HANDED) ? ", the parent's active set verified" : "");
            if (wstatus[k] != status[k + 1] || fabs(wobj[k] - obj[k + 1]) > 1e-8 * (1 + fabs(obj[k + 1]))) bad = 1;
        }
    }
    /* an impossible initial state: status infeasible, objective +inf, a Farkas objective > 0 */
    const double far[NX] = {50.0, 0.0};
    if (hmpc_solve_batch(qp, far, 0, fix, 1, NULL, &out) != HMPC_OK) return fail("hmpc_solve_batch (infeasible)");
    printf("x0 outside the state bounds: status %d, objective %f, Farkas objective %.3e\n", status[0], obj[0], dual_obj[0]);
    if (status[0] != HMPC_INFEASIBLE || !isinf(obj[0]) || !(dual_obj[0] > 0)) bad = 1;
    if (hmpc_destroy(qp) != HMPC_OK) return fail("hmpc_destroy");

    /* the LP entry point: max +-x_i over the box |x_i| <= 1 relaxed by one unit on the row of the cost (mcais.py:169-182) */
    const double E[4 * 2] = {1, 0, -1, 0, 0, 1, 0, -1}, f[4] = {1, 1, 1, 1};
    const int32_t relax[4] = {0, 1, 2, 3};
    double lobj[4], lx[4 * 2], lz[4 * 4];
    int32_t lstatus[4], liters[4];
    if (hmpc_lp_solve_batch(-1, 2, 4, E, E, 2, f, 0, relax, 4, 0.0, 0, lobj, lx, lz, lstatus, liters) != HMPC_OK)
        return fail("hmpc_lp_solve_batch");
    for (int k = 0; k < 4; k++) {
        printf("LP %d: status %d, value %.12f\n", k, lstatus[k], lobj[k]);
        if (lstatus[k] != HMPC_OPTIMAL || fabs(lobj[k] - 2.0) > 1e-12) bad = 1;
    }
    free(primal);
    free(dual);
    printf(bad ? "c_abi_example: WRONG ANSWER\n" : "c_abi_example: ok\n");
    return bad;
}
