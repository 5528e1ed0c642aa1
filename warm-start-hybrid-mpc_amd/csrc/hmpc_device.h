// hmpc_device.h -- structures shared by the host side of the C ABI and the kernel.
#ifndef HMPC_DEVICE_H
#define HMPC_DEVICE_H

#include <stdint.h>

#include "hmpc.h"

// Constraint rows of a stage: [F G] rows scaled to unit 2-norm, then the bound rows of the binaries
// (-ub <= -lb, then ub <= ubmax).  All arrays are device pointers.
struct SparseStage {
    int m, mg;                 // rows with / without the bound rows
    const int *rptr, *rcol;    // rows (CSR):    C w        row-parallel
    const double *rval;
    const int *cptr, *crow;    // columns (CSC): C' v       entry-parallel
    const double *cval;
    const int *gptr, *grow;    // Gram lists:    (C' D C)(i,j) = sum_k gval[k] D[grow[k]], i >= j
    const double *gval;
    const double *h;           // m scaled right-hand sides
    const double *scale;       // mg row scales
};

#define HMPC_KC_STRIDE 16

struct DevProb {
    int nx, nu, nub, nuc, nz, T, nc, ncL, nT, mreg, Toff, M, Mpad, n, ne, nq, nr, nqT, n_primal, n_dual;
    unsigned mreg_magic;                       // ceil(2^32 / mreg): r / mreg == umulhi(r, magic) for r < 2^16
    int nnz0, nng0;                            // nonzeros / Gram terms of the regular stage (staged in LDS)
    SparseStage reg;                           // stage rows: [F G] (nc) then the bounds of the binaries (2 nub)
    const double *Creg;                        // the same rows as a dense mreg x nz matrix
    const double *ccv;                         // columns of the stage rows padded to HMPC_KC_STRIDE (value, local row)
    const int *cci;
    int kcol;                                  // longest column of the stage rows
    int ngram;                                 // entries of C' D C with a nonempty term list (numbered first)
    const double *F_raw, *G_raw, *h_raw, *hT_raw; // unscaled [F G | h] and h_Tm1 (warm-start shift, hmpc_shift.hip)
    const double *shift_Mmu, *shift_Mrho, *shift_V; // maps of the shift (hmpc_set_shift_maps), null until set
    int *work_counter;                         // nodes handed out beyond the first gridDim.x (zeroed before each launch)
    unsigned *check_flag;                      // diagnostic build (HMPC_CHECK): bits raised by failed in-kernel checks; follows work_counter
    double *fac_ws;                            // streaming form: per-workgroup slab for multipliers and cost-to-go
    int fac_stride;                            //   doubles per workgroup
    int ring;                                  //   stages per chunk the sweeps of a solve stage in LDS (1 or 2: what fits)
    // generic kernel, nz >= 16: the stage rows split into DENSE rows (two or more nonzeros: matrix-core contractions and
    // dense products) and SINGLETON rows (one nonzero -- bounds; they only touch the diagonal of C'DC and one component
    // of a product), staged in LDS when they fit beside the rest (split_lds)
    int nd, ndp, ns;                           // dense rows, the same padded to a multiple of 4 (zero rows), singleton rows
    const double *Cdn;                         // ndp x nz
    const int *drow;                           // ndp: local row of dense row k (padding: row 0, zero coefficients)
    const int *rinfo;                          // mreg: singleton row -> its column; dense row k -> -(k + 1)
    const double *sval;                        // mreg: coefficient of a singleton row
    const int *sptr, *srow;                    // singleton rows by column: sptr[nz + 1], srow[ns]
    const int *nrow;                           // the rows that are not dense, in row order (mreg - nd)
    unsigned nd_magic, nn_magic;               // ceil(2^32 / nd), ceil(2^32 / (mreg - nd)): divisions of a row position (as mreg_magic)
    int split_lds;
    int static_rows;                           // every [F G] row has at most two nonzero input coefficients
    const double *Ct, *ht, *sct;               // terminal-set rows of the last stage: dense nT x nz, rhs, row scales
    const double *A, *B, *P, *PT, *Q, *R, *QT; // P = 2 cs (Q'Q (+) R'R), PT = 2 cs QT'QT
    const int *ei, *ej;                        // lower-triangle entry -> (i, j)
    double cs;                                 // cost scale
    double tol, tol_inf;
    int max_iter, lazy, refine;
    int dbg;                                   // diagnostic build only (HMPC_DBG): phases to skip when timing
    int polish;                                // active-set polish of optimal iterates (hmpc_options.polish)
    int polish_l1;                             //   1: the polish runs at its second penalty level (1e7) from the start and ends there.  The two levels
                                               //   exist for costs of tiny curvature (cart-pole: 1.4e-4 after scaling -- eps rho over the curvature is the
                                               //   accuracy of the vertex, so the first choice there is 1e5 and a set verified at 1e7 goes through 1e5 once more);
                                               //   with a curvature of order one (smallest positive diagonal entry of the scaled Hessians >= 1e-2) 1e7 gives
                                               //   1e-9 and the dance between the levels only costs factorisations (configs[4]: 22 % of the polish rounds)
    double ptol;                               //   tried once the scaled residuals and gap are below ptol
};

struct DevOut {
    double *obj, *dual_obj;
    int32_t *status, *iters;
    double *primal, *dual;
};

// Parent records handed down to the nodes of a launch (hmpc_warm): node b tries the active set of row index[b] of
// (primal, dual) before its first interior-point iteration; index null or index[b] < 0: none.
struct DevWarm {
    const double *primal, *dual;
    const int32_t *index;
    // TWO-LAUNCH FORM of the lazy terminal set (large cold batches, hmpc_solve_batch_device): the first launch solves every
    // node with the terminal-set rows masked and, instead of solving a node that needs them a second time in place --
    // 24+ iterations at one wave per node among short nodes: the tail of the launch --, appends it to `pend` (pend[0]: how
    // many, pend[1 ..]: which); the second launch (`second` set; four waves per node) takes those nodes only, each from its
    // own first record (primal / dual are then the output arrays): the active set of the masked solve plus the terminal
    // rows it violates is tried before anything else.
    int32_t *pend;
    int32_t second;
};

// LDS bytes per workgroup (must mirror the carve in hmpc_qp_kernel).  kc: entries per padded column of
// the kernel's compile-time shape (Dims::kKC), 0 for the generic kernel; big: the generic kernel's streaming
// form (lists and Riccati factor in global memory).
static inline size_t hmpc_lds_bytes(const DevProb &p, int kc, int big)
{
    const size_t n = p.n, T = p.T, nx = p.nx, nu = p.nu, nz = p.nz, nub = p.nub, M = p.M;
    size_t d = 0, i = 0, b = 0;
    const size_t nxs = nx * (nx + 1) / 2, lms = nx * nu + nu * (nu - 1) / 2;
    const size_t dir = n + (T + 1) * nx + T * nub, fscr = nz * nz + (kc > 0 ? nz * nz : nx * nz);
    d += dir;                                                             // w lam nuf
    d += M;                                                               // e (row vector: z / D / D.*rhs / dz in turn)
    d += T * nu + (big ? 0 : T * lms + (T + 1) * nxs);                    // dinv ; Lm Pr (global slab if big)
    d += n + T * nx + (T + 1) * nx + n + (T + 1) * nx;                    // rd rdyn edyn g pv
    d += dir + (dir > fscr ? dir : fscr);                                 // w1.. ; w2.. (doubles as factor scratch)
    d += (nx + 3) / 4 * 4 + nz + 40;                                      // q (padded to four) mv red
    d += nx;                                                              // x0
    if (big) {                                                            // AB padded, P, PT
        const size_t nur = (nu + 3) / 4 * 4, nup = (nu + 1 + 3) / 4 * 4, lrows = nz + 1 > nx + nur ? nz + 1 : nx + nur;
        d += ((nx + 1 + 3) / 4 * 4) * lrows + nz * nz + nx * nx;
        d += 2 * (size_t)p.ring * lrows * nup + lms;                      // two chunks of `ring` padded stage blocks of multipliers; the stage a factorisation works on
    } else {
        d += nx * nz + nz * nz + nx * nx;
    }
    i += 2 + T * nub + 2 * (size_t)p.ne;                                  // flag fix ei ej
    if (!big) {
        i += (p.ne + 1) + p.nng0;                                         // gptr0 grow0
        d += p.mreg + p.nng0;                                             // h0 gval0
    }
    if (big) {
        // the lists stay in global memory
    } else if (kc > 0) {
        d += nz * (size_t)kc;                                             // padded column values
        b += (nz * (size_t)kc + 3) / 4 * 4;                               // padded column rows (bytes)
    } else {
        d += 2 * (size_t)p.nnz0;                                          // rval0 cval0
        i += (p.mreg + 1) + p.nnz0 + (nz + 1) + p.nnz0;                   // rptr0 rcol0 cptr0 crow0
    }
    if (kc == 0 && p.split_lds) {                                          // split stage rows (dense block, singleton lists)
        d += (size_t)p.ndp * nz + p.mreg;
        i += (size_t)p.ndp + p.mreg + (nz + 1) + p.ns + (p.mreg - p.nd);
    }
    return d * sizeof(double) + i * sizeof(int) + b;
}

#endif
