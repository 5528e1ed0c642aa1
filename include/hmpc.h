/*
 * hmpc.h -- C ABI of the MI355X batched QP-relaxation solver for hybrid MPC.
 *
 * This is the drop-in boundary for ONE hot path of TobiaMarcucci/warm-start-hybrid-mpc:
 * the QP relaxation solved at every branch-and-bound node.  In the reference that path is
 *
 *   warm_start_hmpc/controller.py:365-376   solver(identifier, cutoff, extra) closure
 *   warm_start_hmpc/controller.py:229-271   _solve_subproblem  (set bounds, set x0, optimize)
 *   warm_start_hmpc/controller.py:273-298   _set_bound_binaries
 *   warm_start_hmpc/bounded_qp.py:200-228   BoundedQP.optimize (+ Farkas proof if infeasible)
 *   warm_start_hmpc/subproblem_solution.py:18-168  extraction of the primal/dual record
 *
 * and it bottoms out in Gurobi's C library through gurobipy.  The entry points below are
 * what a binding for that path needs: build the node-independent QP once (the role of
 * controller.py:119-184 _build_mip), then solve any number of nodes -- each given by its
 * vector of binary fixings and an initial state -- in one call.  Plain pointers and sizes
 * only; all matrices row-major float64.
 *
 * The QP of one node (reference statement: controller.py:47-56, 119-184):
 *
 *   min  sum_{t<T} |Q x_t|^2 + |R u_t|^2  +  |Q_T x_T|^2
 *   s.t. x_0 = x0                                  (multiplier lam_0)
 *        x_{t+1} = A x_t + B u_t                   (lam_{t+1})
 *        F x_t + G u_t <= h            t < T-1     (mu_t >= 0)
 *        F_Tm1 x_t + G_Tm1 u_t <= h_Tm1  t = T-1   (mu_{T-1} >= 0)
 *        lb_t <= ub_t <= ubmax_t                   (nu_lb_t, nu_ub_t >= 0)
 *
 * with u_t = (uc_t, ub_t), ub_t the last nub entries, and (lb, ubmax) = (0, 1) for a free
 * binary, (v, v) for a binary fixed to v.  Sign conventions are the reference's
 * (bounded_qp.py:260-332): inequality multipliers are nonnegative; an infeasible node has
 * objective +inf, no primal, and multipliers that form a Farkas proof whose "dual
 * objective" -sum(rhs * multiplier) is positive.
 */
#ifndef HMPC_H
#define HMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-node status (hmpc_result.status) */
#define HMPC_OPTIMAL 0
#define HMPC_INFEASIBLE 1
#define HMPC_MAXITER 2   /* not converged: the caller must not use the record */
#define HMPC_NUMERICAL 3 /* numerical breakdown: idem */
#define HMPC_UNBOUNDED 4 /* hmpc_lp_solve_batch only: the cost grows without bound on the set */

#define HMPC_ITERS_POLISHED 0x10000 /* flag in hmpc_result.iters */
#define HMPC_ITERS_WEAK 0x20000     /* flag in hmpc_result.iters: HMPC_INFEASIBLE, but the ray is no proof to tolerance */
#define HMPC_ITERS_TERMINAL 0x80000 /* flag in hmpc_result.iters (launches with hmpc_warm, and large cold batches in the two-launch form): the terminal-set rows were needed */
#define HMPC_ITERS_HANDED 0x40000   /* flag in hmpc_result.iters: the active set handed down by the parent (hmpc_warm) verified */
#define HMPC_ITERS_UNCERTIFIED 0x100000 /* flag in hmpc_result.iters, with HMPC_ITERS_WEAK: the node was pruned on the collapse of tau ALONE
                                         * (tau <= 1e-12 kappa), no ray met even the loose bound of the weak exit -- the drivers count
                                         * such prunes and say so when one removed a node whose bound lay below the final incumbent */

/* return codes */
#define HMPC_OK 0
#define HMPC_EINVAL -1 /* bad argument (sizes, null pointers) */
#define HMPC_EDEVICE -2 /* HIP error; text in hmpc_last_error() */
#define HMPC_ETOOBIG -3 /* the per-node vectors exceed one CU's LDS even with lists and factor in global memory */

/* Node-independent problem data (replaces the Gurobi model built by controller.py:119-184).
 * nc  = rows of [F G];  ncT = rows of [F_Tm1 G_Tm1] (= nc + terminal facets, controller.py:85-87).
 * nq, nr, nqT = rows of Q, R, Q_T. */
typedef struct hmpc_problem {
    int32_t nx, nu, nub, T, nc, ncT, nq, nr, nqT;
    const double *A;     /* nx  x nx */
    const double *B;     /* nx  x nu */
    const double *F;     /* nc  x nx */
    const double *G;     /* nc  x nu */
    const double *h;     /* nc       */
    const double *F_Tm1; /* ncT x nx */
    const double *G_Tm1; /* ncT x nu */
    const double *h_Tm1; /* ncT      */
    const double *Q;     /* nq  x nx */
    const double *R;     /* nr  x nu */
    const double *Q_T;   /* nqT x nx */
} hmpc_problem;

/* Solver options (the role of the reference's gurobi_params, controller.py:778-796). */
typedef struct hmpc_options {
    double tol;            /* optimality: scaled residuals and relative gap (default 1e-8)   */
    double tol_inf;        /* infeasibility proof: |A'y| <= tol_inf * (-b'y) (default 1e-6) */
    int32_t max_iter;      /* interior-point iterations per solve (default 100)             */
    int32_t lazy_terminal; /* 1: try each node without the terminal-set rows first          */
    int32_t refine;        /* 1: one step of iterative refinement per Newton direction      */
    int32_t device;        /* HIP device ordinal, -1 = current                              */
    int32_t polish;        /* 1: active-set polish -- an optimal node returns the vertex     */
                           /*    solution of its active set (what Gurobi's simplex/crossover */
                           /*    returns, bounded_qp.py:208), exactly complementary          */
    int32_t reserved;      /* 0                                                             */
    double polish_tol;     /* the polish is tried once residuals and gap are below this     */
                           /*    (default 1e-4; it is accepted only if it verifies)          */
} hmpc_options;

/* Structure-of-arrays result of a batch of B nodes.  Any pointer may be NULL (not wanted).
 * primal row (n_primal doubles): x_0..x_T (nx each), then u_0..u_{T-1} (nu each; uc then ub).
 *   NaN for an infeasible node.
 * dual row (n_dual doubles), in this order (subproblem_solution.py:142-166):
 *   lam_0..lam_T (nx) | mu_0..mu_{T-2} (nc), mu_{T-1} (ncT) | nu_lb_0.. (nub) | nu_ub_0.. (nub)
 *   | rho_0..rho_{T-1} (nq), rho_T (nqT) | sigma_0..sigma_{T-1} (nr)
 *   with rho_t = 2 Q x_t, rho_T = 2 Q_T x_T, sigma_t = 2 R u_t, zeros for an infeasible node. */
typedef struct hmpc_result {
    double *obj;      /* B : primal objective, +inf if infeasible (bounded_qp.py:292-311)      */
    double *dual_obj; /* B : dual objective / Farkas objective   (bounded_qp.py:313-332)      */
    int32_t *status;  /* B                                                                    */
    int32_t *iters;   /* B : interior-point iterations spent on the node (low 16 bits);       */
                      /*     bit 16 (HMPC_ITERS_POLISHED): the record is the polished vertex    */
                      /*     solution (exactly complementary), not the interior-point iterate.  */
                      /*     An OPTIMAL record WITHOUT the bit is the iterate that met the      */
                      /*     stopping test (gap <= 1e-8, or 1e-6 at the floor of the barrier    */
                      /*     parameter): optimal to that tolerance and carrying its own KKT     */
                      /*     certificate, but reproducible across arithmetic orders only to     */
                      /*     ~1e-4 in the trajectory.  Every optimal node of the cart-pole      */
                      /*     systems polishes; 2 % of BASELINE configs[4]'s do not;              */
                      /*     bit 17 (HMPC_ITERS_WEAK): infeasible by about the accuracy of the  */
                      /*     arithmetic -- the embedding's tau collapsed, the node has no point */
                      /*     feasible to tolerance and is pruned, but its ray misses the proof  */
                      /*     tolerance (tol_inf): it must not be carried to the next MPC step   */
                      /*     (hmpc_fleet_* and the Python drivers reopen such a leaf at a shift) */
    double *primal;   /* B x n_primal                                                         */
    double *dual;     /* B x n_dual                                                           */
} hmpc_result;

/* Parent -> child hand-down (nullable everywhere).  The reference hands the parent node's simplex basis to the child
 * (controller.py:260-264, 426; subproblem_solution.py:37-43, with Gurobi's dual simplex); here the child receives the
 * parent's RECORD: node b tries the active set of row index[b] of (primal, dual) -- rows in the layout of hmpc_result,
 * e.g. the output arrays of an earlier call; index[b] < 0: nothing handed down -- before its first interior-point
 * iteration, multipliers and proximal centre from the parent.  A child whose optimum lies on the parent's active set
 * (the branch that fixes a binary where the relaxation had it) then costs one factorisation and a few solves instead
 * of ~11 iterations, and returns with HMPC_ITERS_HANDED set; where the set does not verify the node is solved from
 * the cold start exactly as without the hand-down.  Only records of OPTIMAL nodes flagged HMPC_ITERS_POLISHED may be
 * handed down (their multipliers are exactly complementary), and only to nodes with the same initial state.  States,
 * cost and the inputs the cost sees do not depend on the hand-down; where multipliers (dependent active rows) or
 * cost-free inputs are not unique a handed-down solve may return another optimal choice than a cold one.
 * Host-pointer form: host arrays of `rows` rows; device-pointer form: device arrays (`rows` unused), and the rows must
 * not be output rows of the same call. */
typedef struct hmpc_warm {
    const double *primal; /* rows x n_primal */
    const double *dual;   /* rows x n_dual   */
    const int32_t *index; /* B : row of node b's parent, or -1 */
    int32_t rows;
} hmpc_warm;

typedef struct hmpc_handle hmpc_handle;

/* Copies the problem to the device, precomputes scalings and sparse row/column lists.
 * options may be NULL (defaults).  The handle is bound to one device and is not thread-safe; it owns one set of
 * device workspaces: at most ONE launch per handle may be in flight (the device-pointer forms are asynchronous --
 * synchronise the stream, or use one handle per stream, before launching on the same handle again). */
int hmpc_create(const hmpc_problem *problem, const hmpc_options *options, hmpc_handle **out);
int hmpc_destroy(hmpc_handle *h);

/* Row lengths of hmpc_result.primal / .dual for this problem. */
int hmpc_record_sizes(const hmpc_handle *h, int32_t *n_primal, int32_t *n_dual);

/* Solves B nodes.  fix: B x (T*nub) int8, entry (t*nub+i) = -1 free, 0 or 1 fixed
 * (the identifier dictionaries of controller.py:273-298 flattened).  x0: nx doubles shared by
 * all nodes when x0_stride == 0, else B rows of stride x0_stride.
 * Host-pointer form: copies in, runs, copies out, returns when done.  Replaces
 * B calls of controller.py:229-271. */
int hmpc_solve_batch(hmpc_handle *h, const double *x0, int32_t x0_stride, const int8_t *fix, int32_t B,
                     const hmpc_warm *warm /* nullable */, const hmpc_result *out);

/* Device-pointer form: every pointer (x0, fix, and the members of out) is device memory on the
 * handle's device; the launch is asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int hmpc_solve_batch_device(hmpc_handle *h, const double *d_x0, int32_t x0_stride, const int8_t *d_fix, int32_t B,
                            const hmpc_warm *d_warm /* nullable */, const hmpc_result *d_out, void *stream);

/* ---- Warm-start node shift (reference: controller.py:431-564 construct_warm_start, :615-721) --------
 * After an MPC step the leaves of the branch-and-bound tree become the initial cover of the next
 * step's tree: leaves that disagree with the applied binaries are dropped (controller.py:566-613),
 * identifiers and multipliers move one stage towards the present, the last stage is filled through
 * two precomputed maps (:635-666), and the dual objective -- the leaf's new lower bound -- changes
 * by the pi-sum of :668-721, then by the model error and the clipping / reopening rule of :541-558.
 * One wavefront per leaf; the kernel is memory bound (it reads and writes one dual row per leaf). */
typedef struct hmpc_shift_maps {
    const double *M_mu;  /* nc x ncT : mu'_{T-2}  = M_mu  mu_{T-1}   (controller.py:186-227 _update_mu) */
    const double *M_rho; /* nq x nqT : rho'_{T-1} = M_rho rho_T                                        */
    const double *V;     /* nub x nu : the binaries of an input vector (MLDSystem.V)                    */
} hmpc_shift_maps;
int hmpc_set_shift_maps(hmpc_handle *h, const hmpc_shift_maps *maps);

/* Shifts B leaves that belong to K trees (MPC instances advanced together).  owner[b] in [0, K) is the
 * tree of leaf b; x0, u0, e0 hold K rows: the state the step started from, the applied input (uc, ub)
 * and the model error of the step.  Per leaf in: fix (T*nub int8), lower bound (+inf: proved infeasible),
 * dual row, dual objective.  Per leaf out: shifted fix / dual row / dual objective, the new lower bound,
 * and flags: bit 0 = keep (the leaf agrees with the applied binaries; otherwise the other outputs are
 * undefined), bit 1 = reopened (its infeasibility proof did not survive the shift: lb = 0, multipliers
 * to be ignored).  Host-pointer form: copies in, runs, copies out. */
int hmpc_shift_batch(hmpc_handle *h, int32_t B, int32_t K, const int32_t *owner, const double *x0, const double *u0,
                     const double *e0, const int8_t *fix, const double *lb, const double *dual, const double *dual_obj,
                     int8_t *fix_out, double *lb_out, double *dual_out, double *dual_obj_out, uint8_t *flags);
/* Device-pointer form, asynchronous on `stream`. */
int hmpc_shift_batch_device(hmpc_handle *h, int32_t B, int32_t K, const int32_t *d_owner, const double *d_x0,
                            const double *d_u0, const double *d_e0, const int8_t *d_fix, const double *d_lb,
                            const double *d_dual, const double *d_dual_obj, int8_t *d_fix_out, double *d_lb_out,
                            double *d_dual_out, double *d_dual_obj_out, uint8_t *d_flags, void *stream);

/* ---- Closed loops in lockstep ("fleet") ---------------------------------------------------------------
 * K independent closed loops of the controller advanced together -- the shape of the reference's Monte-Carlo
 * study (notebooks/cart_pole_with_walls/statistical_analysis.py:93-196: per step one warm-started branch and
 * bound, branch_and_bound.py:408-499 + controller.py:395-429, then construct_warm_start, controller.py:431-564).
 * The trees live behind the handle: topology and bounds on the host, every multiplier row in HBM (written by
 * the QP kernel, referenced by index, shifted in place by the shift kernel); one call per MPC step, the rounds
 * of all trees share kernel launches.  Needs hmpc_set_shift_maps.  Per loop and step:
 *     hmpc_fleet_solve : MIQP from x0[k] by branch and bound started from the loop's current tree (its root for a
 *                        cold loop), `width` candidates per tree and round (1 = the reference's node order);
 *                        speculation = k > 0: descendants through the next k binaries ride in the launch of a
 *                        node and are consumed only if the search gets there (same result, fewer launches);
 *                        speculation < 0: dive prediction -- with a node whose parent's record is at hand, the rest of
 *                        the dive predicted from the parent's rounded relaxed binaries and the sibling of every step
 *                        ride along (2 (T nub - depth) nodes; for few loops: a cold start in a handful of launches);
 *                        cost[k] (+inf: infeasible, the loop stops), u0[k] (nu: applied input), x1[k] (nx: the
 *                        model's next state A x0 + B u0), solves[k], leaves[k]
 *     hmpc_fleet_shift : the tree becomes the warm start of the next step given the model error e0[k] of the step
 *                        (next state = x1 + e0); cover[k] nodes, reopened[k] of them lost their infeasibility proof
 * Any output pointer may be NULL.  hmpc_fleet_reset(f, k) makes loop k (-1: all) cold again.  A call that fails midway
 * (device error, a consumed node that did not converge) leaves some trees advanced and others not: the fleet then refuses
 * further steps until hmpc_fleet_reset(f, -1). */
typedef struct hmpc_fleet hmpc_fleet;
int hmpc_fleet_create(hmpc_handle *h, int32_t K, hmpc_fleet **out);
int hmpc_fleet_destroy(hmpc_fleet *f);
int hmpc_fleet_reset(hmpc_fleet *f, int32_t k);
/* Loop k has ended: its tree is dropped, it takes no part in further steps (no launches, no pool rows) until it is reset. */
int hmpc_fleet_stop(hmpc_fleet *f, int32_t k);
/* Rows of the multiplier pools in use / allocated.  Rows nobody references are reclaimed: hmpc_fleet_shift compacts, and
 * hmpc_fleet_solve starts the pools from zero when every tree is cold. */
int hmpc_fleet_rows(const hmpc_fleet *f, int64_t *used, int64_t *capacity);
int hmpc_fleet_solve(hmpc_fleet *f, const double *x0 /* K x nx */, int32_t width, int32_t speculation, double tol, double *cost,
                     double *u0, double *x1, int32_t *solves, int32_t *leaves);
int hmpc_fleet_shift(hmpc_fleet *f, const double *e0 /* K x nx */, int32_t *cover, int32_t *reopened);
/* kernel launches (rounds) and nodes sent to the QP kernel since creation */
int hmpc_fleet_stats(const hmpc_fleet *f, int64_t *rounds, int64_t *launched);
/* Nodes the fleet's searches pruned WITHOUT a certificate (HMPC_ITERS_UNCERTIFIED), and how many searches ended with such a
 * node whose bound before its solve lay below the final incumbent (or with no incumbent): there the returned optimum rests on
 * that prune.  Since creation. */
int hmpc_fleet_uncertified(const hmpc_fleet *f, int64_t *pruned, int64_t *searches_resting_on_one);
/* Parent -> child hand-down inside the fleet's searches (hmpc_warm; on by default): a child that is solved in a later
 * round than its parent receives the parent's record, which already lies in the fleet's HBM pools.  enable: 1 / 0, < 0:
 * leave as is.  verified (nullable): solves since creation whose handed-down active set verified. */
int hmpc_fleet_handdown(hmpc_fleet *f, int32_t enable, int64_t *verified);
/* Host wall time of the fleet's calls by phase since creation (5 doubles, seconds): candidate selection, staging of the
 * rounds' nodes, device (copies, kernel, synchronisation), consumption of the results, node shifts. */
int hmpc_fleet_timing(const hmpc_fleet *f, double *seconds5);

/* ---- Incumbent exchange between the GPUs of a node (RCCL over xGMI) ----------------------------------
 * A frontier is sharded by node (node k to rank k mod nranks; nodes are independent, no data-path collective).
 * Once per branch-and-bound round every rank calls hmpc_allreduce_incumbent with its best upper bound and its
 * number of open candidates; on return *ub is the global best (every rank prunes against it) and *open the
 * largest count on any rank (zero exactly when all ranks are done: they stop in the same round).  One all-reduce
 * (MIN) of two float64.  Setup as for any RCCL communicator: rank 0 calls hmpc_comm_unique_id and hands the
 * 128 bytes to the other ranks (the caller's transport), then every rank calls hmpc_comm_create.  RCCL is
 * loaded at run time; a process that never creates a communicator does not need it.
 * A rank whose own search failed still owes the others the round they are waiting in: it calls once more with
 * *ub = -INFINITY, and every rank that receives -INFINITY stops with an error of its own (the convention of
 * warm_start_hmpc_amd/distributed.py: IncumbentExchange.abort / PeerFailure). */
typedef struct hmpc_comm hmpc_comm;
int hmpc_comm_unique_id(void *id128 /* 128 bytes out */);
int hmpc_comm_create(hmpc_handle *h, int32_t nranks, int32_t rank, const void *id128, hmpc_comm **out);
int hmpc_allreduce_incumbent(hmpc_comm *c, double *ub /* in/out */, int32_t *open /* in/out */);
/* The same exchange on device memory and on the caller's stream: pair_device[0] = upper bound, pair_device[1] =
 * -(open candidates), reduced in place (MIN); enqueued, not waited for -- work launched on `stream` afterwards sees the
 * global pair.  `stream` is a hipStream_t (NULL: the default stream). */
int hmpc_allreduce_incumbent_device(hmpc_comm *c, double *pair_device /* 2 doubles, in/out */, void *stream);
/* After the last round (once per search; every rank calls it): which rank owns the global incumbent, and its binary
 * assignment on every rank.  In: *ub this rank's best upper bound (+INFINITY: none), assignment its incumbent's nbytes
 * bytes (T*nub binaries for this solver; ignored on ranks that do not own the winner).  Out: *ub the global best, *owner
 * the lowest rank that holds it (-1: no rank has an incumbent, the MIQP is infeasible; assignment untouched), assignment
 * the owner's bytes on every rank (one ncclBroadcast).  SURVEY.md 8(b)/(e).
 * nbytes <= T*nub of the handle the communicator was created on.  A bound of -INFINITY or NaN on any rank (the abort
 * convention above) makes EVERY rank return HMPC_EINVAL after the first reduction; any other failure between the
 * collectives aborts the communicator (the peers are already waiting in the next one). */
int hmpc_publish_incumbent(hmpc_comm *c, double *ub /* in/out */, int8_t *assignment /* in/out */, int32_t nbytes, int32_t *owner);
int hmpc_comm_destroy(hmpc_comm *c);

/* ---- batched dense LPs of the offline terminal ingredients ----
 *
 *     maximise c_k'x   subject to   A x <= b_k (+ 1 on row relax[k]),   x in R^n free,   k = 0 .. B-1
 *
 * One launch replaces one sweep of the reference's one-Gurobi-LP-at-a-time loops:
 *   warm_start_hmpc/mcais.py:103-118       A = D_inf, c_k = (D A^t)_k, b = e_inf shared          (one sweep per horizon t)
 *   warm_start_hmpc/mcais.py:169-182       A = E, c_k = E_k, b = f shared, relax[k] = k         (redundant facets)
 *   warm_start_hmpc/controller.py:205-226  A = [F G], c_k = row k of [F_Tm1 G_Tm1], b = h shared; column k of the
 *                                          multiplier map M is z[k] (the reference states the dual LP
 *                                          min h'mu s.t. [F G]'mu = c_k, mu >= 0; HMPC_UNBOUNDED here is its "infeasible")
 * Host arrays, row-major: A[m][n]; c[B][n] with c_stride = n, or one c[n] with c_stride = 0; b likewise with b_stride =
 * m or 0; relax[B] or NULL (entries -1: none).  device < 0: the current device.  tol <= 0, max_iter <= 0: defaults
 * (1e-9, 100).  Out: obj[B] = c_k'x (NaN unless optimal), x[B][n], z[B][m] >= 0 or NULL (A'z = c_k at an optimum; a
 * Farkas ray A'z = 0, b'z < 0 when the set is empty), status[B] in {HMPC_OPTIMAL, HMPC_INFEASIBLE (empty set),
 * HMPC_UNBOUNDED, HMPC_MAXITER, HMPC_NUMERICAL}, iters[B].  Limits: n <= 64; the row vectors of one LP must fit one
 * CU's LDS (m <~ 1800).  Values of an optimum are exact to rounding (the interior-point iterate is moved to the
 * vertex), so the reference's comparisons with 0 (mcais.py:128) and 1e-7 (mcais.py:181) see what a simplex code shows. */
int hmpc_lp_solve_batch(int32_t device, int32_t n, int32_t m, const double *A, const double *c, int32_t c_stride,
                        const double *b, int32_t b_stride, const int32_t *relax, int32_t B, double tol, int32_t max_iter,
                        double *obj, double *x, double *z, int32_t *status, int32_t *iters);

/* Kernels compiled with the problem's sizes.  The reference takes any MLDSystem at one speed (warm_start_hmpc/controller.py:58-117).
 * hmpc_create compiles the kernel of each wave count (1 / 2 / 4 waves per node) ONCE MORE for the problem it is given, from the
 * sources next to the library, with the offline compiler (a child process), its integer sizes as constants of the translation
 * unit, into an on-disk cache (csrc/hmpc_jit.h: environment HMPC_JIT / HMPC_JIT_SIZED / HMPC_JIT_CACHE / HMPC_HIPCC): the
 * register kernel (static row map: rows and recursions in registers) where the problem admits it -- nx + nu <= 16, every
 * [F G] row with at most two input coefficients, columns of at most 16 entries, at most 128 Gram entries with terms, at least
 * one binary --, the run-time-sized kernel or, beyond one CU's LDS, its streaming form elsewhere.  Same source, same code
 * paths, same feature set as the shipped kernels, which serve wherever the compilation is not possible (no compiler, no
 * sources, HMPC_JIT=0 / HMPC_JIT_SIZED=0): the built-in register kernels of the two cart-pole shapes, the run-time-sized kernel
 * for every other system.
 * Two nets around code nobody has run before (the reference never hands back an undecided node, bounded_qp.py:216-228):
 *   first-use check : the first launch through a wave count solves 64 nodes -- spread over its batch, plus the root relaxation
 *                      and the deepest node -- with the compiled and the shipped kernel and compares statuses and objectives;
 *                      a kernel that disagrees is dropped for the handle (message on stderr).  One stream synchronisation per
 *                      wave count, inside that solve call -- or ahead of time through hmpc_validate_kernels (callers that
 *                      capture their stream or must not block).  HMPC_JIT_SELFCHECK=0 switches both nets off.
 *   second opinion  : in EVERY solve call (hmpc_solve_batch, hmpc_solve_batch_device, hmpc_fleet_solve) the nodes a compiled
 *                      kernel leaves MAXITER / NUMERICAL are listed on the device and solved again by the shipped kernel in
 *                      the same stream, without synchronisation; its records replace theirs.  A compiled kernel that leaves
 *                      nodes undecided which the shipped kernel decides is dropped when the counts arrive (the next call, or
 *                      hmpc_second_opinion_review after the caller has synchronised).
 *   hmpc_kernel_info : which kernel serves the problem for 1 / 2 / 4 waves per node: 0 run-time-sized, 1 its streaming
 *                      form, 2 built-in register kernel, 4 / 5 / 6 the run-time-sized kernel / its streaming form / the register
 *                      kernel compiled with this problem's sizes.
 *   hmpc_kernel_recipe : 1 per wave count whose compiled kernel was built with the compiler's ILP schedule (8 - 15 % faster; only
 *                      binaries listed in the cache's VALIDATED manifest, which tests/gpu_validate_ilp.py writes after running each of
 *                      them against the oracle: that schedule produced most of the wrong binaries of rounds 4 and 5), 0: the
 *                      compiler's default schedule -- the recipe of every kernel compiled for a problem nobody has validated.
 *   hmpc_jit_stats   : compiled kernels dropped by the nets; solve calls with a second opinion; batches on which it agreed.
 *   hmpc_jit_build_problem : everything hmpc_create would compile for this problem, ahead of time and without a GPU (the host
 *                      side of hmpc_create, nothing uploaded); paths: the shared objects, newline separated (may be NULL). */
int hmpc_kernel_info(const hmpc_handle *h, int32_t *kind3);
int hmpc_kernel_recipe(const hmpc_handle *h, int32_t *ilp3);
int hmpc_jit_stats(const hmpc_handle *h, int32_t *dropped, int32_t *second_runs, int32_t *second_agreed);
int hmpc_jit_build_problem(const hmpc_problem *problem, const hmpc_options *options, char *paths, int32_t paths_len);
int hmpc_validate_kernels(hmpc_handle *h, const double *d_x0, int32_t x0_stride, const int8_t *d_fix, int32_t B, void *stream);
int hmpc_second_opinion_review(hmpc_handle *h);

/* Number of workgroups the last launch used, and LDS bytes per workgroup (for reports). */
int hmpc_launch_info(const hmpc_handle *h, int32_t *grid, int32_t *lds_bytes);

/* Text of the last error on this thread ("" if none). */
const char *hmpc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HMPC_H */
