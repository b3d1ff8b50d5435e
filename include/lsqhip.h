/*
 * lsqhip.h -- C ABI of the MI355X (gfx950) nonlinear-least-squares hot path.
 *
 * This is the drop-in boundary for LeastSquaresOptim.jl's linear-algebra inner loop.  The reference
 * has no FFI layer: its plug points are Julia multiple dispatch on AbstractAllocatedSolver and a
 * duck-typed operator interface.  Every entry point below names the reference interface it
 * replaces (file:line relative to the reference repo); INTEGRATION.md shows the `ccall` shim a
 * maintainer would add on the Julia side.
 *
 * Conventions: plain pointers and sizes only; `double*` arguments named d_* are DEVICE pointers
 * (fp64), h_* are host pointers; matrices are column-major; indices are 0-based int32; every call
 * returns an lsq_status (0 = ok) and never aborts; lsq_last_error() gives the message.
 * All work is enqueued on the context's HIP stream; calls that return host scalars synchronise it.
 */
#ifndef LSQHIP_H
#define LSQHIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LSQ_OK = 0,
    LSQ_EDIM = 1,       /* DimensionMismatch / ArgumentError (types.jl:14-15, dense_qr.jl:10,61) */
    LSQ_ENOTPD = 2,     /* PosDefException from cholesky! (dense_cholesky.jl:57) */
    LSQ_ERANK = 3,      /* RankDeficientException from pivoted cholesky! (dense_cholesky.jl:33) */
    LSQ_ENONFINITE = 4, /* IsFiniteException (utils.jl:63-75) */
    LSQ_EBOUNDS = 5,    /* "Initial guess must be within bounds" (levenberg_marquardt.jl:51) */
    LSQ_EHIP = 6,       /* a HIP runtime call failed */
    LSQ_EARG = 7,       /* invalid argument (e.g. QR on a sparse Jacobian, types.jl:115-117) */
    LSQ_ECALLBACK = 8,  /* a user callback reported failure */
    LSQ_ERCCL = 9       /* sharded run: the exchange reported that a peer rank left its loop with an error */
} lsq_status;

typedef struct lsq_ctx lsq_ctx;       /* one per device/stream; not thread-safe */
typedef struct lsq_mat lsq_mat;       /* Jacobian: dense column-major or CSC (+CSR mirror) */
typedef struct lsq_solver lsq_solver; /* AbstractAllocatedSolver (types.jl:138-139) */
typedef struct lsq_model lsq_model;   /* built-in device-side f!/g! (synthetic benchmarks) */

typedef enum { LSQ_QR = 0, LSQ_CHOLESKY = 1, LSQ_LSMR = 2 } lsq_solver_kind;       /* types.jl:79-86 */
typedef enum { LSQ_DOGLEG = 0, LSQ_LEVENBERG_MARQUARDT = 1 } lsq_optimizer_kind;   /* types.jl:90-98 */

const char *lsq_last_error(void);
int lsq_version(void);

/* ---- context & raw device memory (Julia GC owns nothing here; explicit create/destroy) ---- */
int lsq_ctx_create(int device, void *hip_stream_or_null, lsq_ctx **out);
int lsq_ctx_destroy(lsq_ctx *ctx);
int lsq_ctx_sync(lsq_ctx *ctx);
void *lsq_ctx_stream(lsq_ctx *ctx);
int lsq_malloc(lsq_ctx *ctx, size_t bytes, void **d_out);
int lsq_free(lsq_ctx *ctx, void *d_ptr);
int lsq_h2d(lsq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int lsq_d2h(lsq_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int lsq_d2d(lsq_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* page-locked host memory: what a host-side g! should write the Jacobian values into, so that the upload after every
 * g!(J, x) (levenberg_marquardt.jl:77-81: 80 MB at C4) runs at the PCIe rate, asynchronously (lsq_mat_set_values_async) */
int lsq_host_alloc(lsq_ctx *ctx, size_t bytes, void **h_out);
int lsq_host_free(lsq_ctx *ctx, void *h_ptr);

/* ---- Jacobian handles ---- */
/* Dense m x n, column-major, device-owned (the `J::Matrix` of types.jl:36). */
int lsq_dense_create(lsq_ctx *ctx, int m, int n, lsq_mat **out);
/* CSC pattern given once (SparseMatrixCSC with a FIXED pattern, as the reference's sparse g!
 * requires: test/nonlinearleastsquares.jl:47-86).  Builds the CSR mirror + CSC->CSR value map. */
int lsq_csc_create(lsq_ctx *ctx, int m, int n, const int *h_colptr, const int *h_rowval, lsq_mat **out);
int lsq_mat_destroy(lsq_mat *J);
int lsq_mat_size(const lsq_mat *J, int *m, int *n, long long *nnz);
/* Upload values after a host-side g!(J, x): dense m*n column-major, or nzval in CSC order. */
int lsq_mat_set_values(lsq_mat *J, const double *h_values);
int lsq_mat_get_values(const lsq_mat *J, double *h_values);
/* The same upload without blocking the host: the copy is queued on the context's copy stream (the compute stream keeps
 * running whatever was queued before) and every later use of J on the compute stream waits for it on the DEVICE.
 * h_values must be page-locked (lsq_host_alloc) and must not be written again before lsq_mat_upload_wait(J) returns
 * (or the next call that reads results back has returned).  Pageable memory is accepted and falls back to the
 * synchronous path. */
int lsq_mat_set_values_async(lsq_mat *J, const double *h_pinned_values);
int lsq_mat_upload_wait(lsq_mat *J);
/* Device pointer to the values a device-side g! writes (dense buffer / CSC nzval) ... */
double *lsq_mat_values(lsq_mat *J);
/* ... after which the CSR mirror must be refreshed (no-op for dense). */
int lsq_mat_refresh(lsq_mat *J);

/* ---- column-scaled Jacobians J = V diag(s) ----
 * Models of the form r(x) = V phi(x) - b (phi elementwise) have J(x) = V diag(phi'(x)): the pattern AND the stored values V
 * are fixed, only n factors change from one g!(J, x) to the next (test/nonlinearleastsquares.jl:47-86 is the general case,
 * where g! rewrites nonzeros(J); this is the structured special case).  lsq_mat_set_colscale(J, d_s) declares the values J
 * holds at that moment (lsq_mat_set_values / lsq_mat_values) to be V and d_s (n device doubles, owned by the caller, read
 * whenever J is used) to be s; every operation on the handle -- lsq_mul, lsq_colsumabs2, lsq_rowsumabs2, lsq_ldiv*,
 * lsq_optimize -- then acts on V diag(s).  On the sliced layouts of big sparse patterns nothing is multiplied out: J*x
 * gathers s.*x, J'y = s.*(V'y), colsumabs2(J) = s.^2 .* colsumabs2(V) (cached): a g! costs an n-vector instead of two passes
 * over nnz values.  Other matrices (small, dense, segment-kernel patterns) keep V aside and multiply the values out each
 * time.  A g! that changed s calls lsq_mat_colscale_changed(J) (instead of lsq_mat_refresh); lsq_mat_set_values /
 * lsq_mat_values / lsq_mat_get_values keep addressing V.  d_s = NULL turns J back into the plain matrix V.
 * Entry (i,j) of J is used as V_ij * s_j formed on the fly where the reference would have stored fl(V_ij s_j): results agree
 * with a multiplied-out J to a few ulp per product, not bit for bit (tests/test_a_gpu_contract.py::test_column_scaled_jacobian). */
int lsq_mat_set_colscale(lsq_mat *J, const double *d_s);
int lsq_mat_colscale_changed(lsq_mat *J);

/* ---- custom Jacobian operators (README.md:37-47: any type with mul!, mul! of the adjoint, colsumabs2!,
 *      size, eltype works with LSMR) ----
 * A matrix-free handle: `mul(trans, d_x, d_out, user)` must write J*x (trans = 0, m entries) or J'*x
 * (trans = 1, n entries) into d_out; `colsumabs2(d_out, user)` the n column sums of squares.  Both
 * are HOST callbacks operating on device pointers; the library drains its stream before calling and
 * the callback must have finished its device work when it returns (slow path, like any host-defined
 * operator in the reference).  All fusions of the solvers still apply: the library runs its epilogue
 * over the vector the callback produced.  Works with LSMR (ldiv, lsq_optimize); QR / Cholesky need
 * the entries and refuse it. */
typedef int (*lsq_op_mul_callback)(int trans, const double *d_x, double *d_out, void *user);
typedef int (*lsq_op_colsum_callback)(double *d_out, void *user);
int lsq_op_create(lsq_ctx *ctx, int m, int n, lsq_op_mul_callback mul, lsq_op_colsum_callback colsumabs2,
                  void *user, lsq_mat **out);

/* ---- operator interface (README.md:37-47; used at lsmr.jl:73,76,118,122 and by the optimizers) ---- */
/* mul!(y, J, x, alpha, beta) / mul!(x, J', y, alpha, beta): trans = 0 / 1.
 * Replaces SparseArrays/BLAS mul! at levenberg_marquardt.jl:102,114; dogleg.jl:99,109,171;
 * iterative_lsmr.jl:32,40,91,106. */
int lsq_mul(lsq_mat *J, int trans, double alpha, const double *d_x, double beta, double *d_y);
/* colsumabs2!(out, J): utils.jl:139-151 */
int lsq_colsumabs2(lsq_mat *J, double *d_out);
/* rowsumabs2!(out, J) = colsumabs2!(out, J') for adjoint Jacobians: utils.jl:153-161 (out has m entries) */
int lsq_rowsumabs2(lsq_mat *J, double *d_out);

/* BLAS-1 on device vectors (what lsmr.jl:30-44 and the optimizer loops need from a vector type) */
int lsq_axpy(lsq_ctx *ctx, int n, double a, const double *d_x, double *d_y);       /* axpy!   */
int lsq_scal(lsq_ctx *ctx, int n, double a, double *d_x);                          /* rmul!   */
int lsq_copy(lsq_ctx *ctx, int n, const double *d_x, double *d_y);                 /* copyto! */
int lsq_fill(lsq_ctx *ctx, int n, double a, double *d_x);                          /* fill!   */
int lsq_sumsq(lsq_ctx *ctx, int n, const double *d_x, double *h_out);              /* sum(abs2, x) */
int lsq_sum(lsq_ctx *ctx, int n, const double *d_x, double *h_out);                /* sum(x)  */
int lsq_dot(lsq_ctx *ctx, int n, const double *d_x, const double *d_y, double *h_out);   /* dot(x, y) */
int lsq_emul(lsq_ctx *ctx, int n, const double *d_x, const double *d_y, double *d_out);  /* map!(*, out, x, y): iterative_lsmr.jl:121 */
int lsq_nrm2(lsq_ctx *ctx, int n, const double *d_x, double *h_out);               /* norm(x) */
int lsq_wdot(lsq_ctx *ctx, int n, const double *d_x, const double *d_y, const double *d_w,
             double *h_out);                                                       /* utils.jl:165-175 */
int lsq_amax(lsq_ctx *ctx, int n, const double *d_x, double *h_out);               /* maximum(abs, x) */
/* maxabs_projected_gradient (utils.jl:39-55); d_lower / d_upper may be NULL */
int lsq_amax_projected(lsq_ctx *ctx, int n, const double *d_g, const double *d_x,
                       const double *d_lower, const double *d_upper, double *h_out);
int lsq_clamp(lsq_ctx *ctx, int n, double lo, double hi, double *d_x);             /* clamp!  */
int lsq_ediv(lsq_ctx *ctx, int n, const double *d_x, const double *d_y, double *d_out); /* map!(/, ..) */
/* box step clipping dx = min(dx, x-lower), dx = max(dx, x-upper) (levenberg_marquardt.jl:89-98) */
int lsq_box_clip(lsq_ctx *ctx, int n, double *d_dx, const double *d_x, const double *d_lower,
                 const double *d_upper);
/* first non-finite index or -1 (check_isfinite, utils.jl:70-75) */
int lsq_first_nonfinite(lsq_ctx *ctx, int n, const double *d_x, int *h_index);

/* ---- linear least-squares solvers: THE plug point ---- */
/* AbstractAllocatedSolver(nls, optimizer): dense_qr.jl:25-28,50-54; dense_cholesky.jl:19-21;
 * iterative_lsmr.jl:173-177,233-236.  `for_lm` selects the damped (LevenbergMarquardt) flavour. */
int lsq_solver_create(lsq_ctx *ctx, lsq_mat *J, int solver_kind, int for_lm, lsq_solver **out);
int lsq_solver_destroy(lsq_solver *s);
/* ldiv!(x, J, y, A) -> (x, nmul)       Dogleg: dense_qr.jl:30, dense_cholesky.jl:29,
 *                                              iterative_lsmr.jl:179.  y is preserved. */
int lsq_ldiv(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_x, int *nmul);
/* ldiv!(x, J, y, damp, A) -> (x, nmul) LM: dense_qr.jl:56, dense_cholesky.jl:43,
 *                                          iterative_lsmr.jl:238.  damp MAY be clobbered (LSMR
 *                                          leaves sqrt(damp) in it, iterative_lsmr.jl:252). */
int lsq_ldiv_damped(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x,
                    int *nmul);
/* LSMR(preconditioner!, P) (types.jl:82-86; README.md:47; default iterative_lsmr.jl:129-141): replace the
 * built-in Jacobi preconditioner of an LSMR solver.  Before every solve the callback must fill d_P (n
 * entries) with the factors the preconditioner solve MULTIPLIES by (an InverseDiagonal stores the
 * inverse, iterative_lsmr.jl:117-122) for the operator J'J + diag(damp); d_damp is the un-rooted
 * damping, NULL for the undamped (Dogleg) solve.  Diagonal preconditioners only.  The callback runs
 * on the host after the library's stream has been drained (the reference's slow path too); the
 * reference-order small-problem path is not used with a custom preconditioner.  NULL restores the default. */
typedef int (*lsq_precond_callback)(double *d_P, lsq_mat *J, const double *d_damp, void *user);
int lsq_solver_set_preconditioner(lsq_solver *s, lsq_precond_callback cb, void *user);
/* LSMR(preconditioner!, P) with ANY P that supports ldiv! (types.jl:82-86; README.md:47: "The preconditioner can be any type
 * that supports A_ldiv_B!(x, P, y)") -- block-diagonal, incomplete factors, whatever the caller keeps behind `user`:
 *   update(J, d_damp, user) = preconditioner!(P, x, J, damp): refresh P for J'J + diag(damp) (d_damp un-rooted; NULL for the
 *                             undamped Dogleg solve); may be NULL if P never changes;
 *   ldiv(d_out, d_in, user) = ldiv!(out, P, in) on n-vectors in device memory.
 * Both are HOST callbacks on device pointers; the library drains its stream before each call and the callback must have
 * finished its device work when it returns.  A general P cannot be folded into the fused kernels, so the solve runs as the
 * reference's own structure at the operator level (lsmr.jl:53-238 on PreconditionedMatrix(DampenedMatrix(J, sqrt(damp)), P),
 * every vector operation a launch, scalars on the host: lsq_lsmr_general.hip) -- the slow path, as a user-supplied P is in the
 * reference.  Like the reference (iterative_lsmr.jl:36-51) the adjoint applies ldiv!(., P, .) again, i.e. P is taken to be
 * symmetric.  ldiv = NULL restores the built-in path. */
typedef int (*lsq_precond_update_callback)(lsq_mat *J, const double *d_damp, void *user);
typedef int (*lsq_precond_ldiv_callback)(double *d_out, const double *d_in, void *user);
int lsq_solver_set_general_preconditioner(lsq_solver *s, lsq_precond_update_callback update, lsq_precond_ldiv_callback ldiv,
                                          void *user);
/* Row-sharded SINGLE problem: J is split by residual rows, rank p holds the m_p x n block J_p and the
 * matching slices of the m-vectors; every n-vector is replicated and every rank runs the same scalar control flow.  What
 * crosses ranks is a SUM all-reduce of a device buffer, IN PLACE, ordered on the library's stream (the hook enqueues it
 * there -- e.g. ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, comm, hip_stream): RCCL over xGMI -- or makes that
 * stream wait for it; it must not need the host to wait).  Called per LSMR inner iteration with n + 1 doubles (J_p'u_p with
 * sum(u_p.^2) riding along: u is kept unnormalised, so the norm the reference needs BEFORE the adjoint product,
 * lsmr.jl:119-122, travels WITH it), and per outer iteration for colsumabs2 (n, only after g!), the gradient J'f (n) and
 * {trial ssr, predicted ssr} (2).  LevenbergMarquardt(LSMR()) only. */
typedef int (*lsq_device_allreduce_callback)(double *d_buf, int count, void *hip_stream, void *user);

/* row-sharded operator-level use (lsq_ldiv / lsq_ldiv_damped with LSMR on a row block J_p; y is the local slice, x the
 * replicated solution): installs the all-reduce hook of lsq_options.row_allreduce on this solver.  NULL removes it. */
int lsq_solver_set_row_allreduce(lsq_solver *s, lsq_device_allreduce_callback cb, void *user, long long global_rows);
/* diagnostics of the last solve: LSMR istop / iterations, QR numerical rank */
int lsq_solver_info(const lsq_solver *s, int *lsmr_iter, int *lsmr_istop, int *qr_rank);
/* which factorisation the last QR solve used (dense_qr.jl:37,83 always runs geqp3; this build only needs the
 * pivoted sweep when the rank decision is open):  0 none yet, 1 one-stage pivoted Householder,
 * 2 blocked unpivoted QR + pivoted sweep on R,  3 blocked unpivoted QR + full-rank certificate
 * (||R||_F ||inv(R)||_F * rcond * 16 <= 1 proves xGELSY's rank = n; no pivoting needed) */
int lsq_solver_qr_path(const lsq_solver *s, int *path);
/* how the 64-column panels of the last blocked QR were factored: 0 no blocked factorisation yet, 1 column-by-column
 * Householder steps (k_qr1_step_multi), 2 CholeskyQR2 + basis-kernel block reflector (lsq_qr_cholqr.hip; falls back to 1
 * for the whole solve when a panel is too ill-conditioned for it) */
int lsq_solver_qr_panel(const lsq_solver *s, int *kind);
/* the same for Cholesky() (dense_cholesky.jl:29-59):  0 none yet, 1 one-workgroup kernel (dpotf2 / pivoted dpstf2),
 * 2 blocked unpivoted factorisation (LM: J'J + damp),  3 blocked unpivoted factorisation + full-rank certificate
 * (Dogleg: 1 / ||inv(U)||_F^2 > 16 n eps max diag proves that cholesky!(.., Val(true)) would not stop early),
 * 4 like 2 with the whole factorisation in ONE launch (k_chol_tiles: one resident workgroup per 64 x 64 upper tile,
 * n <= 1408; repeated as 2 if one of its bounded waits gives up) */
int lsq_solver_chol_path(const lsq_solver *s, int *path);

/* The fast paths above that rely on co-resident workgroups (one-launch Cholesky, pipelined triangular solves, the QR panel's slab
 * exchange + pipelined certified solve) wait with a bound; a wait that gives up makes the solve repeat itself on the
 * launch-per-step path, is COUNTED, and pauses that fast path for a number of solves (16, then 64, ... up to 4096 while it
 * keeps failing right after being armed again; back to 16 after a clean run) instead of switching it off for good: a
 * neighbour on the device (an RCCL kernel of a sharded run) may be gone by then.  Index: 0 one-launch Cholesky (k_chol_tiles),
 * 1 pipelined triangular solves, 2 QR slab exchange / pipelined certified solve, 3 CholeskyQR2 panel breakdowns (numerical:
 * ill-conditioned panels -- counted and paused the same way).  h_giveups: totals of this solver; h_paused: solves left before
 * the path is tried again (0 = armed).  Either pointer may be NULL. */
int lsq_solver_stats(const lsq_solver *s, int h_giveups[4], int h_paused[4]);
/* the same totals over every solver the context has run (the solver lsq_optimize keeps inside the context included) */
int lsq_ctx_fallback_stats(const lsq_ctx *ctx, int h_giveups[4]);
/* What the context's device looks like to the launch heuristics: compute units (256 on an unpartitioned MI355X, 32 in CPX mode:
 * the slab exchanges of the QR panel need all 256 -- lsq_qr_stage1.hip), XCDs inferred from that (8 CUs x 4 per XCD -> num_cus / 32),
 * and the architecture name (at most name_cap - 1 characters).  Any pointer may be null. */
int lsq_ctx_device_info(const lsq_ctx *ctx, int *num_cus, int *num_xcds, char *arch_name, int name_cap);

/* LM + LSMR with a device-side f!: the kernels that follow the inner solve (step, predicted residual, trial residual) are
 * enqueued behind the inner iteration at which the PREVIOUS solve stopped and skip themselves if this one is not over by then
 * (DESIGN 4.2).  h_out = {solves that were given such a guess, guesses that were wrong}; LSQ_NO_TAIL_SPECULATION=1 turns it off. */
int lsq_ctx_tail_stats(const lsq_ctx *ctx, long long h_out[2]);

/* ---- whole trust-region loop on device buffers (host control, device arrays) ---- */
/* f!(out, x) and g!(J, x) on DEVICE pointers; g writes lsq_mat_values(J) (the library refreshes
 * the CSR mirror afterwards).  Return non-zero to abort with LSQ_ECALLBACK. */
typedef int (*lsq_f_callback)(double *d_out, const double *d_x, void *user);
typedef int (*lsq_g_callback)(lsq_mat *J, const double *d_x, void *user);
/* optional global reduction hook for sharded problems: called exactly once per outer
 * iteration with vals = {ssr_local, maxabs_gr_local, converged_local} (the values the rank held at
 * the top of the iteration); it overwrites them with the global {sum, max, min}.  NULL = single
 * problem.  A rank that reports converged_local = 1 is frozen: it is called at the top of the
 * iteration and leaves the loop when the returned min is 1.  A rank that is still iterating is
 * called LATER in the iteration, once device work has been queued (inside the LSMR driver when its
 * look-ahead window is full), so the exchange overlaps the device; it never acts on the result
 * ("all converged" cannot hold while it is not converged itself), which lets an implementation
 * return the previous exchange's values to such a rank and leave the new one in flight
 * (leastsquaresoptim.jl_amd/sharding.py does; bench.py carries it over the RCCL process group).
 * Error protocol: a rank that leaves its loop with an error calls the hook one last time with
 * converged_local = -1; an implementation reports "some rank has left" to the others either by returning 2
 * or by setting the returned third value to -1 -- the loop then returns LSQ_ERCCL instead of entering another
 * collective that the departed rank will never join. */
typedef int (*lsq_allreduce_callback)(double *h_vals, int count, void *user);

typedef struct {
    double x_tol, f_tol, g_tol; /* 1e-8 defaults of levenberg_marquardt.jl:41 / dogleg.jl:43 */
    int iterations;             /* 1000 */
    double delta;               /* <= 0: 10.0 for LM, 1.0 for Dogleg */
    const double *h_lower;      /* NULL or n (host) */
    const double *h_upper;      /* NULL or n (host) */
    lsq_allreduce_callback allreduce;
    void *allreduce_user;
    /* optional trace buffers (host), capacity trace_cap iterations */
    int trace_cap;
    double *trace_ssr, *trace_gnorm, *trace_delta, *trace_rho;
    int *trace_inner, *trace_accept;
    double *trace_x;            /* trace_cap * n, or NULL */
    lsq_precond_callback preconditioner;   /* LSMR only; NULL = default Jacobi */
    void *preconditioner_user;
    lsq_device_allreduce_callback row_allreduce;   /* row-sharded single problem (see above); NULL = J holds all rows */
    void *row_allreduce_user;
    long long global_rows;      /* row-sharded: sum of the ranks' row counts (lsmr.jl:55 maxiter = max(size(A)...)) */
    lsq_precond_update_callback precond_update;    /* LSMR(preconditioner!, P) with a general P (see */
    lsq_precond_ldiv_callback precond_ldiv;        /* lsq_solver_set_general_preconditioner); NULL = not used */
    void *precond_general_user;
} lsq_options;

typedef struct {
    int optimizer;              /* lsq_optimizer_kind actually used */
    double ssr;
    int iterations;
    int converged, x_converged, f_converged, g_converged;
    int f_calls, g_calls, mul_calls;
    int status;
    int bad_index;              /* for LSQ_ENONFINITE */
    double seconds;             /* wall time of the loop (host clock, includes syncs) */
    long long lsmr_iterations;  /* total inner iterations */
    double ssr0;                /* sum(abs2, f(x0)): state 0 of the reference's trace (levenberg_marquardt.jl:70) */
} lsq_result;

void lsq_options_default(lsq_options *opt);
/* optimize!(LeastSquaresProblemAllocated; kwargs...) -- levenberg_marquardt.jl:39-144,
 * dogleg.jl:41-203.  d_x (n) and d_fcur (m) are updated in place like nls.x / nls.y. */
int lsq_optimize(lsq_ctx *ctx, int optimizer, int solver_kind, lsq_mat *J, double *d_x,
                 double *d_fcur, lsq_f_callback f, lsq_g_callback g, void *user,
                 const lsq_options *opt, lsq_result *res);

/* ---- built-in device-side model for the synthetic benchmarks:
 *      r(x) = A tanh(x) - b,  J = A diag(1 - tanh(x)^2); A has J's pattern. ---- */
int lsq_model_tanh_create(lsq_ctx *ctx, lsq_mat *J, const double *h_Avalues, const double *h_b,
                          lsq_model **out);
int lsq_model_destroy(lsq_model *md);
lsq_f_callback lsq_model_f(void);
lsq_g_callback lsq_model_g(void);

/* ---- deterministic synthetic inputs (host side; counter-based RNG) ---- */
/* CSC with exactly `per_col` distinct sorted rows per column, values N(0,1)/sqrt(per_col). */
int lsq_synth_sparse(int m, int n, int per_col, unsigned long long seed, int *h_colptr,
                     int *h_rowval, double *h_nzval);
/* dense column-major N(0,1)/sqrt(m) */
int lsq_synth_dense(int m, int n, unsigned long long seed, double *h_values);
/* x_true ~ U(-1,1) (n) and noise ~ N(0,1) (m) streams */
int lsq_synth_uniform(int n, unsigned long long seed, double lo, double hi, double *h_out);
int lsq_synth_normal(int n, unsigned long long seed, double *h_out);

/* ---- measurement helpers ---- */
/* Times `reps` launches of y <- alpha*J*x + beta*y (trans=0) or the transpose (trans=1) with HIP
 * events on the context stream; returns the average milliseconds per launch. */
int lsq_bench_mul(lsq_mat *J, int trans, int reps, const double *d_x, double *d_y, double beta,
                  float *h_ms_per_launch);

/* A neighbour on the device, for tests and measurements of the paths above: `workgroups` workgroups of 256 threads, each
 * holding lds_bytes of LDS, spin for `milliseconds` on a stream of their own (returns at once; lsq_bench_occupy_wait joins). */
int lsq_bench_occupy(lsq_ctx *ctx, int workgroups, int lds_bytes, double milliseconds);
int lsq_bench_occupy_wait(lsq_ctx *ctx);

/* Reference-order arithmetic for small problems (m, n <= 2048, nnz <= 2^18): every sum is taken in
 * the order of the reference's serial loops, so iteration / mul counts reproduce the CPU
 * restatement exactly even where LSMR's stop iteration depends on the last bit.
 * on = 1 / 0 forces it on / off, -1 restores the default (on; env LSQ_EXACT=0 turns it off). */
int lsq_set_exact(int on);

/* Debug modes, process-wide (the hazard / serialised build the reference gets from Julia's --check-bounds and a
 * single thread; also environment LSQ_DEBUG_LAUNCH_JITTER=<us>, LSQ_DEBUG_SERIAL=<0|1|2>, read when the first context is made):
 *   launch_jitter_us > 0  random host stalls of up to that many microseconds (one in 64: 20x) in front of one kernel launch in
 *                         four -- any hand-off between kernels that holds only because the next launch follows at once breaks;
 *   serial = 1            every launch is waited for before the host goes on: no side-stream concurrency, no LSMR look-ahead,
 *                         no speculation -- SAME kernels and arithmetic, results must be bit-identical to the normal mode;
 *   serial = 2            additionally without the in-kernel workgroup exchanges (QR slab exchange, pipelined triangular
 *                         solves, one-launch Cholesky): the multi-launch fallbacks, equal to the fast paths to round-off.
 * A negative argument leaves that setting alone.  lsq_debug_get also reports the number of stalls injected so far. */
int lsq_debug_set(int launch_jitter_us, int serial);
int lsq_debug_get(int *launch_jitter_us, int *serial, long long *stalls);

/* HIP-event instrumentation of the two dominant kernels inside lsq_ldiv(_damped) with LSMR:
 * kernel 0 = K1 (u <- J t - cu u, the J*v product), kernel 1 = K2 (v <- J'u ..., the J'*u product).
 * Events are recorded on the context stream around each launch while enabled (up to max_samples
 * per kernel); lsq_prof_end waits for them and returns average milliseconds and sample counts. */
int lsq_prof_begin(lsq_ctx *ctx, int max_samples);
/* bit k of kernel_mask: instrument kernel k (default 3 = both).  A timed launch costs a few
 * microseconds of pipeline gaps, so bench.py times only kernel 0 inside the timed region. */
int lsq_prof_select(lsq_ctx *ctx, int kernel_mask);
int lsq_prof_end(lsq_ctx *ctx, double h_avg_ms[2], int h_count[2]);
/* average milliseconds between two HIP events recorded back to back with NOTHING in between: the
 * marker overhead contained in every bracketed interval above (for calibration). */
int lsq_prof_overhead(lsq_ctx *ctx, int pairs, double *h_ms);

#ifdef __cplusplus
}
#endif
#endif
