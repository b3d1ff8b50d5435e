/*
 * lsq_oracle.h -- CPU restatement of the LeastSquaresOptim.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the CPU reference that is timed beside the GPU.
 *
 * PARITY PINNING: the reference (Julia) cannot be run in this environment.  Its
 * own tests hold (a) the NIST StRD certified parameter values
 * (test/nonlinearfitting.jl) and (b) outcome pins (ssr <= 1e-3, converged,
 * |x-x*| <= 1e-6, default-selection strings) -- never a trajectory or an
 * iteration count.  This oracle is pinned against (i) the NIST certified
 * values (tests/golden/nist.json: hit wherever the problem's start allows, same
 * miss set as the HIP path), (ii) the outcome pins on the restated MINPACK /
 * factor-model / bounds problems, (iii) the hand-derived known-answer
 * trajectories of SURVEY.md 8(c) (KAT-DL, KAT-LM), (iv) scipy's LSMR and the
 * LAPACK routines Julia dispatches to (dgeqp3/dormqr/dgelsy/dpotrf/dpstrf).
 * Iteration counts depend on how Julia's stdlib associates its sums, which
 * cannot be observed here: orc_set_sum_mode models the candidates (index order,
 * mapreduce, 4/8/16 SIMD lanes, extended-precision nrm2, wave trees, random
 * orders, one-rounding perturbations) and tests/golden/count_stable.json records
 * which runs keep their counts under all of them.  Trajectory / iteration-count
 * parity with a live Julia run is UNPINNED.
 *
 * All indices are 0-based, matrices column-major, fp64.
 */
#ifndef LSQ_ORACLE_H
#define LSQ_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DENSE = 0, ORC_CSC = 1 };

typedef struct {
    int kind;          /* ORC_DENSE or ORC_CSC */
    int m, n;
    double *val;       /* dense: m*n column-major; CSC: nzval[nnz] */
    const int *colptr; /* CSC: n+1 */
    const int *rowval; /* CSC: nnz, sorted within a column */
} orc_mat;

/* solver / optimizer ids mirror types.jl:79-98 */
enum { ORC_QR = 0, ORC_CHOLESKY = 1, ORC_LSMR = 2 };
enum { ORC_DOGLEG = 0, ORC_LM = 1 };

/* status codes (the exceptions of the reference, SURVEY 8b "Errors") */
enum {
    ORC_OK = 0,
    ORC_EDIM = 1,        /* DimensionMismatch / ArgumentError */
    ORC_ENOTPD = 2,      /* PosDefException */
    ORC_ERANK = 3,       /* RankDeficientException */
    ORC_ENONFINITE = 4,  /* IsFiniteException */
    ORC_EBOUNDS = 5      /* "Initial guess must be within bounds" */
};

typedef void (*orc_f_cb)(double *out, const double *x, void *ud);
typedef void (*orc_g_cb)(double *Jval, const double *x, void *ud);

typedef struct {
    double x_tol, f_tol, g_tol;
    int iterations;
    double delta;            /* <= 0 : optimizer default (10 for LM, 1 for Dogleg) */
    const double *lower;     /* NULL or n */
    const double *upper;     /* NULL or n */
    /* LSMR parameter overrides (<0 => reference defaults) */
    int trace_cap;           /* capacity (iterations) of the trace buffers below */
    double *trace_ssr;       /* [trace_cap] ssr after iteration k  */
    double *trace_gnorm;     /* [trace_cap] maxabs_gr              */
    double *trace_delta;     /* [trace_cap] Delta after iteration  */
    double *trace_rho;       /* [trace_cap]                        */
    int *trace_inner;        /* [trace_cap] ls iterations (mul count from ldiv!) */
    int *trace_accept;       /* [trace_cap]                        */
    double *trace_x;         /* [trace_cap*n] x after iteration, or NULL */
} orc_options;

typedef struct {
    int optimizer;
    double ssr;
    int iterations;
    int converged, x_converged, f_converged, g_converged;
    int f_calls, g_calls, mul_calls;
    int status;
    int bad_index;           /* first non-finite index for ORC_ENONFINITE */
} orc_result;

/* summation-order model of the stdlib reductions (lsq_oracle.c header): 0 = index order (default), 1 = mapreduce
 * without SIMD, 2/3/4 = 4/8/16 SIMD lanes, 5 = 16 lanes + extended-precision nrm2.  Process-global. */
void orc_set_sum_mode(int mode);
int orc_get_sum_mode(void);

/* --- kernels (utils.jl, SparseArrays/BLAS semantics) --- */
void orc_colsumabs2(double *v, const orc_mat *A);
void orc_rowsumabs2(double *v, const orc_mat *A);
void orc_mul(double *y, const orc_mat *A, const double *x, double alpha, double beta);
void orc_mulT(double *x, const orc_mat *A, const double *y, double alpha, double beta);
double orc_wdot(const double *x, const double *y, const double *w, int n);
double orc_wnorm(const double *x, const double *w, int n);
double orc_maxabs_projected_gradient(const double *g, const double *x, const double *lower,
                                     const double *upper, int n);

/* --- linear least-squares solvers (the drop-in boundary) --- */
/* Each returns status; *nmul receives the reference's second return value. */
int orc_ldiv_lsmr(double *x, const orc_mat *J, const double *y, int *nmul);
int orc_ldiv_lsmr_damped(double *x, const orc_mat *J, const double *y, double *damp, int *nmul);
int orc_ldiv_cholesky(double *x, const orc_mat *J, const double *y, int *nmul);
int orc_ldiv_cholesky_damped(double *x, const orc_mat *J, const double *y, const double *damp,
                             int *nmul);
int orc_ldiv_qr(double *x, const orc_mat *J, const double *y, int *nmul, int *rank_out);
int orc_ldiv_qr_damped(double *x, const orc_mat *J, const double *y, const double *damp, int *nmul,
                       int *rank_out);

/* raw LSMR on the (optionally damped, optionally preconditioned) operator
 *   A = [J; diag(diag)] * diag(P)   (diag == NULL: undamped; P == NULL: identity)
 * b = (by (m), 0 (n)).  x must be zero on entry (callers always pass zeros).
 * Returns iteration count; *istop receives the stop rule. */
int orc_lsmr(double *x, const orc_mat *J, const double *diag, const double *P, double *by,
             double atol, double btol, double conlim, int maxiter, int *istop,
             double *normr_out, double *normAr_out);

/* --- pieces exposed for cross-checks against LAPACK --- */
int orc_potrf_upper(double *A, int n);                         /* 0 ok, k>0 not PD at k */
int orc_pstrf_upper(double *A, int n, int *piv, int *rank, double tol);
void orc_geqp3(double *A, int m, int n, int *jpvt, double *tau);
/* test hook: serve orc_geqp3 (and with it the QR solvers' factorisation) from LAPACK's dgeqp3 at sizes the scalar
 * restatement needs minutes for; NULL restores the restatement */
typedef void (*orc_geqp3_fn)(double *A, int m, int n, int *jpvt, double *tau);
void orc_set_geqp3_backend(orc_geqp3_fn fn);
int orc_qrp_solve(double *A, int m, int n, const int *jpvt, const double *tau, double *b, int lenb,
                  double rcond);                               /* returns rank; b[0:n] = solution */

/* --- optimizers --- */
int orc_optimize(int optimizer, int solver, orc_mat *J, double *x, double *fcur, orc_f_cb f,
                 orc_g_cb g, void *ud, const orc_options *opt, orc_result *res);

/* --- built-in synthetic model  r(x) = A tanh(x) - b,  J = A diag(1 - tanh(x)^2) --- */
typedef struct {
    const orc_mat *A;  /* fixed matrix (same pattern as J) */
    const double *b;
    double *t;         /* scratch n */
    orc_mat *J;        /* destination (so g can compute column scaling) */
    int threads;       /* unused in the scalar port (always 1) */
} orc_tanh_model;
void orc_tanh_f(double *out, const double *x, void *ud);
void orc_tanh_g(double *Jval, const double *x, void *ud);

#ifdef __cplusplus
}
#endif
#endif
