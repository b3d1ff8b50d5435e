/*
 * lsq_oracle.c -- CPU (scalar, fp64, one thread) restatement of the LeastSquaresOptim.jl
 * Levenberg-Marquardt / Dogleg linear-algebra hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see lsq_oracle.h for who may use it and for the parity
 * pinning statement ("trajectory parity with a live Julia run: UNPINNED").
 *
 * Every function cites the reference lines (relative to /root/reference) it follows.
 * Arithmetic that the reference delegates to Julia's stdlib (LinearAlgebra -> LAPACK /
 * OpenBLAS, SparseArrays; Julia compat "1.10", no Manifest pinned) is restated from the
 * published LAPACK algorithms: dpotf2, dpstf2, dlaqp2/dlarfg/dlarf (dgeqp3 semantics),
 * dorm2r, dlaic1, dlatrz/dlarz (dtzrzf), dormr3 (dormrz), and the rank-revealing
 * minimum-norm solve of LinearAlgebra.ldiv!(::QRPivoted, b) (the xGELSY algorithm with
 * rcond = min(m,n)*eps).
 *
 * SUMMATION ORDER.  By default (mode 0) every sum is taken in plain index order.  That is a
 * guess where the reference hands the sum to Julia's stdlib: `sum(abs2, .)` / `sum(.)` are
 * Base.mapreduce (pairwise above 1024 elements, an @simd loop below, i.e. reassociated into
 * 4 / 8 / 16 lanes depending on the CPU), `norm` is BLAS nrm2 from 32 elements up (OpenBLAS on
 * x86-64: extended-precision accumulation), dense `mul!` is BLAS gemv (SIMD dot products).
 * orc_set_sum_mode() switches those sites -- and only those: loops the reference writes out
 * itself (wdot, utils.jl:165-172; the SparseArrays products) stay sequential -- to other
 * plausible orders, so that tests can tell which results do NOT depend on the guess
 * (tests/golden/count_stable.json).
 */
#include "lsq_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MIN_DELTA 1e-16        /* types.jl:107 */
#define MAX_DELTA 1e16         /* types.jl:108 */
#define MIN_STEP_QUALITY 1e-3  /* types.jl:109 */
#define MIN_DIAGONAL 1e-6      /* types.jl:110 */
#define MAX_DIAGONAL 1e32      /* types.jl:111 */
#define DECREASE_THRESHOLD 0.25 /* dogleg.jl:38 */
#define INCREASE_THRESHOLD 0.75 /* dogleg.jl:39 */

/* ------------------------------------------------------------------------------------------
 * Summation-order model of the stdlib reductions (see the header comment).
 *   0  sequential (default)
 *   1  Base.mapreduce without SIMD: v = t0 + t1, then sequential; pairwise halves above 1024
 *   2/3/4  Base.mapreduce with its @simd loop vectorised into 4 / 8 / 16 lanes (lane j takes the
 *      elements 2+j, 2+j+L, ...; lanes combined by halving; scalar remainder added last);
 *      BLAS-backed sites (nrm2 from 32 elements, dense gemv 'T' dots) use the same lanes
 *   5  as 4, but nrm2 accumulates in long double (OpenBLAS's x87 dnrm2 kernel on x86-64)
 *   6  NOT a model of the reference: 64 lanes + halving tree at EVERY reduction, including the ones the
 *      reference writes as sequential loops (wdot, the sparse products) -- the class of orders the
 *      wavefront reductions of the HIP fast path use.  Separates "differs from the oracle because the
 *      run is round-off chaotic" from "differs because something is wrong".
 * ---------------------------------------------------------------------------------------- */
static int g_sum_mode = 0;
static unsigned long long g_rng = 0; /* modes >= 100: every reduction adds its terms in a random order (seed = mode) */
void orc_set_sum_mode(int mode) {
    g_sum_mode = mode;
    g_rng = 0x9E3779B97F4A7C15ull * (unsigned long long)(mode + 1);
}
static int everywhere(void) { return g_sum_mode == 6 || g_sum_mode >= 100; } /* also the reference's own loops */
/* modes >= 1000: index order, but the result of every reduction is moved by one rounding error (factor 1 +- 2^-53 at
 * random): the effect of ANY algebraically equivalent reformulation around the sums (1/beta folded into the next
 * product, fused epilogues, ...), which is what the HIP fast path does on top of reordering. */
static int noisy(void) { return g_sum_mode >= 1000; }
static unsigned long long rng_next(void) {
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return g_rng;
}
int orc_get_sum_mode(void) { return g_sum_mode; }
static int mode_lanes(void) {
    switch (g_sum_mode) {
    case 2: return 4;
    case 3: return 8;
    case 4: case 5: return 16;
    case 6: return 64;
    default: return g_sum_mode >= 100 ? 64 : 1;
    }
}
/* sum of t[first..last) in L lanes + sequential remainder, starting from v in lane 0 */
static double lanes_sum(double v, const double *t, int n, int L) {
    if (noisy()) {
        for (int i = 0; i < n; ++i) v += t[i];
        return v * ((rng_next() & 1ull) ? 1.0 + DBL_EPSILON / 2 * 2 : 1.0 - DBL_EPSILON / 2);
    }
    if (g_sum_mode >= 100) { /* random order */
        double *p = malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
        memcpy(p, t, (size_t)n * sizeof(double));
        for (int i = n - 1; i > 0; --i) {
            int j = (int)(rng_next() % (unsigned long long)(i + 1));
            double tmp = p[i]; p[i] = p[j]; p[j] = tmp;
        }
        for (int i = 0; i < n; ++i) v += p[i];
        free(p);
        return v;
    }
    if (L <= 1) {
        for (int i = 0; i < n; ++i) v += t[i];
        return v;
    }
    double acc[64];
    int body = (n / L) * L;
    if (body == 0) {
        for (int i = 0; i < n; ++i) v += t[i];
        return v;
    }
    acc[0] = v;
    for (int j = 1; j < L; ++j) acc[j] = 0.0;
    for (int i = 0; i < body; i += L)
        for (int j = 0; j < L; ++j) acc[j] += t[i + j];
    for (int h = L / 2; h >= 1; h /= 2)
        for (int j = 0; j < h; ++j) acc[j] += acc[j + h];
    v = acc[0];
    for (int i = body; i < n; ++i) v += t[i];
    return v;
}
/* Base.mapreduce_impl(identity, +, t, 1, n, 1024) */
static double mapreduce_sum(const double *t, int n) {
    if (n == 0) return 0.0;
    if (n == 1) return t[0];
    if (n < 1024) return lanes_sum(t[0] + t[1], t + 2, n - 2, mode_lanes());
    int mid = (n - 1) / 2 + 1; /* imid = ifirst + (ilast - ifirst) >> 1, first half = ifirst..imid */
    return mapreduce_sum(t, mid) + mapreduce_sum(t + mid, n - mid);
}
static double *scratch(int n) {
    static double *buf = NULL;
    static int cap = 0;
    if (n > cap) {
        free(buf);
        cap = n + 1024;
        buf = malloc((size_t)cap * sizeof(double));
    }
    return buf;
}
/* sum(x) as Julia's `sum` (levenberg_marquardt.jl:84) */
static double sum_stdlib(const double *x, int n) {
    if (g_sum_mode == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += x[i];
        return s;
    }
    return mapreduce_sum(x, n);
}
/* sum(abs2, x): levenberg_marquardt.jl:60,111,117; dogleg.jl:68,111,168,174; utils.jl:141,148 */
static double sumsq(const double *x, int n) {
    if (g_sum_mode == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += x[i] * x[i];
        return s;
    }
    double *t = scratch(n);
    for (int i = 0; i < n; ++i) t[i] = x[i] * x[i];
    return mapreduce_sum(t, n);
}
/* dot product as a BLAS kernel forms it (dense gemv 'T', syrk) */
static double dot_blas(const double *a, const double *b, int n) {
    int L = mode_lanes();
    if (L <= 1) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    double *t = scratch(n);
    for (int i = 0; i < n; ++i) t[i] = a[i] * b[i];
    return lanes_sum(0.0, t, n, L);
}
/* norm(x): LinearAlgebra.norm2 = generic_norm2 below 32 elements (a sequential sum of squares when no
 * scaling is needed), BLAS nrm2 from 32 elements up [stdlib].  lsmr.jl:74,119,123,206. */
static double nrm2(const double *x, int n) {
    if (g_sum_mode == 0 || n < 32) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += x[i] * x[i];
        return sqrt(s);
    }
    if (g_sum_mode == 5) {
        long double s = 0.0L;
        for (int i = 0; i < n; ++i) s += (long double)x[i] * (long double)x[i];
        return (double)sqrtl(s);
    }
    double *t = scratch(n);
    for (int i = 0; i < n; ++i) t[i] = x[i] * x[i];
    return sqrt(lanes_sum(0.0, t, n, mode_lanes()));
}
static void scal(double *x, int n, double a) {
    for (int i = 0; i < n; ++i) x[i] *= a;
}
static double maxabs(const double *x, int n) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) {
        double a = fabs(x[i]);
        if (isnan(a)) return a;
        if (a > m) m = a;
    }
    return m;
}
static int nnz_of(const orc_mat *A) { return A->kind == ORC_CSC ? A->colptr[A->n] : A->m * A->n; }

/* utils.jl:139-151 -- column sums of squares, dense and CSC. */
void orc_colsumabs2(double *v, const orc_mat *A) {
    if (A->kind == ORC_DENSE) {
        for (int j = 0; j < A->n; ++j) v[j] = sumsq(A->val + (size_t)j * A->m, A->m);
    } else {
        for (int j = 0; j < A->n; ++j) v[j] = sumsq(A->val + A->colptr[j], A->colptr[j + 1] - A->colptr[j]);
    }
}

/* utils.jl:155-161 -- row sums of squares (adjoint Jacobians). */
void orc_rowsumabs2(double *v, const orc_mat *A) {
    for (int i = 0; i < A->m; ++i) v[i] = 0.0;
    if (A->kind == ORC_DENSE) {
        for (int j = 0; j < A->n; ++j)
            for (int i = 0; i < A->m; ++i) {
                double a = A->val[(size_t)j * A->m + i];
                v[i] += a * a;
            }
    } else {
        int nz = nnz_of(A);
        for (int k = 0; k < nz; ++k) v[A->rowval[k]] += A->val[k] * A->val[k];
    }
}

static void scale_or_fill(double *y, int n, double beta) {
    /* LinearAlgebra._rmul_or_fill! [stdlib]: beta==0 overwrites (kills NaN), else scales */
    if (beta == 1.0) return;
    if (beta == 0.0)
        for (int i = 0; i < n; ++i) y[i] = 0.0;
    else
        scal(y, n, beta);
}

/* y <- alpha*A*x + beta*y.  SparseArrays mul! semantics [stdlib]: scale y by beta first, then
 * for each column scatter nzval*(x[col]*alpha).  Call sites: levenberg_marquardt.jl:114,
 * dogleg.jl:109,171, iterative_lsmr.jl:32,91.  Dense: gemv 'N' as column axpys. */
void orc_mul(double *y, const orc_mat *A, const double *x, double alpha, double beta) {
    scale_or_fill(y, A->m, beta);
    if (everywhere()) { /* every row's products summed as a 64-lane tree, then added to beta*y */
        int m = A->m, n = A->n;
        double *rows = calloc((size_t)m * (n > 0 ? n : 1), sizeof(double));
        int *cnt = calloc(m > 0 ? m : 1, sizeof(int));
        for (int j = 0; j < n; ++j) {
            double ax = x[j] * alpha;
            if (A->kind == ORC_DENSE) {
                for (int i = 0; i < m; ++i) rows[(size_t)i * n + cnt[i]++] = A->val[(size_t)j * m + i] * ax;
            } else {
                for (int k = A->colptr[j]; k < A->colptr[j + 1]; ++k) {
                    int i = A->rowval[k];
                    rows[(size_t)i * n + cnt[i]++] = A->val[k] * ax;
                }
            }
        }
        for (int i = 0; i < m; ++i) y[i] += lanes_sum(0.0, rows + (size_t)i * n, cnt[i], 64);
        free(rows); free(cnt);
        return;
    }
    if (A->kind == ORC_DENSE) {
        for (int j = 0; j < A->n; ++j) {
            double ax = x[j] * alpha;
            const double *col = A->val + (size_t)j * A->m;
            for (int i = 0; i < A->m; ++i) y[i] += col[i] * ax;
        }
    } else {
        for (int j = 0; j < A->n; ++j) {
            double ax = x[j] * alpha;
            for (int k = A->colptr[j]; k < A->colptr[j + 1]; ++k) y[A->rowval[k]] += A->val[k] * ax;
        }
    }
}

/* x <- alpha*A'*y + beta*x.  SparseArrays adjoint mul! [stdlib]: column dot products.
 * Call sites: levenberg_marquardt.jl:102, dogleg.jl:99, iterative_lsmr.jl:40,106. */
void orc_mulT(double *x, const orc_mat *A, const double *y, double alpha, double beta) {
    scale_or_fill(x, A->n, beta);
    if (A->kind == ORC_DENSE) {
        for (int j = 0; j < A->n; ++j) {
            const double *col = A->val + (size_t)j * A->m;
            double t = dot_blas(col, y, A->m);
            x[j] += t * alpha;
        }
    } else {
        for (int j = 0; j < A->n; ++j) {
            double t = 0.0;
            if (everywhere()) {
                int len = A->colptr[j + 1] - A->colptr[j];
                double *tb = scratch(len);
                for (int k = 0; k < len; ++k) tb[k] = A->val[A->colptr[j] + k] * y[A->rowval[A->colptr[j] + k]];
                t = lanes_sum(0.0, tb, len, 64);
            } else {
                for (int k = A->colptr[j]; k < A->colptr[j + 1]; ++k) t += A->val[k] * y[A->rowval[k]];
            }
            x[j] += t * alpha;
        }
    }
}

/* utils.jl:165-176 */
double orc_wdot(const double *x, const double *y, const double *w, int n) {
    if (everywhere()) {
        double *tb = scratch(n);
        for (int i = 0; i < n; ++i) tb[i] = w[i] * x[i] * y[i];
        return lanes_sum(0.0, tb, n, 64);
    }
    double out = 0.0;
    for (int i = 0; i < n; ++i) out += w[i] * x[i] * y[i];
    return out;
}
double orc_wnorm(const double *x, const double *w, int n) { return sqrt(orc_wdot(x, x, w, n)); }

/* utils.jl:39-55 */
double orc_maxabs_projected_gradient(const double *g, const double *x, const double *lower,
                                     const double *upper, int n) {
    if (!lower && !upper) return maxabs(g, n);
    double m = 0.0;
    for (int i = 0; i < n; ++i) {
        double gi = g[i];
        if (lower && x[i] <= lower[i] && gi > 0.0)
            gi = 0.0;
        else if (upper && x[i] >= upper[i] && gi < 0.0)
            gi = 0.0;
        double a = fabs(gi);
        if (a > m) m = a;
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * LSMR on  A = [J; diag(dg)] * diag(P)  acting on split vectors (y in R^m, x in R^n).
 * Operator wrappers: iterative_lsmr.jl:12-51 (PreconditionedMatrix / MyAdjoint),
 * :61-109 (DampenedVector / DampenedMatrix), :117-122 (InverseDiagonal: ldiv! MULTIPLIES).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const orc_mat *J;
    const double *dg; /* sqrt(damp) or NULL */
    const double *P;  /* or NULL (identity) */
    double *tmp, *tmp2;
} lsmr_op;

/* b <- alpha*A*a + beta*b : iterative_lsmr.jl:30-34 then :87-94 (damped) or plain mul!. */
static void op_mul(const lsmr_op *op, double *by, double *bx, const double *a, double alpha,
                   double beta) {
    int n = op->J->n, m = op->J->m;
    const double *t = a;
    if (op->P) {
        for (int i = 0; i < n; ++i) op->tmp[i] = a[i] * op->P[i]; /* :121 map!(*, y, x, ID._) */
        t = op->tmp;
    }
    if (op->dg) {
        if (beta != 1.0) { /* :88-90 rmul! on BOTH halves */
            scal(by, m, beta);
            scal(bx, n, beta);
        }
        orc_mul(by, op->J, t, alpha, 1.0);                              /* :91 */
        for (int i = 0; i < n; ++i) bx[i] = bx[i] + alpha * t[i] * op->dg[i]; /* :92 */
    } else {
        orc_mul(by, op->J, t, alpha, beta);
    }
}

/* b <- alpha*A'*a + beta*b : iterative_lsmr.jl:36-51 then :95-109 (damped). */
static void op_mulT(const lsmr_op *op, double *b, const double *ay, const double *ax,
                    double alpha, double beta) {
    int n = op->J->n;
    double *t = op->tmp;
    if (op->dg) {
        for (int i = 0; i < n; ++i) t[i] = 0.0;                  /* :99-100 fill!(b, 0) */
        orc_mulT(t, op->J, ay, 1.0, 1.0);                        /* :106 */
        for (int i = 0; i < n; ++i) t[i] = t[i] + 1.0 * ax[i] * op->dg[i]; /* :107 */
    } else {
        orc_mulT(t, op->J, ay, 1.0, 0.0);                        /* :40 */
    }
    const double *t2 = t;
    if (op->P) {
        for (int i = 0; i < n; ++i) op->tmp2[i] = t[i] * op->P[i]; /* :41 */
        t2 = op->tmp2;
    }
    scale_or_fill(b, n, beta);                                   /* :42-48 */
    for (int i = 0; i < n; ++i) b[i] += alpha * t2[i];           /* :49 axpy! */
}

/* norm(::DampenedVector) = sqrt(norm(y)^2 + norm(x)^2), iterative_lsmr.jl:72 -- literally: the square of a
 * square root is not the sum it came from (the round-1 oracle formed sqrt(sum y^2 + sum x^2)). */
static double dampened_norm(const double *y, int m, const double *x, int n) {
    double ny = nrm2(y, m), nx = nrm2(x, n);
    return sqrt(ny * ny + nx * nx);
}

/* lsmr.jl:53-238.  lambda == 0 for every caller.  Returns iter (mvps = 2*iter, :236). */
/* debugging aid (tools only): ORC_LSMR_TRACE=1 prints the stopping quantities of every inner iteration to stderr */
static int g_lsmr_trace = -1;
int orc_lsmr(double *x, const orc_mat *J, const double *diag, const double *P, double *by,
             double atol, double btol, double conlim, int maxiter, int *istop_out,
             double *normr_out, double *normAr_out) {
    const int m = J->m, n = J->n;
    const double lambda = 0.0;
    if (g_lsmr_trace < 0) g_lsmr_trace = getenv("ORC_LSMR_TRACE") != NULL;
    double *v = calloc(n, sizeof(double)), *h = calloc(n, sizeof(double));
    double *hbar = calloc(n, sizeof(double)), *tmp = calloc(n, sizeof(double));
    double *tmp2 = calloc(n, sizeof(double));
    double *bx = diag ? calloc(n, sizeof(double)) : NULL; /* zerosvector, iterative_lsmr.jl:246 */
    lsmr_op op = {J, diag, P, tmp, tmp2};
    if (maxiter < 0) maxiter = diag ? (m + n > n ? m + n : n) : (m > n ? m : n); /* lsmr.jl:55 */
    double ctol = conlim > 0 ? 1.0 / conlim : 0.0;               /* :71 */

    op_mul(&op, by, bx, x, -1.0, 1.0);                           /* :73 u = b - A x */
    double *uy = by, *ux = bx;
    double beta = diag ? dampened_norm(uy, m, ux, n) : nrm2(uy, m);  /* :74, il:72 */
    if (beta > 0) {
        double ib = 1.0 / beta;
        scal(uy, m, ib);
        if (ux) scal(ux, n, ib);
    }
    op_mulT(&op, v, uy, ux, 1.0, 0.0);                           /* :76 */
    double alpha = nrm2(v, n);
    if (alpha > 0) scal(v, n, 1.0 / alpha);

    double zetabar = alpha * beta, alphabar = alpha, rho = 1.0, rhobar = 1.0, cbar = 1.0,
           sbar = 0.0;                                           /* :82-87 */
    memcpy(h, v, n * sizeof(double));                            /* :89 */
    for (int i = 0; i < n; ++i) hbar[i] = 0.0;
    double betadd = beta, betad = 0.0, rhodold = 1.0, tautildeold = 0.0, thetatilde = 0.0,
           zeta = 0.0, d = 0.0;                                  /* :93-99 */
    double normA = -1.0, condA = -1.0, normx = -1.0;
    double normA2 = alpha * alpha, maxrbar = 0.0, minrbar = 1e100; /* :102-105 */
    double normb = beta, normr = beta, normAr = alpha * beta;
    int istop = 0, iter = 0;
    (void)normx;
    if (normAr != 0) {                                           /* :115 */
        while (iter < maxiter) {
            iter += 1;
            op_mul(&op, uy, ux, v, 1.0, -alpha);                 /* :118 */
            beta = diag ? dampened_norm(uy, m, ux, n) : nrm2(uy, m);
            if (beta > 0) {
                double ib = 1.0 / beta;
                scal(uy, m, ib);
                if (ux) scal(ux, n, ib);
                op_mulT(&op, v, uy, ux, 1.0, -beta);             /* :122 */
                alpha = nrm2(v, n);
                if (alpha > 0) scal(v, n, 1.0 / alpha);
            }
            /* :127-130 rotation Qhat */
            double alphahat = sqrt(alphabar * alphabar + lambda * lambda);
            double chat = alphabar / alphahat, shat = lambda / alphahat;
            /* :132-138 rotation Q_i */
            double rhoold = rho;
            rho = sqrt(alphahat * alphahat + beta * beta);
            double c = alphahat / rho, s = beta / rho;
            double thetanew = s * alpha;
            alphabar = c * alpha;
            /* :140-149 rotation Qbar_i */
            double rhobarold = rhobar, zetaold = zeta;
            double thetabar = sbar * rho;
            double rhotemp = cbar * rho;
            rhobar = sqrt((cbar * rho) * (cbar * rho) + thetanew * thetanew);
            cbar = cbar * rho / rhobar;
            sbar = thetanew / rhobar;
            zeta = cbar * zetabar;
            zetabar = -sbar * zetabar;
            /* :152-156 update hbar, x, h */
            double c1 = -thetabar * rho / (rhoold * rhobarold);
            double c2 = zeta / (rho * rhobar);
            double c3 = -thetanew / rho;
            for (int i = 0; i < n; ++i) hbar[i] *= c1;
            for (int i = 0; i < n; ++i) hbar[i] += 1.0 * h[i];
            for (int i = 0; i < n; ++i) x[i] += c2 * hbar[i];
            for (int i = 0; i < n; ++i) h[i] *= c3;
            for (int i = 0; i < n; ++i) h[i] += 1.0 * v[i];
            /* :164-184 estimate of ||r|| */
            double betaacute = chat * betadd, betacheck = -shat * betadd;
            double betahat = c * betaacute;
            betadd = -s * betaacute;
            double thetatildeold = thetatilde;
            double rhotildeold = sqrt(rhodold * rhodold + thetabar * thetabar);
            double ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
            thetatilde = stildeold * rhobar;
            rhodold = ctildeold * rhobar;
            betad = -stildeold * betad + ctildeold * betahat;
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
            double taud = (zeta - thetatilde * tautildeold) / rhodold;
            d = d + betacheck * betacheck;
            normr = sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
            /* :187-189 ||A|| */
            normA2 = normA2 + beta * beta;
            normA = sqrt(normA2);
            normA2 = normA2 + alpha * alpha;
            /* :192-196 cond(A) */
            maxrbar = maxrbar > rhobarold ? maxrbar : rhobarold;
            if (iter > 1) minrbar = minrbar < rhobarold ? minrbar : rhobarold;
            condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
            /* :205-221 */
            normAr = fabs(zetabar);
            normx = nrm2(x, n);
            double test1 = normr / normb;
            double test2 = normAr / (normA * normr);
            double test3 = 1.0 / condA;
            double t1 = test1 / (1.0 + normA * normx / normb);
            double rtol = btol + atol * normA * normx / normb;
            if (g_lsmr_trace) fprintf(stderr, "orc_lsmr iter %d test1 %.6e rtol %.6e test2 %.3e\n", iter, test1, rtol, test2);
            /* :224-231, first hit wins */
            if (iter >= maxiter) { istop = 7; break; }
            if (1.0 + test3 <= 1.0) { istop = 6; break; }
            if (1.0 + test2 <= 1.0) { istop = 5; break; }
            if (1.0 + t1 <= 1.0) { istop = 4; break; }
            if (test3 <= ctol) { istop = 3; break; }
            if (test2 <= atol) { istop = 2; break; }
            if (test1 <= rtol) { istop = 1; break; }
        }
    }
    if (istop_out) *istop_out = istop;
    if (normr_out) *normr_out = normr;
    if (normAr_out) *normAr_out = normAr;
    free(v); free(h); free(hbar); free(tmp); free(tmp2); free(bx);
    return iter;
}

/* default Jacobi preconditioner, iterative_lsmr.jl:129-141.  damp is PRE-sqrt (dtd/Delta) or NULL
 * for the literal 0 of the undamped path (:190). */
static void jacobi_preconditioner(double *P, const orc_mat *J, const double *damp) {
    int n = J->n;
    orc_colsumabs2(P, J);
    if (damp)
        for (int i = 0; i < n; ++i) P[i] += 1.0 * damp[i];
    for (int i = 0; i < n; ++i) P[i] = P[i] > 0.0 ? 1.0 / sqrt(P[i]) : 0.0;
}

/* iterative_lsmr.jl:179-198 (Dogleg: atol = btol = 1e-6 defaults). */
int orc_ldiv_lsmr(double *x, const orc_mat *J, const double *y, int *nmul) {
    int m = J->m, n = J->n;
    double *u = malloc(m * sizeof(double)), *P = malloc(n * sizeof(double));
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    memcpy(u, y, m * sizeof(double));
    jacobi_preconditioner(P, J, NULL);
    int iter = orc_lsmr(x, J, NULL, P, u, 1e-6, 1e-6, 1e8, -1, NULL, NULL, NULL);
    for (int i = 0; i < n; ++i) x[i] = x[i] * P[i]; /* :195-196 */
    if (nmul) *nmul = 2 * iter;
    free(u); free(P);
    return ORC_OK;
}

/* iterative_lsmr.jl:238-259 (LM: btol = 0.5; damp is clobbered with sqrt(damp), :252). */
int orc_ldiv_lsmr_damped(double *x, const orc_mat *J, const double *y, double *damp, int *nmul) {
    int m = J->m, n = J->n;
    double *u = malloc(m * sizeof(double)), *P = malloc(n * sizeof(double));
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    memcpy(u, y, m * sizeof(double));
    jacobi_preconditioner(P, J, damp);
    for (int i = 0; i < n; ++i) damp[i] = sqrt(damp[i]);
    int iter = orc_lsmr(x, J, damp, P, u, 1e-6, 0.5, 1e8, -1, NULL, NULL, NULL);
    for (int i = 0; i < n; ++i) x[i] = x[i] * P[i]; /* :256-257 */
    if (nmul) *nmul = 2 * iter;
    free(u); free(P);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Dense normal equations + Cholesky: dense_cholesky.jl:29-59.
 * ---------------------------------------------------------------------------------------- */
static void normal_matrix(double *C, const orc_mat *J) { /* mul!(cholm, J', J) :31,48 */
    int m = J->m, n = J->n;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            const double *a = J->val + (size_t)i * m, *b = J->val + (size_t)j * m;
            double s = dot_blas(a, b, m);
            C[(size_t)j * n + i] = s;
            C[(size_t)i * n + j] = s;
        }
}

/* LAPACK dpotf2, uplo='U' (what cholesky!(Symmetric(cholm)) runs, dense_cholesky.jl:57). */
int orc_potrf_upper(double *A, int n) {
    for (int j = 0; j < n; ++j) {
        double *cj = A + (size_t)j * n;
        double ajj = cj[j] - sumsq(cj, j);
        if (ajj <= 0.0 || isnan(ajj)) {
            cj[j] = ajj;
            return j + 1;
        }
        ajj = sqrt(ajj);
        cj[j] = ajj;
        for (int k = j + 1; k < n; ++k) {
            double *ck = A + (size_t)k * n;
            double s = 0.0;
            for (int i = 0; i < j; ++i) s += cj[i] * ck[i];
            ck[j] = (ck[j] - s) / ajj;
        }
    }
    return 0;
}

static void solve_UtU(const double *U, int n, double *b) {
    for (int i = 0; i < n; ++i) { /* U' z = b */
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= U[(size_t)i * n + k] * b[k];
        b[i] = s / U[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) { /* U x = z */
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= U[(size_t)k * n + i] * b[k];
        b[i] = s / U[(size_t)i * n + i];
    }
}

/* LAPACK dpstf2, uplo='U' (cholesky!(Symmetric(cholm), Val(true)), dense_cholesky.jl:33; Julia
 * passes tol = 0.0 [stdlib]).  Returns info (0 full rank, 1 rank deficient). */
int orc_pstrf_upper(double *A, int n, int *piv, int *rank, double tol) {
    double *work = calloc(2 * (size_t)n, sizeof(double));
    for (int i = 0; i < n; ++i) piv[i] = i;
    int pvt = 0;
    double ajj = n > 0 ? A[0] : 0.0;
    for (int i = 1; i < n; ++i)
        if (A[(size_t)i * n + i] > ajj) { pvt = i; ajj = A[(size_t)i * n + i]; }
    if (n == 0 || ajj <= 0.0 || isnan(ajj)) { *rank = 0; free(work); return 1; }
    double dstop = tol < 0 ? n * (DBL_EPSILON / 2) * ajj : tol;
    int j;
    for (j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) {
            if (j > 0) { double a = A[(size_t)i * n + (j - 1)]; work[i] += a * a; }
            work[n + i] = A[(size_t)i * n + i] - work[i];
        }
        if (j > 0) {
            pvt = j;
            for (int i = j + 1; i < n; ++i)
                if (work[n + i] > work[n + pvt]) pvt = i;
            ajj = work[n + pvt];
            if (ajj <= dstop || isnan(ajj)) {
                A[(size_t)j * n + j] = ajj;
                *rank = j;
                free(work);
                return 1;
            }
        }
        if (j != pvt) {
            A[(size_t)pvt * n + pvt] = A[(size_t)j * n + j];
            for (int i = 0; i < j; ++i) { /* swap columns j,pvt above row j */
                double t = A[(size_t)j * n + i];
                A[(size_t)j * n + i] = A[(size_t)pvt * n + i];
                A[(size_t)pvt * n + i] = t;
            }
            for (int k = pvt + 1; k < n; ++k) { /* swap rows j,pvt right of pvt */
                double t = A[(size_t)k * n + j];
                A[(size_t)k * n + j] = A[(size_t)k * n + pvt];
                A[(size_t)k * n + pvt] = t;
            }
            for (int i = j + 1; i < pvt; ++i) { /* A(j,i) <-> A(i,pvt) */
                double t = A[(size_t)i * n + j];
                A[(size_t)i * n + j] = A[(size_t)pvt * n + i];
                A[(size_t)pvt * n + i] = t;
            }
            double t = work[j]; work[j] = work[pvt]; work[pvt] = t;
            int ti = piv[pvt]; piv[pvt] = piv[j]; piv[j] = ti;
        }
        ajj = sqrt(ajj);
        A[(size_t)j * n + j] = ajj;
        for (int k = j + 1; k < n; ++k) {
            double s = 0.0;
            for (int i = 0; i < j; ++i) s += A[(size_t)j * n + i] * A[(size_t)k * n + i];
            A[(size_t)k * n + j] = (A[(size_t)k * n + j] - s) / ajj;
        }
    }
    *rank = n;
    free(work);
    return 0;
}

/* dense_cholesky.jl:29-35 (Dogleg: pivoted). */
int orc_ldiv_cholesky(double *x, const orc_mat *J, const double *y, int *nmul) {
    int n = J->n;
    if (J->kind != ORC_DENSE) return ORC_EDIM;
    double *C = malloc((size_t)n * n * sizeof(double));
    int *piv = malloc(n * sizeof(int));
    double *w = malloc(n * sizeof(double));
    normal_matrix(C, J);
    orc_mulT(x, J, y, 1.0, 0.0);
    int rank;
    int info = orc_pstrf_upper(C, n, piv, &rank, 0.0);
    int st = ORC_OK;
    if (info != 0 || rank < n) {
        st = ORC_ERANK; /* chkfullrank -> RankDeficientException [stdlib] */
    } else {
        for (int i = 0; i < n; ++i) w[i] = x[piv[i]]; /* permute!(B, piv) */
        solve_UtU(C, n, w);
        for (int i = 0; i < n; ++i) x[piv[i]] = w[i]; /* invpermute! */
    }
    if (nmul) *nmul = 1;
    free(C); free(piv); free(w);
    return st;
}

/* dense_cholesky.jl:43-59 (LM: + damp on the diagonal, unpivoted). */
int orc_ldiv_cholesky_damped(double *x, const orc_mat *J, const double *y, const double *damp,
                             int *nmul) {
    int n = J->n;
    if (J->kind != ORC_DENSE) return ORC_EDIM;
    double *C = malloc((size_t)n * n * sizeof(double));
    normal_matrix(C, J);
    for (int i = 0; i < n; ++i) C[(size_t)i * n + i] += damp[i];
    orc_mulT(x, J, y, 1.0, 0.0);
    int info = orc_potrf_upper(C, n);
    int st = ORC_OK;
    if (info != 0) st = ORC_ENOTPD;
    else solve_UtU(C, n, x);
    if (nmul) *nmul = 1;
    free(C);
    return st;
}

/* ------------------------------------------------------------------------------------------
 * Column-pivoted Householder QR + rank-revealing minimum-norm solve: dense_qr.jl:30-88.
 * ---------------------------------------------------------------------------------------- */
/* LAPACK dlarfg (without the safmin rescaling loop: inputs here are far from underflow). */
static double larfg(int n, double *alpha, double *x, int incx) {
    if (n <= 1) return 0.0;
    double xn = 0.0;
    for (int i = 0; i < n - 1; ++i) xn += x[(size_t)i * incx] * x[(size_t)i * incx];
    xn = sqrt(xn);
    if (xn == 0.0) return 0.0;
    double beta = -copysign(hypot(*alpha, xn), *alpha);
    double tau = (beta - *alpha) / beta;
    double sc = 1.0 / (*alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[(size_t)i * incx] *= sc;
    *alpha = beta;
    return tau;
}

/* LAPACK dgeqp3 semantics via the unblocked dlaqp2 recurrence (qr!(qrm, ColumnNorm()),
 * dense_qr.jl:37,83). */
/* Large shapes (C3: 16384 x 2048 = 1.3e11 flops, minutes in this scalar loop): tests may hand the factorisation -- and
 * only the factorisation; rank decision, Q'b, triangular solve and minimum-norm completion stay here -- to the very LAPACK
 * routine Julia dispatches to (scipy's dgeqp3), through this hook.  Same output convention: reflectors below the
 * diagonal, tau, 0-based jpvt.  NULL (default) = the restatement below. */
static orc_geqp3_fn g_geqp3_backend = 0;
void orc_set_geqp3_backend(orc_geqp3_fn fn) { g_geqp3_backend = fn; }

void orc_geqp3(double *A, int m, int n, int *jpvt, double *tau) {
    if (g_geqp3_backend) { g_geqp3_backend(A, m, n, jpvt, tau); return; }
    int mn = m < n ? m : n;
    double *vn1 = malloc(n * sizeof(double)), *vn2 = malloc(n * sizeof(double));
    double *w = malloc(n * sizeof(double));
    const double tol3z = sqrt(DBL_EPSILON / 2);
    for (int j = 0; j < n; ++j) {
        jpvt[j] = j;
        vn1[j] = vn2[j] = nrm2(A + (size_t)j * m, m);
    }
    for (int i = 0; i < mn; ++i) {
        int pvt = i;
        for (int j = i + 1; j < n; ++j)
            if (vn1[j] > vn1[pvt]) pvt = j;
        if (pvt != i) {
            double *a = A + (size_t)pvt * m, *b = A + (size_t)i * m;
            for (int k = 0; k < m; ++k) { double t = a[k]; a[k] = b[k]; b[k] = t; }
            int t = jpvt[pvt]; jpvt[pvt] = jpvt[i]; jpvt[i] = t;
            vn1[pvt] = vn1[i];
            vn2[pvt] = vn2[i];
        }
        double *ci = A + (size_t)i * m;
        if (i < m - 1) tau[i] = larfg(m - i, &ci[i], &ci[i + 1], 1);
        else tau[i] = larfg(1, &ci[i], &ci[i], 1);
        if (i < n - 1) { /* dlarf: apply H(i)' from the left to A(i:m, i+1:n) */
            double aii = ci[i];
            ci[i] = 1.0;
            for (int j = i + 1; j < n; ++j) {
                double *cj = A + (size_t)j * m;
                double s = 0.0;
                for (int k = i; k < m; ++k) s += ci[k] * cj[k];
                w[j] = s;
            }
            for (int j = i + 1; j < n; ++j) {
                double *cj = A + (size_t)j * m;
                double tw = tau[i] * w[j];
                for (int k = i; k < m; ++k) cj[k] -= ci[k] * tw;
            }
            ci[i] = aii;
        }
        for (int j = i + 1; j < n; ++j) { /* partial column norm downdate */
            if (vn1[j] != 0.0) {
                double r = fabs(A[(size_t)j * m + i]) / vn1[j];
                double temp = 1.0 - r * r;
                if (temp < 0.0) temp = 0.0;
                double q = vn1[j] / vn2[j];
                double temp2 = temp * q * q;
                if (temp2 <= tol3z) {
                    if (i < m - 1) {
                        vn1[j] = nrm2(A + (size_t)j * m + i + 1, m - i - 1);
                        vn2[j] = vn1[j];
                    } else {
                        vn1[j] = 0.0;
                        vn2[j] = 0.0;
                    }
                } else {
                    vn1[j] *= sqrt(temp);
                }
            }
        }
    }
    free(vn1); free(vn2); free(w);
}

/* LAPACK dlaic1: incremental condition estimation (job 1: largest, job 2: smallest). */
static void laic1(int job, int j, const double *x, double sest, const double *w, double gamma,
                  double *sestpr, double *s, double *c) {
    const double eps = DBL_EPSILON / 2;
    double alpha = 0.0;
    for (int i = 0; i < j; ++i) alpha += x[i] * w[i];
    double absalp = fabs(alpha), absgam = fabs(gamma), absest = fabs(sest);
    double s1, s2, tmp, b, cc, t, zeta1, zeta2, sine, cosine;
    if (job == 1) {
        if (sest == 0.0) {
            s1 = absgam > absalp ? absgam : absalp;
            if (s1 == 0.0) { *s = 0; *c = 1; *sestpr = 0; }
            else {
                *s = alpha / s1; *c = gamma / s1;
                tmp = sqrt(*s * *s + *c * *c);
                *s /= tmp; *c /= tmp; *sestpr = s1 * tmp;
            }
        } else if (absgam <= eps * absest) {
            *s = 1; *c = 0;
            tmp = absest > absalp ? absest : absalp;
            s1 = absest / tmp; s2 = absalp / tmp;
            *sestpr = tmp * sqrt(s1 * s1 + s2 * s2);
        } else if (absalp <= eps * absest) {
            s1 = absgam; s2 = absest;
            if (s1 <= s2) { *s = 1; *c = 0; *sestpr = s2; }
            else { *s = 0; *c = 1; *sestpr = s1; }
        } else if (absest <= eps * absalp || absest <= eps * absgam) {
            s1 = absgam; s2 = absalp;
            if (s1 <= s2) {
                tmp = s1 / s2; *s = sqrt(1 + tmp * tmp); *sestpr = s2 * *s;
                *c = (gamma / s2) / *s; *s = copysign(1.0, alpha) / *s;
            } else {
                tmp = s2 / s1; *c = sqrt(1 + tmp * tmp); *sestpr = s1 * *c;
                *s = (alpha / s1) / *c; *c = copysign(1.0, gamma) / *c;
            }
        } else {
            zeta1 = alpha / absest; zeta2 = gamma / absest;
            b = (1 - zeta1 * zeta1 - zeta2 * zeta2) * 0.5;
            cc = zeta1 * zeta1;
            t = b > 0 ? cc / (b + sqrt(b * b + cc)) : sqrt(b * b + cc) - b;
            sine = -zeta1 / t; cosine = -zeta2 / (1 + t);
            tmp = sqrt(sine * sine + cosine * cosine);
            *s = sine / tmp; *c = cosine / tmp;
            *sestpr = sqrt(t + 1) * absest;
        }
    } else {
        if (sest == 0.0) {
            *sestpr = 0;
            if ((absgam > absalp ? absgam : absalp) == 0.0) { sine = 1; cosine = 0; }
            else { sine = -gamma; cosine = alpha; }
            s1 = fabs(sine) > fabs(cosine) ? fabs(sine) : fabs(cosine);
            *s = sine / s1; *c = cosine / s1;
            tmp = sqrt(*s * *s + *c * *c);
            *s /= tmp; *c /= tmp;
        } else if (absgam <= eps * absest) {
            *s = 0; *c = 1; *sestpr = absgam;
        } else if (absalp <= eps * absest) {
            s1 = absgam; s2 = absest;
            if (s1 <= s2) { *s = 0; *c = 1; *sestpr = s1; }
            else { *s = 1; *c = 0; *sestpr = s2; }
        } else if (absest <= eps * absalp || absest <= eps * absgam) {
            s1 = absgam; s2 = absalp;
            if (s1 <= s2) {
                tmp = s1 / s2; *c = sqrt(1 + tmp * tmp); *sestpr = absest * (tmp / *c);
                *s = -(gamma / s2) / *c; *c = copysign(1.0, alpha) / *c;
            } else {
                tmp = s2 / s1; *s = sqrt(1 + tmp * tmp); *sestpr = absest / *s;
                *c = (alpha / s1) / *s; *s = -copysign(1.0, gamma) / *s;
            }
        } else {
            zeta1 = alpha / absest; zeta2 = gamma / absest;
            double n1 = 1 + zeta1 * zeta1 + fabs(zeta1 * zeta2);
            double n2 = fabs(zeta1 * zeta2) + zeta2 * zeta2;
            double norma = n1 > n2 ? n1 : n2;
            double test = 1 + 2 * (zeta1 - zeta2) * (zeta1 + zeta2);
            if (test >= 0) {
                b = (zeta1 * zeta1 + zeta2 * zeta2 + 1) * 0.5;
                cc = zeta2 * zeta2;
                t = cc / (b + sqrt(fabs(b * b - cc)));
                sine = zeta1 / (1 - t); cosine = -zeta2 / t;
                *sestpr = sqrt(t + 4 * eps * eps * norma) * absest;
            } else {
                b = (zeta2 * zeta2 + zeta1 * zeta1 - 1) * 0.5;
                cc = zeta1 * zeta1;
                t = b >= 0 ? -cc / (b + sqrt(b * b + cc)) : b - sqrt(b * b + cc);
                sine = -zeta1 / t; cosine = -zeta2 / (1 + t);
                *sestpr = sqrt(1 + t + 4 * eps * eps * norma) * absest;
            }
            tmp = sqrt(sine * sine + cosine * cosine);
            *s = sine / tmp; *c = cosine / tmp;
        }
    }
}

/* LinearAlgebra.ldiv!(A::QRPivoted, B, rcond) [stdlib; = LAPACK xGELSY]: Q'b (dorm2r), ICE rank
 * detection (dlaic1), triangular solve, minimum-norm completion by an RZ factorisation of
 * R(1:rnk,:) (dlatrz + dormr3) when rank-deficient, un-permutation.  A is m x n factored in
 * place by orc_geqp3; b has length lenb >= max(m,n); returns the detected rank. */
int orc_qrp_solve(double *A, int m, int n, const int *jpvt, const double *tau, double *b, int lenb,
                  double rcond) {
    int mn = m < n ? m : n;
    (void)lenb;
    if (m == 0 || n == 0) return 0;
    double smax = fabs(A[0]), smin = smax;
    if (smax == 0.0) {
        for (int i = 0; i < lenb; ++i) b[i] = 0.0;
        return 0;
    }
    double *wmin = calloc(mn, sizeof(double)), *wmax = calloc(mn, sizeof(double));
    int rnk = 1;
    wmin[0] = 1.0; wmax[0] = 1.0;
    while (rnk < mn) {
        int i = rnk;
        double sminpr, s1, c1, smaxpr, s2, c2;
        const double *col = A + (size_t)i * m;
        laic1(2, rnk, wmin, smin, col, col[i], &sminpr, &s1, &c1);
        laic1(1, rnk, wmax, smax, col, col[i], &smaxpr, &s2, &c2);
        if (smaxpr * rcond > sminpr) break;
        for (int j = 0; j < rnk; ++j) { wmin[j] *= s1; wmax[j] *= s2; }
        wmin[i] = c1; wmax[i] = c2;
        smin = sminpr; smax = smaxpr;
        rnk += 1;
    }
    free(wmin); free(wmax);
    /* Q'b: H(0), H(1), ... applied in order (dorm2r 'L','T') */
    for (int i = 0; i < mn; ++i) {
        const double *ci = A + (size_t)i * m;
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s += ci[k] * b[k];
        s *= tau[i];
        b[i] -= s;
        for (int k = i + 1; k < m; ++k) b[k] -= ci[k] * s;
    }
    if (rnk < n) {
        /* C = R(1:rnk, :) copy; RZ factorisation [R11 R12] = [T 0] Z (dlatrz) */
        int l = n - rnk;
        double *C = malloc((size_t)rnk * n * sizeof(double));
        double *tz = calloc(rnk, sizeof(double));
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < rnk; ++i) C[(size_t)j * rnk + i] = (i <= j) ? A[(size_t)j * m + i] : 0.0;
        for (int i = rnk - 1; i >= 0; --i) {
            double *vrow = C + (size_t)(n - l) * rnk + i; /* C(i, n-l:n-1), stride rnk */
            tz[i] = larfg(l + 1, &C[(size_t)i * rnk + i], vrow, rnk);
            /* dlarz 'R': apply H(i) to C(0:i-1, i:n-1) */
            for (int r = 0; r < i; ++r) {
                double wv = C[(size_t)i * rnk + r];
                for (int k = 0; k < l; ++k) wv += C[(size_t)(n - l + k) * rnk + r] * vrow[(size_t)k * rnk];
                C[(size_t)i * rnk + r] -= tz[i] * wv;
                for (int k = 0; k < l; ++k) C[(size_t)(n - l + k) * rnk + r] -= tz[i] * wv * vrow[(size_t)k * rnk];
            }
        }
        for (int i = rnk - 1; i >= 0; --i) { /* T z = (Q'b)(1:rnk) */
            double s = b[i];
            for (int k = i + 1; k < rnk; ++k) s -= C[(size_t)k * rnk + i] * b[k];
            b[i] = s / C[(size_t)i * rnk + i];
        }
        for (int i = rnk; i < n; ++i) b[i] = 0.0;
        for (int i = 0; i < rnk; ++i) { /* Z'b (dormr3 'L','T': i ascending) */
            const double *vrow = C + (size_t)(n - l) * rnk + i;
            double wv = b[i];
            for (int k = 0; k < l; ++k) wv += vrow[(size_t)k * rnk] * b[n - l + k];
            b[i] -= tz[i] * wv;
            for (int k = 0; k < l; ++k) b[n - l + k] -= tz[i] * vrow[(size_t)k * rnk] * wv;
        }
        free(C); free(tz);
    } else {
        for (int i = n - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * m + i] * b[k];
            b[i] = s / A[(size_t)i * m + i];
        }
    }
    double *work = malloc(n * sizeof(double));
    for (int i = 0; i < n; ++i) work[jpvt[i]] = b[i];
    for (int i = 0; i < n; ++i) b[i] = work[i];
    free(work);
    return rnk;
}

/* dense_qr.jl:30-42 */
int orc_ldiv_qr(double *x, const orc_mat *J, const double *y, int *nmul, int *rank_out) {
    if (J->kind != ORC_DENSE) return ORC_EDIM;
    int m = J->m, n = J->n, lu = m > n ? m : n, mn = m < n ? m : n;
    double *qrm = malloc((size_t)m * n * sizeof(double));
    double *u = calloc(lu, sizeof(double)), *tau = calloc(mn > 0 ? mn : 1, sizeof(double));
    int *jp = malloc(n * sizeof(int));
    memcpy(qrm, J->val, (size_t)m * n * sizeof(double));
    memcpy(u, y, m * sizeof(double));
    orc_geqp3(qrm, m, n, jp, tau);
    int rnk = orc_qrp_solve(qrm, m, n, jp, tau, u, lu, mn * DBL_EPSILON);
    memcpy(x, u, n * sizeof(double));
    if (nmul) *nmul = 1;
    if (rank_out) *rank_out = rnk;
    free(qrm); free(u); free(tau); free(jp);
    return ORC_OK;
}

/* dense_qr.jl:56-88 : QR of [J; diag(sqrt(damp))], rhs (y, 0). */
int orc_ldiv_qr_damped(double *x, const orc_mat *J, const double *y, const double *damp, int *nmul,
                       int *rank_out) {
    if (J->kind != ORC_DENSE) return ORC_EDIM;
    int m = J->m, n = J->n, M = m + n;
    double *qrm = calloc((size_t)M * n, sizeof(double));
    double *u = calloc(M, sizeof(double)), *tau = calloc(n > 0 ? n : 1, sizeof(double));
    int *jp = malloc(n * sizeof(int));
    for (int j = 0; j < n; ++j) {
        memcpy(qrm + (size_t)j * M, J->val + (size_t)j * m, m * sizeof(double));
        qrm[(size_t)j * M + m + j] = sqrt(damp[j]);
    }
    memcpy(u, y, m * sizeof(double));
    orc_geqp3(qrm, M, n, jp, tau);
    int rnk = orc_qrp_solve(qrm, M, n, jp, tau, u, M, n * DBL_EPSILON);
    memcpy(x, u, n * sizeof(double));
    if (nmul) *nmul = 1;
    if (rank_out) *rank_out = rnk;
    free(qrm); free(u); free(tau); free(jp);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Trust-region loops.
 * ---------------------------------------------------------------------------------------- */
/* utils.jl:7-31 : an if/elseif chain -- at most one flag is set. */
static int assess_convergence(const double *dx, int n, double maxabs_gr, double ssr,
                              double trial_ssr, double xtol, double ftol, double gtol, int accepted,
                              int *xc, int *fc, int *gc) {
    *xc = *fc = *gc = 0;
    if (accepted && fabs(trial_ssr - ssr) <= ftol * (fabs(ssr) + ftol)) *fc = 1;
    else if (maxabs(dx, n) <= xtol) *xc = 1;
    else if (maxabs_gr <= gtol) *gc = 1;
    return *xc || *fc || *gc;
}

static int first_nonfinite(const double *x, int n) { /* utils.jl:70-75 */
    for (int i = 0; i < n; ++i)
        if (!isfinite(x[i])) return i;
    return -1;
}

static int check_bounds(const double *x, int n, const double *lo, const double *hi) {
    /* levenberg_marquardt.jl:49-51, dogleg.jl:52-54 */
    for (int i = 0; i < n; ++i) {
        if (lo && !(x[i] >= lo[i])) return 0;
        if (hi && !(x[i] <= hi[i])) return 0;
    }
    return 1;
}

static void apply_box(double *dx, const double *x, int n, const double *lo, const double *hi) {
    /* levenberg_marquardt.jl:89-98, dogleg.jl:148-160: the STEP is clipped, x_new = x - dx */
    /* Julia's min / max return NaN when EITHER argument is NaN */
    if (lo)
        for (int i = 0; i < n; ++i) {
            double a = x[i] - lo[i];
            dx[i] = (isnan(dx[i]) || isnan(a)) ? dx[i] + a : (dx[i] < a ? dx[i] : a);
        }
    if (hi)
        for (int i = 0; i < n; ++i) {
            double a = x[i] - hi[i];
            dx[i] = (isnan(dx[i]) || isnan(a)) ? dx[i] + a : (dx[i] > a ? dx[i] : a);
        }
}

static void record(const orc_options *o, int it, int n, double ssr, double g, double delta,
                   double rho, int inner, int acc, const double *x) {
    if (it > o->trace_cap) return;
    int k = it - 1;
    if (o->trace_ssr) o->trace_ssr[k] = ssr;
    if (o->trace_gnorm) o->trace_gnorm[k] = g;
    if (o->trace_delta) o->trace_delta[k] = delta;
    if (o->trace_rho) o->trace_rho[k] = rho;
    if (o->trace_inner) o->trace_inner[k] = inner;
    if (o->trace_accept) o->trace_accept[k] = acc;
    if (o->trace_x) memcpy(o->trace_x + (size_t)k * n, x, n * sizeof(double));
}

static int solve_damped(int solver, double *dx, const orc_mat *J, const double *f, double *damp,
                        int *nmul) {
    switch (solver) {
    case ORC_LSMR: return orc_ldiv_lsmr_damped(dx, J, f, damp, nmul);
    case ORC_CHOLESKY: return orc_ldiv_cholesky_damped(dx, J, f, damp, nmul);
    default: return orc_ldiv_qr_damped(dx, J, f, damp, nmul, NULL);
    }
}
static int solve_gn(int solver, double *dx, const orc_mat *J, const double *f, int *nmul) {
    switch (solver) {
    case ORC_LSMR: return orc_ldiv_lsmr(dx, J, f, nmul);
    case ORC_CHOLESKY: return orc_ldiv_cholesky(dx, J, f, nmul);
    default: return orc_ldiv_qr(dx, J, f, nmul, NULL);
    }
}

/* levenberg_marquardt.jl:39-144 */
static int optimize_lm(int solver, orc_mat *J, double *x, double *fcur, orc_f_cb f, orc_g_cb g,
                       void *ud, const orc_options *o, orc_result *r) {
    int m = J->m, n = J->n;
    double *dx = calloc(n, sizeof(double)), *dtd = calloc(n, sizeof(double));
    double *ftrial = calloc(m, sizeof(double)), *fpred = calloc(m, sizeof(double));
    double delta = o->delta > 0 ? o->delta : 10.0;
    double decrease_factor = 2.0;
    int f_calls = 0, g_calls = 0, mul_calls = 0, converged = 0, xc = 0, fc = 0, gc = 0;
    int st = ORC_OK;
    f(fcur, x, ud); f_calls++;
    double ssr = sumsq(fcur, m), maxabs_gr = INFINITY;
    int need_jac = 1, iter = 0;
    while (!converged && iter < o->iterations) {
        iter++;
        int bad = first_nonfinite(x, n);
        if (bad >= 0) { st = ORC_ENONFINITE; r->bad_index = bad; iter--; break; }
        if (need_jac) { g(J->val, x, ud); g_calls++; need_jac = 0; }
        orc_colsumabs2(dtd, J);                                             /* :82 */
        double mean = sum_stdlib(dtd, n) / n;                               /* :84 */
        double lo = MIN_DIAGONAL * mean, hi = MAX_DIAGONAL * mean;
        for (int i = 0; i < n; ++i) dtd[i] = dtd[i] > hi ? hi : (dtd[i] < lo ? lo : dtd[i]);
        scal(dtd, n, 1.0 / delta);                                          /* :86 */
        int lmiter = 0;
        st = solve_damped(solver, dx, J, fcur, dtd, &lmiter);               /* :87 */
        if (st != ORC_OK) break;
        apply_box(dx, x, n, o->lower, o->upper);
        mul_calls += lmiter;
        orc_mulT(dtd, J, fcur, 1.0, 0.0);                                   /* :102 */
        mul_calls++;
        maxabs_gr = orc_maxabs_projected_gradient(dtd, x, o->lower, o->upper, n);
        for (int i = 0; i < n; ++i) x[i] += -1.0 * dx[i];                   /* :106 */
        f(ftrial, x, ud); f_calls++;
        double trial_ssr = sumsq(ftrial, m);
        orc_mul(fpred, J, dx, 1.0, 0.0);                                    /* :114 */
        mul_calls++;
        for (int i = 0; i < m; ++i) fpred[i] += -1.0 * fcur[i];
        double predicted_ssr = sumsq(fpred, m);
        double pred_red = fabs(ssr - predicted_ssr);
        double rho = pred_red > 0 ? (ssr - trial_ssr) / pred_red : 0.0;
        int accepted = rho > MIN_STEP_QUALITY;                              /* :122 strict */
        converged = assess_convergence(dx, n, maxabs_gr, ssr, trial_ssr, o->x_tol, o->f_tol,
                                       o->g_tol, accepted, &xc, &fc, &gc);
        if (accepted) {
            memcpy(fcur, ftrial, m * sizeof(double));
            ssr = trial_ssr;
            double q = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
            double dn = delta / (1.0 / 3.0 > q ? 1.0 / 3.0 : q);            /* :130 */
            delta = dn < MAX_DELTA ? dn : MAX_DELTA;
            decrease_factor = 2.0;
            need_jac = 1;
        } else {
            for (int i = 0; i < n; ++i) x[i] += 1.0 * dx[i];                /* :135 */
            double dn = delta / decrease_factor;
            delta = dn > MIN_DELTA ? dn : MIN_DELTA;
            decrease_factor *= 2.0;
        }
        record(o, iter, n, ssr, maxabs_gr, delta, rho, lmiter, accepted, x);
    }
    r->optimizer = ORC_LM; r->ssr = ssr; r->iterations = iter; r->converged = converged;
    r->x_converged = xc; r->f_converged = fc; r->g_converged = gc;
    r->f_calls = f_calls; r->g_calls = g_calls; r->mul_calls = mul_calls; r->status = st;
    free(dx); free(dtd); free(ftrial); free(fpred);
    return st;
}

/* dogleg.jl:41-203 */
static int optimize_dogleg(int solver, orc_mat *J, double *x, double *fcur, orc_f_cb f, orc_g_cb g,
                           void *ud, const orc_options *o, orc_result *r) {
    int m = J->m, n = J->n;
    double *dgn = calloc(n, sizeof(double)), *dgr = calloc(n, sizeof(double));
    double *dx = calloc(n, sizeof(double)), *dtd = calloc(n, sizeof(double));
    double *ftrial = calloc(m, sizeof(double)), *fpred = calloc(m, sizeof(double));
    double delta = o->delta > 0 ? o->delta : 1.0;
    int reuse = 0;
    double wnorm_dgn = 0.0, wnorm_dgr = 0.0, alpha = 0.0;
    int f_calls = 0, g_calls = 0, mul_calls = 0, converged = 0, xc = 0, fc = 0, gc = 0;
    int st = ORC_OK;
    f(fcur, x, ud); f_calls++;
    double ssr = sumsq(fcur, m), maxabs_gr = INFINITY;
    int iter = 0;
    while (!converged && iter < o->iterations) {
        iter++;
        int bad = first_nonfinite(x, n);
        if (bad >= 0) { st = ORC_ENONFINITE; r->bad_index = bad; iter--; break; }
        int ls_iter = 0;
        if (!reuse) {
            g(J->val, x, ud); g_calls++;
            orc_colsumabs2(dtd, J);                                         /* :85 */
            for (int i = 0; i < n; ++i)                                     /* :90 absolute clamp */
                dtd[i] = dtd[i] > MAX_DIAGONAL ? MAX_DIAGONAL : (dtd[i] < MIN_DIAGONAL ? MIN_DIAGONAL : dtd[i]);
            if (iter == 1) {
                double wx = orc_wnorm(x, dtd, n);
                if (wx > 0) delta *= wx;                                    /* :92-97 */
            }
            orc_mulT(dgr, J, fcur, 1.0, 0.0); mul_calls++;                  /* :99 */
            maxabs_gr = orc_maxabs_projected_gradient(dgr, x, o->lower, o->upper, n);
            for (int i = 0; i < n; ++i) dgr[i] = dgr[i] / dtd[i];           /* :105 */
            wnorm_dgr = orc_wnorm(dgr, dtd, n);
            orc_mul(fpred, J, dgr, 1.0, 0.0); mul_calls++;                  /* :109 */
            alpha = wnorm_dgr * wnorm_dgr / sumsq(fpred, m);                /* :111 */
            for (int i = 0; i < n; ++i) dgn[i] = 0.0;
            st = solve_gn(solver, dgn, J, fcur, &ls_iter);                  /* :115 */
            if (st != ORC_OK) break;
            mul_calls += ls_iter;
            wnorm_dgn = orc_wnorm(dgn, dtd, n);
        }
        double wnorm_dx;
        if (wnorm_dgn <= delta) {                                           /* :120 case 1 */
            memcpy(dx, dgn, n * sizeof(double));
            wnorm_dx = wnorm_dgn;
        } else if (wnorm_dgr * alpha >= delta) {                            /* :124 case 2 */
            memcpy(dx, dgr, n * sizeof(double));
            scal(dx, n, delta / wnorm_dgr);
            wnorm_dx = delta;
        } else {                                                            /* :131 case 3 */
            double b_dot_a = alpha * orc_wdot(dgr, dgn, dtd, n);
            double a2 = (alpha * wnorm_dgr) * (alpha * wnorm_dgr);
            double bma2 = a2 - 2 * b_dot_a + wnorm_dgn * wnorm_dgn;
            double c = b_dot_a - a2;
            double d = sqrt(c * c + bma2 * (delta * delta - a2));
            double beta = (c <= 0) ? (d - c) / bma2 : (delta * delta - a2) / (d + c);
            memcpy(dx, dgn, n * sizeof(double));
            scal(dx, n, beta);
            double ab = alpha * (1 - beta);
            for (int i = 0; i < n; ++i) dx[i] += ab * dgr[i];
            wnorm_dx = orc_wnorm(dx, dtd, n);
        }
        apply_box(dx, x, n, o->lower, o->upper);
        for (int i = 0; i < n; ++i) x[i] += -1.0 * dx[i];                   /* :160 */
        f(ftrial, x, ud); f_calls++;
        double trial_ssr = sumsq(ftrial, m);
        orc_mul(fpred, J, dx, 1.0, 0.0); mul_calls++;                       /* :171 */
        for (int i = 0; i < m; ++i) fpred[i] += -1.0 * fcur[i];
        double predicted_ssr = sumsq(fpred, m);
        double pred_red = fabs(ssr - predicted_ssr);
        double rho = pred_red > 0 ? (ssr - trial_ssr) / pred_red : 0.0;
        int accepted = rho >= MIN_STEP_QUALITY;                             /* :178 non-strict */
        converged = assess_convergence(dx, n, maxabs_gr, ssr, trial_ssr, o->x_tol, o->f_tol,
                                       o->g_tol, accepted, &xc, &fc, &gc);
        if (accepted) {
            reuse = 0;
            memcpy(fcur, ftrial, m * sizeof(double));
            ssr = trial_ssr;
        } else {
            reuse = 1;
            for (int i = 0; i < n; ++i) x[i] += 1.0 * dx[i];
        }
        if (rho < DECREASE_THRESHOLD) {                                     /* :193-197 */
            double dn = delta * 0.5;
            delta = dn > MIN_DELTA ? dn : MIN_DELTA;
        } else if (rho > INCREASE_THRESHOLD) {
            double dn = 3.0 * wnorm_dx;
            delta = delta > dn ? delta : dn;
        }
        record(o, iter, n, ssr, maxabs_gr, delta, rho, ls_iter, accepted, x);
    }
    r->optimizer = ORC_DOGLEG; r->ssr = ssr; r->iterations = iter; r->converged = converged;
    r->x_converged = xc; r->f_converged = fc; r->g_converged = gc;
    r->f_calls = f_calls; r->g_calls = g_calls; r->mul_calls = mul_calls; r->status = st;
    free(dgn); free(dgr); free(dx); free(dtd); free(ftrial); free(fpred);
    return st;
}

/* types.jl:207-209 front door (after default resolution, done by the caller). */
int orc_optimize(int optimizer, int solver, orc_mat *J, double *x, double *fcur, orc_f_cb f,
                 orc_g_cb g, void *ud, const orc_options *opt, orc_result *res) {
    memset(res, 0, sizeof(*res));
    res->bad_index = -1;
    int n = J->n;
    if (!check_bounds(x, n, opt->lower, opt->upper)) { res->status = ORC_EBOUNDS; return ORC_EBOUNDS; }
    if (solver != ORC_LSMR && J->kind != ORC_DENSE) { res->status = ORC_EDIM; return ORC_EDIM; } /* types.jl:115-117 */
    if (optimizer == ORC_LM) return optimize_lm(solver, J, x, fcur, f, g, ud, opt, res);
    return optimize_dogleg(solver, J, x, fcur, f, g, ud, opt, res);
}

/* ------------------------------------------------------------------------------------------
 * Synthetic benchmark model: r(x) = A tanh(x) - b ; J = A diag(1 - tanh(x)^2).
 * ---------------------------------------------------------------------------------------- */
void orc_tanh_f(double *out, const double *x, void *ud) {
    orc_tanh_model *md = (orc_tanh_model *)ud;
    int n = md->A->n, m = md->A->m;
    for (int j = 0; j < n; ++j) md->t[j] = tanh(x[j]);
    orc_mul(out, md->A, md->t, 1.0, 0.0);
    for (int i = 0; i < m; ++i) out[i] -= md->b[i];
}
void orc_tanh_g(double *Jval, const double *x, void *ud) {
    orc_tanh_model *md = (orc_tanh_model *)ud;
    const orc_mat *A = md->A;
    for (int j = 0; j < A->n; ++j) {
        double t = tanh(x[j]);
        double s = 1.0 - t * t;
        if (A->kind == ORC_DENSE) {
            for (int i = 0; i < A->m; ++i) Jval[(size_t)j * A->m + i] = A->val[(size_t)j * A->m + i] * s;
        } else {
            for (int k = A->colptr[j]; k < A->colptr[j + 1]; ++k) Jval[k] = A->val[k] * s;
        }
    }
}
