/*
 * lsq_oracle_omp.c -- "generous CPU" baseline: the oracle's Levenberg-Marquardt + LSMR path on the sparse tanh model
 * (bench.py's C4 workload), parallelised with OpenMP over ALL host cores.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see lsq_oracle.h): bench.py's cpu_baseline leg reports it NEXT TO the
 * one-thread oracle, so that the GPU / CPU ratio is not inflated by the reference's serial loops.
 * It is NOT a parity oracle: the reference's SparseArrays products and vector loops are serial; here
 *   - J*v runs over a CSR mirror (rows split over the threads), J'*u over the CSC copy (columns split),
 *   - g! writes both copies (as the GPU path does), f! = A tanh(x) - b over the CSR copy of A,
 *   - every norm / sum is an OpenMP reduction (so the summation order, and with it the last bits, differ),
 * i.e. the same algorithm (levenberg_marquardt.jl:39-144, iterative_lsmr.jl:238-259, lsmr.jl:53-238) with the
 * data-parallel restructuring a CPU implementer would do.  tests/test_oracle.py checks it against lsq_oracle.c.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int m, n, nnz;
    const int *colptr, *rowval;   /* CSC pattern */
    const double *A;              /* CSC values of the model matrix */
    int *rowptr, *colidx;         /* CSR mirror */
    double *Acsr, *Jcsc, *Jcsr;
    const double *b;
    double *s;                    /* n: 1 - tanh(x)^2 / tanh(x) scratch */
} omp_prob;

static double psumsq(const double *x, int n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    return s;
}
/* y = alpha * M x + beta * y over a CSR matrix */
static void csr_mul(const omp_prob *p, const double *val, const double *x, double alpha, double beta, double *y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->m; ++i) {
        double t = 0.0;
        for (int k = p->rowptr[i]; k < p->rowptr[i + 1]; ++k) t += val[k] * x[p->colidx[k]];
        y[i] = alpha * t + (beta == 0.0 ? 0.0 : beta * y[i]);
    }
}
/* x = M' y over the CSC copy */
static void csc_mulT(const omp_prob *p, const double *val, const double *y, double *x) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < p->n; ++j) {
        double t = 0.0;
        for (int k = p->colptr[j]; k < p->colptr[j + 1]; ++k) t += val[k] * y[p->rowval[k]];
        x[j] = t;
    }
}
static void model_f(omp_prob *p, const double *x, double *out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < p->n; ++j) p->s[j] = tanh(x[j]);
    csr_mul(p, p->Acsr, p->s, 1.0, 0.0, out);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->m; ++i) out[i] -= p->b[i];
}
static void model_g(omp_prob *p, const double *x) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < p->n; ++j) {
        double t = tanh(x[j]);
        p->s[j] = 1.0 - t * t;
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < p->n; ++j)
        for (int k = p->colptr[j]; k < p->colptr[j + 1]; ++k) p->Jcsc[k] = p->A[k] * p->s[j];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->m; ++i)
        for (int k = p->rowptr[i]; k < p->rowptr[i + 1]; ++k) p->Jcsr[k] = p->Acsr[k] * p->s[p->colidx[k]];
}

/* damped, Jacobi-preconditioned LSMR (iterative_lsmr.jl:238-259 + lsmr.jl:53-238); returns the iteration count */
static int lsmr_damped(omp_prob *p, const double *y, double *damp, double *x, double *u, double *ux, double *v, double *h,
                       double *hbar, double *P, double *tmp) {
    const int m = p->m, n = p->n;
    const double atol = 1e-6, btol = 0.5, ctol = 1e-8;
    const int maxiter = m + n;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
        double sq = 0.0;
        for (int k = p->colptr[j]; k < p->colptr[j + 1]; ++k) sq += p->Jcsc[k] * p->Jcsc[k];
        sq += damp[j];
        P[j] = sq > 0.0 ? 1.0 / sqrt(sq) : 0.0;
        damp[j] = sqrt(damp[j]);
        x[j] = 0.0;
        ux[j] = 0.0;
    }
    memcpy(u, y, (size_t)m * sizeof(double));
    double ny = sqrt(psumsq(u, m));
    double beta = sqrt(ny * ny + 0.0);
    if (beta > 0) {
        const double ib = 1.0 / beta;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < m; ++i) u[i] *= ib;
    }
    csc_mulT(p, p->Jcsc, u, tmp);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) v[j] = (tmp[j] + ux[j] * damp[j]) * P[j];
    double alpha = sqrt(psumsq(v, n));
    if (alpha > 0) {
        const double ia = 1.0 / alpha;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; ++j) v[j] *= ia;
    }
    double zetabar = alpha * beta, alphabar = alpha, rho = 1.0, rhobar = 1.0, cbar = 1.0, sbar = 0.0;
    memcpy(h, v, (size_t)n * sizeof(double));
    memset(hbar, 0, (size_t)n * sizeof(double));
    double betadd = beta, betad = 0.0, rhodold = 1.0, tautildeold = 0.0, thetatilde = 0.0, zeta = 0.0, d = 0.0;
    double normA2 = alpha * alpha, maxrbar = 0.0, minrbar = 1e100;
    const double normb = beta;
    double normr = beta, normAr = alpha * beta;
    int iter = 0;
    if (normAr == 0) return 0;
    while (iter < maxiter) {
        iter++;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; ++j) tmp[j] = v[j] * P[j];
        csr_mul(p, p->Jcsr, tmp, 1.0, -alpha, u);                       /* u <- A v - alpha u */
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; ++j) ux[j] = -alpha * ux[j] + tmp[j] * damp[j];
        const double nu = sqrt(psumsq(u, m)), nx = sqrt(psumsq(ux, n));
        beta = sqrt(nu * nu + nx * nx);
        if (beta > 0) {
            const double ib = 1.0 / beta;
#pragma omp parallel for schedule(static)
            for (int i = 0; i < m; ++i) u[i] *= ib;
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; ++j) ux[j] *= ib;
            csc_mulT(p, p->Jcsc, u, tmp);
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; ++j) v[j] = -beta * v[j] + (tmp[j] + ux[j] * damp[j]) * P[j];
            alpha = sqrt(psumsq(v, n));
            if (alpha > 0) {
                const double ia = 1.0 / alpha;
#pragma omp parallel for schedule(static)
                for (int j = 0; j < n; ++j) v[j] *= ia;
            }
        }
        const double alphahat = alphabar, chat = 1.0, shat = 0.0;          /* lambda = 0 */
        const double rhoold = rho;
        rho = sqrt(alphahat * alphahat + beta * beta);
        const double c = alphahat / rho, s = beta / rho;
        const double thetanew = s * alpha;
        alphabar = c * alpha;
        const double rhobarold = rhobar, zetaold = zeta;
        const double thetabar = sbar * rho, rhotemp = cbar * rho;
        rhobar = sqrt((cbar * rho) * (cbar * rho) + thetanew * thetanew);
        cbar = cbar * rho / rhobar;
        sbar = thetanew / rhobar;
        zeta = cbar * zetabar;
        zetabar = -sbar * zetabar;
        const double c1 = -thetabar * rho / (rhoold * rhobarold), c2 = zeta / (rho * rhobar), c3 = -thetanew / rho;
        double sx = 0.0;
#pragma omp parallel for reduction(+ : sx) schedule(static)
        for (int j = 0; j < n; ++j) {
            const double hb = hbar[j] * c1 + h[j];
            hbar[j] = hb;
            const double xj = x[j] + c2 * hb;
            x[j] = xj;
            h[j] = h[j] * c3 + v[j];
            sx += xj * xj;
        }
        const double betaacute = chat * betadd, betacheck = -shat * betadd;
        const double betahat = c * betaacute;
        betadd = -s * betaacute;
        const double thetatildeold = thetatilde;
        const double rhotildeold = sqrt(rhodold * rhodold + thetabar * thetabar);
        const double ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
        thetatilde = stildeold * rhobar;
        rhodold = ctildeold * rhobar;
        betad = -stildeold * betad + ctildeold * betahat;
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
        const double taud = (zeta - thetatilde * tautildeold) / rhodold;
        d = d + betacheck * betacheck;
        normr = sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
        normA2 = normA2 + beta * beta;
        const double normA = sqrt(normA2);
        normA2 = normA2 + alpha * alpha;
        maxrbar = maxrbar > rhobarold ? maxrbar : rhobarold;
        if (iter > 1) minrbar = minrbar < rhobarold ? minrbar : rhobarold;
        const double condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
        normAr = fabs(zetabar);
        const double normx = sqrt(sx);
        const double test1 = normr / normb, test2 = normAr / (normA * normr), test3 = 1.0 / condA;
        const double t1 = test1 / (1.0 + normA * normx / normb), rtol = btol + atol * normA * normx / normb;
        if (iter >= maxiter || 1.0 + test3 <= 1.0 || 1.0 + test2 <= 1.0 || 1.0 + t1 <= 1.0 || test3 <= ctol || test2 <= atol ||
            test1 <= rtol)
            break;
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) x[j] *= P[j];
    return iter;
}

/* ---- handle: CSR mirror and every work array built ONCE, pages first-touched by the threads that will stream them ---- */
typedef struct {
    omp_prob p;
    double *fcur, *ftrial, *u, *dx, *dtd, *ux, *v, *h, *hbar, *P, *tmp;
    int threads;
} omp_handle;

static double *palloc(size_t n) {   /* parallel first touch with the static partition of the loops above */
    double *a = malloc((n > 0 ? n : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) a[i] = 0.0;
    return a;
}

void *orc_omp_create(int m, int n, const int *colptr, const int *rowval, const double *Aval, const double *b, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    omp_handle *H = calloc(1, sizeof(omp_handle));
    omp_prob *p = &H->p;
    H->threads = threads > 0 ? threads : omp_get_max_threads();
    p->m = m; p->n = n; p->nnz = colptr[n]; p->b = b;
    const int nnz = p->nnz;
    /* private, first-touched copies of the CSC pattern / values too (columns are split statically over the threads) */
    int *cp = malloc(((size_t)n + 1) * sizeof(int)), *rv = malloc((size_t)nnz * sizeof(int));
    double *Ac = malloc((size_t)nnz * sizeof(double));
    memcpy(cp, colptr, ((size_t)n + 1) * sizeof(int));
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j)
        for (int k = colptr[j]; k < colptr[j + 1]; ++k) { rv[k] = rowval[k]; Ac[k] = Aval[k]; }
    p->colptr = cp; p->rowval = rv; p->A = Ac;
    p->rowptr = calloc((size_t)m + 1, sizeof(int));
    for (int k = 0; k < nnz; ++k) p->rowptr[rowval[k] + 1]++;
    for (int i = 0; i < m; ++i) p->rowptr[i + 1] += p->rowptr[i];
    int *ci = malloc((size_t)nnz * sizeof(int));
    double *Ar = malloc((size_t)nnz * sizeof(double));
    {   /* serial fill into scratch, then a parallel copy so that every row's entries live near the thread that owns the row */
        int *fill = malloc((size_t)m * sizeof(int));
        memcpy(fill, p->rowptr, (size_t)m * sizeof(int));
        int *ci0 = malloc((size_t)nnz * sizeof(int));
        double *Ar0 = malloc((size_t)nnz * sizeof(double));
        for (int j = 0; j < n; ++j)
            for (int k = colptr[j]; k < colptr[j + 1]; ++k) {
                const int q = fill[rowval[k]]++;
                ci0[q] = j;
                Ar0[q] = Aval[k];
            }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < m; ++i)
            for (int k = p->rowptr[i]; k < p->rowptr[i + 1]; ++k) { ci[k] = ci0[k]; Ar[k] = Ar0[k]; }
        free(fill); free(ci0); free(Ar0);
    }
    p->colidx = ci; p->Acsr = Ar;
    p->Jcsc = malloc((size_t)nnz * sizeof(double));
    p->Jcsr = malloc((size_t)nnz * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j)
        for (int k = cp[j]; k < cp[j + 1]; ++k) p->Jcsc[k] = 0.0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i)
        for (int k = p->rowptr[i]; k < p->rowptr[i + 1]; ++k) p->Jcsr[k] = 0.0;
    p->s = palloc(n);
    H->fcur = palloc(m); H->ftrial = palloc(m); H->u = palloc(m);
    H->dx = palloc(n); H->dtd = palloc(n); H->ux = palloc(n); H->v = palloc(n); H->h = palloc(n); H->hbar = palloc(n);
    H->P = palloc(n); H->tmp = palloc(n);
    return H;
}

void orc_omp_destroy(void *hh) {
    omp_handle *H = hh;
    if (!H) return;
    omp_prob *p = &H->p;
    free((void *)p->colptr); free((void *)p->rowval); free((void *)p->A);
    free(p->rowptr); free(p->colidx); free(p->Acsr); free(p->Jcsc); free(p->Jcsr); free(p->s);
    free(H->fcur); free(H->ftrial); free(H->u); free(H->dx); free(H->dtd); free(H->ux); free(H->v); free(H->h); free(H->hbar);
    free(H->P); free(H->tmp);
    free(H);
}

/* `iterations` LM outer iterations with tolerances 0 (bench.py's schedule) from x (overwritten with the iterate).
 * Returns 0; *inner_total = LSMR iterations, *ssr_out = final sum of squares. */
int orc_omp_run(void *hh, double *x, int iterations, long long *inner_total, double *ssr_out) {
    omp_handle *H = hh;
    omp_prob p = H->p;
    const int m = p.m, n = p.n;
    const int *colptr = p.colptr;
    omp_set_num_threads(H->threads);
    double *fcur = H->fcur, *ftrial = H->ftrial, *u = H->u, *dx = H->dx, *dtd = H->dtd, *ux = H->ux, *v = H->v, *h = H->h,
           *hbar = H->hbar, *P = H->P, *tmp = H->tmp;
    double delta = 10.0, decrease_factor = 2.0, maxabs_gr = 0.0;
    model_f(&p, x, fcur);
    double ssr = psumsq(fcur, m);
    int need_jac = 1;
    long long inner = 0;
    for (int iter = 0; iter < iterations; ++iter) {
        if (need_jac) { model_g(&p, x); need_jac = 0; }
        double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
        for (int j = 0; j < n; ++j) {
            double sq = 0.0;
            for (int k = colptr[j]; k < colptr[j + 1]; ++k) sq += p.Jcsc[k] * p.Jcsc[k];
            dtd[j] = sq;
            sum += sq;
        }
        const double mean = sum / n, lo = 1e-6 * mean, hi = 1e32 * mean, idl = 1.0 / delta;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; ++j) dtd[j] = (dtd[j] > hi ? hi : (dtd[j] < lo ? lo : dtd[j])) * idl;
        inner += lsmr_damped(&p, fcur, dtd, dx, u, ux, v, h, hbar, P, tmp);
        {   /* gradient J'f and its max-norm (levenberg_marquardt.jl:102-104): part of every iteration's work */
            csc_mulT(&p, p.Jcsc, fcur, tmp);
            double g = 0.0;
#pragma omp parallel for reduction(max : g) schedule(static)
            for (int j = 0; j < n; ++j) g = fabs(tmp[j]) > g ? fabs(tmp[j]) : g;
            maxabs_gr = g;
        }
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; ++j) x[j] -= dx[j];
        model_f(&p, x, ftrial);
        const double trial_ssr = psumsq(ftrial, m);
        csr_mul(&p, p.Jcsr, dx, 1.0, 0.0, u);                              /* fpredict = J dx - f */
        double pred = 0.0;
#pragma omp parallel for reduction(+ : pred) schedule(static)
        for (int i = 0; i < m; ++i) {
            const double t = u[i] - fcur[i];
            pred += t * t;
        }
        const double pr = fabs(ssr - pred);
        const double rho = pr > 0 ? (ssr - trial_ssr) / pr : 0.0;
        if (rho > 1e-3) {
            double *t = fcur; fcur = ftrial; ftrial = t;
            ssr = trial_ssr;
            const double q = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
            const double dn = delta / (1.0 / 3.0 > q ? 1.0 / 3.0 : q);
            delta = dn < 1e16 ? dn : 1e16;
            decrease_factor = 2.0;
            need_jac = 1;
        } else {
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; ++j) x[j] += dx[j];
            const double dn = delta / decrease_factor;
            delta = dn > 1e-16 ? dn : 1e-16;
            decrease_factor *= 2.0;
        }
    }
    H->fcur = fcur; H->ftrial = ftrial;
    if (inner_total) *inner_total = inner;
    if (ssr_out) *ssr_out = ssr;
    (void)maxabs_gr;
    return 0;
}
int orc_omp_max_threads(void) { return omp_get_max_threads(); }
