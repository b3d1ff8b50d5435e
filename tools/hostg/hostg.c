/* hostg.c -- a HOST-side g!(J, x) written in C against the public boundary (include/lsqhip.h), for bench.py's
 * `generic_g.host_g_pinned_async` leg: the reference's general sparse case, where g! rewrites nonzeros(J) on the host
 * (test/nonlinearleastsquares.jl:47-86) and the values cross PCIe after every accepted step
 * (levenberg_marquardt.jl:77-81).  The producer is plain C + OpenMP so that the leg measures the upload path
 * (lsq_mat_set_values_async: page-locked staging buffer, copy stream, device-side mirrors) and not an interpreter.
 *
 * Not part of the product library: it is a CONSUMER of the C ABI, like a Julia user's own g! would be.  f! stays on the
 * device (the built-in model's residual, reached through a forwarding callback, so that none of the library's
 * model-specific fast paths -- column-scaled handle, one-pass tail, speculative gradient -- can apply). */
#include <math.h>
#include <stddef.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "lsqhip.h"

typedef struct {
    lsq_ctx *ctx;
    lsq_f_callback inner_f;   /* lsq_model_f() */
    void *inner_user;         /* the lsq_model (created with LSQ_NO_COLSCALE=1: it keeps its own copy of A) */
    int n;
    long long nnz;
    const int *colptr;        /* n + 1 (host) */
    const double *A;          /* nnz (host): J(x) = A diag(1 - tanh(x)^2) */
    double *stage;            /* nnz doubles, page-locked (lsq_host_alloc) */
    double *xh;               /* n (host scratch) */
    int threads;
    /* statistics */
    int g_calls;
    double fill_seconds, g_seconds;
} hostg_t;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int hostg_f(double *d_out, const double *d_x, void *user) {
    hostg_t *h = (hostg_t *)user;
    return h->inner_f(d_out, d_x, h->inner_user);
}

int hostg_g(lsq_mat *J, const double *d_x, void *user) {
    hostg_t *h = (hostg_t *)user;
    const double t0 = now_s();
    /* the staging buffer may still be feeding the previous upload */
    if (lsq_mat_upload_wait(J) != LSQ_OK) return 1;
    if (lsq_d2h(h->ctx, h->xh, d_x, (size_t)h->n * sizeof(double)) != LSQ_OK) return 1;   /* x comes to the host (80 KB) */
    const double t1 = now_s();
    const int n = h->n;
    const int *cp = h->colptr;
    const double *A = h->A;
    double *out = h->stage;
    const double *xh = h->xh;
#pragma omp parallel for schedule(static) num_threads(h->threads)
    for (int j = 0; j < n; ++j) {
        const double t = tanh(xh[j]);
        const double s = 1.0 - t * t;
        for (int k = cp[j]; k < cp[j + 1]; ++k) out[k] = A[k] * s;
    }
    h->fill_seconds += now_s() - t1;
    if (lsq_mat_set_values_async(J, out) != LSQ_OK) return 1;   /* host returns at once; the device waits for the copy */
    h->g_calls += 1;
    h->g_seconds += now_s() - t0;
    return 0;
}

lsq_f_callback hostg_f_ptr(void) { return hostg_f; }
lsq_g_callback hostg_g_ptr(void) { return hostg_g; }
int hostg_sizeof(void) { return (int)sizeof(hostg_t); }
int hostg_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
