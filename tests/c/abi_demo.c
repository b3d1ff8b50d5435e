/* A plain C99 consumer of the drop-in boundary (include/lsqhip.h, include/lsqrccl.h): what a binding in any language
 * with a C FFI sees.  Compiled by tests/test_host.py with `gcc -std=c99 -pedantic -Wall -Werror` against liblsqhip.so.
 *
 *   abi_demo            no device needed: version, error string, default options, struct layout -- the part of the
 *                       "library loads and exports what the header declares" check that Python's ctypes cannot make
 *                       (that the header itself is valid C and agrees with the library about struct sizes)
 *   abi_demo --gpu      on an MI355X: one ldiv! of every dense solver and of LSMR on a small full-rank least-squares
 *                       problem (dense_qr.jl:30-42, dense_cholesky.jl:43-59, iterative_lsmr.jl:179-259 behind lsq_ldiv),
 *                       checked by the normal equations J'(J x - y) = 0 -- no oracle involved.
 * Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsqhip.h"
#include "lsqrccl.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int st__ = (call);                                                                       \
        if (st__ != LSQ_OK) {                                                                    \
            fprintf(stderr, "%s -> status %d: %s\n", #call, st__, lsq_last_error());             \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static double lcg(unsigned long long *s) {          /* uniform in (-1, 1) */
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((*s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 52) - 1.0;
}

static int gpu_part(void) {
    enum { M = 300, N = 7 };
    static double A[M * N], y[M], x[N], r[M];
    unsigned long long seed = 42;
    int i, j, k;
    lsq_ctx *ctx = NULL;
    lsq_mat *J = NULL;
    double *d_y = NULL, *d_x = NULL;
    const int kinds[3] = {LSQ_QR, LSQ_CHOLESKY, LSQ_LSMR};
    const char *names[3] = {"QR", "Cholesky", "LSMR"};
    for (i = 0; i < M * N; ++i) A[i] = lcg(&seed);                /* column-major */
    for (i = 0; i < M; ++i) y[i] = lcg(&seed);
    CHECK(lsq_ctx_create(0, NULL, &ctx));
    CHECK(lsq_dense_create(ctx, M, N, &J));
    CHECK(lsq_mat_set_values(J, A));
    CHECK(lsq_malloc(ctx, M * sizeof(double), (void **)&d_y));
    CHECK(lsq_malloc(ctx, N * sizeof(double), (void **)&d_x));
    CHECK(lsq_h2d(ctx, d_y, y, M * sizeof(double)));
    for (k = 0; k < 3; ++k) {
        lsq_solver *sv = NULL;
        int nmul = -1;
        double worst = 0.0, scale = 0.0;
        CHECK(lsq_solver_create(ctx, J, kinds[k], 0, &sv));
        CHECK(lsq_ldiv(sv, J, d_y, d_x, &nmul));
        CHECK(lsq_d2h(ctx, x, d_x, N * sizeof(double)));
        for (i = 0; i < M; ++i) {
            double s = -y[i];
            for (j = 0; j < N; ++j) s += A[(size_t)j * M + i] * x[j];
            r[i] = s;
        }
        for (j = 0; j < N; ++j) {                                 /* J'(J x - y) against |J'y| */
            double g = 0.0, b = 0.0;
            for (i = 0; i < M; ++i) { g += A[(size_t)j * M + i] * r[i]; b += A[(size_t)j * M + i] * y[i]; }
            if (fabs(g) > worst) worst = fabs(g);
            if (fabs(b) > scale) scale = fabs(b);
        }
        printf("%-8s nmul %3d  max|J'(Jx - y)| / max|J'y| = %.2e\n", names[k], nmul, worst / scale);
        if (!(worst <= (k == 2 ? 1e-5 : 1e-11) * scale)) {         /* LSMR stops at atol = btol = 1e-6 (lsmr.jl:54) */
            fprintf(stderr, "%s: normal equations violated\n", names[k]);
            return 1;
        }
        CHECK(lsq_solver_destroy(sv));
    }
    CHECK(lsq_free(ctx, d_y));
    CHECK(lsq_free(ctx, d_x));
    CHECK(lsq_mat_destroy(J));
    CHECK(lsq_ctx_destroy(ctx));
    return 0;
}

int main(int argc, char **argv) {
    lsq_options opt;
    lsq_result res;
    memset(&res, 0, sizeof res);
    printf("lsq_version %d\n", lsq_version());
    if (lsq_version() <= 0) return 1;
    if (lsq_last_error() == NULL) return 1;
    lsq_options_default(&opt);
    /* the reference's defaults: levenberg_marquardt.jl:41, dogleg.jl:43 */
    if (opt.x_tol != 1e-8 || opt.f_tol != 1e-8 || opt.g_tol != 1e-8 || opt.iterations != 1000) {
        fprintf(stderr, "lsq_options_default: unexpected defaults\n");
        return 1;
    }
    /* struct sizes as THIS translation unit sees them; tests/test_host.py compares them with the ctypes mirrors */
    printf("sizeof lsq_options %lu\nsizeof lsq_result %lu\n", (unsigned long)sizeof(lsq_options), (unsigned long)sizeof(lsq_result));
    if (lsq_debug_set(-1, -1) != LSQ_OK) return 1;   /* (a no-op call through the ABI) */
    if (argc > 1 && strcmp(argv[1], "--gpu") == 0) return gpu_part();
    return 0;
}
