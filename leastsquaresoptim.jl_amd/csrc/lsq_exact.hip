// Reference-order ("exact") kernels for SMALL problems.
//
// On the reference's own test-suite sizes (n <= 40) LSMR runs on ill-conditioned operators well
// past the point where the Krylov basis has lost orthogonality; there the stop iteration depends
// on the last bit of every reduction, so two correct fp64 implementations of lsmr.jl:53-238 that
// differ only in summation order report different iteration counts.  For small problems the
// library therefore evaluates every sum in the SAME order and association as the reference's
// serial loops (left-to-right index order, SparseArrays' "scale by beta, then accumulate products
// column by column" for mul!), which makes iteration / mul / f / g counts -- and in fact every
// iterate -- reproduce the CPU restatement bit for bit.  Small problems are launch-latency bound,
// not bandwidth bound, so this costs nothing: the whole LSMR solve is ONE single-workgroup kernel
// (no per-iteration launches, no host polling).
#include <cmath>
#include <cstdlib>

#include "lsq_solver.h"
#include "lsq_spmv.h"

static int exact_mode() {  // LSQ_EXACT=0 disables (tests use it to exercise the fast kernels on small inputs)
    static int mode = [] {
        const char *e = getenv("LSQ_EXACT");
        return e ? atoi(e) : 1;
    }();
    return mode;
}
void lsq_exact_refresh_env();
static int g_exact_override = -1;
extern "C" int lsq_set_exact(int on) {  // -1: follow LSQ_EXACT / default
    g_exact_override = on;
    return LSQ_OK;
}
static bool exact_enabled() { return g_exact_override >= 0 ? g_exact_override != 0 : exact_mode() != 0; }

bool lsq_small_vec(long long n) { return exact_enabled() && n <= LSQ_EXACT_MAX_DIM; }
bool lsq_small_mat(const lsq_mat *J) {
    return J->kind != LSQ_MAT_OP && exact_enabled() && J->m <= LSQ_EXACT_MAX_DIM && J->n <= LSQ_EXACT_MAX_DIM && J->nnz <= LSQ_EXACT_MAX_NNZ;
}

// ---------------------------------------------------------------------------------------------
// products: one thread per segment, products accumulated left to right starting from 0.0
// (== SparseArrays mul! with alpha = 1, beta = 0 [stdlib]; SQ: utils.jl:139-151)
// ---------------------------------------------------------------------------------------------
template <bool SQ>
__global__ void __launch_bounds__(LSQ_NT)
k_seg_seq(int nseg, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
          const double *__restrict__ x, double *__restrict__ y) {
    for (int s = blockIdx.x * LSQ_NT + threadIdx.x; s < nseg; s += gridDim.x * LSQ_NT) {
        double acc = 0.0;
        for (int k = ptr[s]; k < ptr[s + 1]; ++k) acc += SQ ? val[k] * val[k] : val[k] * x[idx[k]];
        y[s] = acc;
    }
}
// dense J*x: row i accumulates column by column (BLAS-2 'N' as column axpys, like the oracle)
__global__ void __launch_bounds__(LSQ_NT)
k_dense_seq_n(const double *__restrict__ A, int m, int n, const double *__restrict__ x, double *__restrict__ y) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < m; i += gridDim.x * LSQ_NT) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += A[(size_t)j * m + i] * x[j];
        y[i] = acc;
    }
}
template <bool SQ>
__global__ void __launch_bounds__(LSQ_NT)
k_dense_seq_t(const double *__restrict__ A, int m, int n, const double *__restrict__ yv, double *__restrict__ x) {
    for (int j = blockIdx.x * LSQ_NT + threadIdx.x; j < n; j += gridDim.x * LSQ_NT) {
        const double *col = A + (size_t)j * m;
        double acc = 0.0;
        for (int i = 0; i < m; ++i) acc += SQ ? col[i] * col[i] : col[i] * yv[i];
        x[j] = acc;
    }
}

// general mul!(y, J, x, alpha, beta) / mul!(x, J', y, alpha, beta) in the reference's order:
// SparseArrays scales y by beta (fill for beta == 0), then accumulates nz * (x[col] * alpha) column by
// column; the adjoint forms the column dot from 0 and adds dot * alpha.
template <bool TRANS>
__global__ void __launch_bounds__(LSQ_NT)
k_seg_seq_ab(int nseg, const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
             const double *__restrict__ x, double alpha, double beta, double *__restrict__ y) {
    for (int s = blockIdx.x * LSQ_NT + threadIdx.x; s < nseg; s += gridDim.x * LSQ_NT) {
        const double y0 = (beta == 0.0) ? 0.0 : (beta == 1.0 ? y[s] : y[s] * beta);
        if (TRANS) {
            double t = 0.0;
            for (int k = ptr[s]; k < ptr[s + 1]; ++k) t += val[k] * x[idx[k]];
            y[s] = y0 + t * alpha;
        } else {
            double acc = y0;
            for (int k = ptr[s]; k < ptr[s + 1]; ++k) acc += val[k] * (x[idx[k]] * alpha);
            y[s] = acc;
        }
    }
}
template <bool TRANS>
__global__ void __launch_bounds__(LSQ_NT)
k_dense_seq_ab(const double *__restrict__ A, int m, int n, const double *__restrict__ x, double alpha, double beta,
               double *__restrict__ y) {
    const int nseg = TRANS ? n : m;
    for (int s = blockIdx.x * LSQ_NT + threadIdx.x; s < nseg; s += gridDim.x * LSQ_NT) {
        const double y0 = (beta == 0.0) ? 0.0 : (beta == 1.0 ? y[s] : y[s] * beta);
        if (TRANS) {
            const double *col = A + (size_t)s * m;
            double t = 0.0;
            for (int i = 0; i < m; ++i) t += col[i] * x[i];
            y[s] = y0 + t * alpha;
        } else {
            double acc = y0;
            for (int j = 0; j < n; ++j) acc += A[(size_t)j * m + s] * (x[j] * alpha);
            y[s] = acc;
        }
    }
}
int lsq_exact_mul(lsq_mat *J, int trans, double alpha, const double *x, double beta, double *y) {
    lsq_ctx *c = J->ctx;
    const int nseg = trans ? J->n : J->m;
    if (nseg <= 0) return LSQ_OK;
    const int grid = lsq_div_up(nseg, LSQ_NT);
    if (J->kind == LSQ_MAT_CSC) {
        if (!trans) {
            LSQ_TRY(lsq_ensure_csr(J));
            LSQ_LAUNCH((k_seg_seq_ab<false>), dim3(grid), dim3(LSQ_NT), 0, c->stream, nseg, J->csr.d_ptr,
                               J->csr.d_idx, J->csr.d_val, x, alpha, beta, y);
        } else {
            LSQ_LAUNCH((k_seg_seq_ab<true>), dim3(grid), dim3(LSQ_NT), 0, c->stream, nseg, J->csc.d_ptr,
                               J->csc.d_idx, J->csc.d_val, x, alpha, beta, y);
        }
    } else if (!trans) {
        LSQ_LAUNCH((k_dense_seq_ab<false>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x,
                           alpha, beta, y);
    } else {
        LSQ_LAUNCH((k_dense_seq_ab<true>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x,
                           alpha, beta, y);
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_exact_product(lsq_mat *J, int trans, const double *x, double *y) {
    lsq_ctx *c = J->ctx;
    const int nseg = trans ? J->n : J->m;
    if (nseg <= 0) return LSQ_OK;
    const int grid = lsq_div_up(nseg, LSQ_NT);
    if (J->kind == LSQ_MAT_CSC) {
        if (!trans) LSQ_TRY(lsq_ensure_csr(J));
        const LsqSegs &S = trans ? J->csc : J->csr;
        LSQ_LAUNCH((k_seg_seq<false>), dim3(grid), dim3(LSQ_NT), 0, c->stream, nseg, S.d_ptr, S.d_idx, S.d_val,
                           x, y);
    } else if (!trans) {
        LSQ_LAUNCH(k_dense_seq_n, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x, y);
    } else {
        LSQ_LAUNCH((k_dense_seq_t<false>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x, y);
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_exact_colsumabs2(lsq_mat *J, double *out) {
    lsq_ctx *c = J->ctx;
    if (J->n <= 0) return LSQ_OK;
    const int grid = lsq_div_up(J->n, LSQ_NT);
    if (J->kind == LSQ_MAT_CSC)
        LSQ_LAUNCH((k_seg_seq<true>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->n, J->csc.d_ptr, J->csc.d_idx,
                           J->csc.d_val, (const double *)nullptr, out);
    else
        LSQ_LAUNCH((k_dense_seq_t<true>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n,
                           (const double *)nullptr, out);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// sequential reductions (one thread; n <= LSQ_EXACT_MAX_DIM)
//   0 sum(x)   1 sum(x^2)   2 sum(w*x*y) (wdot, utils.jl:165-175)   3 sum((x - y)^2)
// ---------------------------------------------------------------------------------------------
// The TERMS are formed by all threads (element-wise: same roundings as in a loop) and parked in LDS; one thread then adds
// them left to right.  (A one-thread loop over global memory paid a memory latency per element: 347 us for n = 2048.)
__global__ void __launch_bounds__(256) k_seq_reduce(int mode, int n, const double *x, const double *y, const double *w, double *out) {
    __shared__ double term[2048];
    double acc = 0.0;
    for (int base = 0; base < n; base += 2048) {
        const int cnt = min(2048, n - base);
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int i = base + k;
            double t;
            if (mode == 0) t = x[i];
            else if (mode == 1) t = x[i] * x[i];
            else if (mode == 2) t = w ? w[i] * x[i] * y[i] : x[i] * y[i];   // wdot (utils.jl:165-172) / dot
            else { const double r = x[i] + -1.0 * y[i]; t = r * r; }        // axpy!(-1, fcur, fpredict) then abs2
            term[k] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < cnt; ++k) acc += term[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = acc;
}
int lsq_seq_reduce(lsq_ctx *c, int mode, int n, const double *x, const double *y, const double *w, double *d_out) {
    LSQ_LAUNCH(k_seq_reduce, dim3(1), dim3(256), 0, c->stream, mode, n, x, y, w, d_out);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// LM damping with the mean formed sequentially (levenberg_marquardt.jl:84-86)
__global__ void __launch_bounds__(LSQ_NT)
k_lm_damp_seq(int n, const double *__restrict__ colsum, double inv_delta, double *__restrict__ dtd) {
    __shared__ double s_mean;
    __shared__ double cs_l[LSQ_EXACT_MAX_DIM];   // (n <= LSQ_EXACT_MAX_DIM here: staged, then added left to right)
    for (int i = threadIdx.x; i < n; i += LSQ_NT) cs_l[i] = colsum[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += cs_l[i];
        s_mean = t / n;
    }
    __syncthreads();
    const double lo = 1e-6 * s_mean, hi = 1e32 * s_mean;
    for (int i = threadIdx.x; i < n; i += LSQ_NT) {
        double v = colsum[i];
        v = v > hi ? hi : (v < lo ? lo : v);
        dtd[i] = v * inv_delta;
    }
}
int lsq_exact_lm_damp(lsq_ctx *c, int n, const double *colsum, double inv_delta, double *dtd) {
    LSQ_LAUNCH(k_lm_damp_seq, dim3(1), dim3(LSQ_NT), 0, c->stream, n, colsum, inv_delta, dtd);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// whole LSMR solve in one workgroup, reference order (lsmr.jl:53-238 through the wrappers of
// iterative_lsmr.jl:12-122; the CPU restatement used by the tests performs the same sequence)
// ---------------------------------------------------------------------------------------------
struct ExactMat {
    int dense, m, n;
    const double *A;                          // dense column-major
    const int *rptr, *cidx; const double *rval;   // CSR rows
    const int *cptr, *ridx; const double *cval;   // CSC columns
};

constexpr int EX_NT = 256;
__device__ __forceinline__ double seq_sumsq(const double *x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    return s;
}

__global__ void __launch_bounds__(EX_NT)
k_lsmr_exact(ExactMat M, const double *__restrict__ y, const double *__restrict__ colsum, double *damp,
             double *__restrict__ xout, double *u, double *ux, double *v, double *h, double *hbar, double *tmp,
             double *tmp2, double *P, double *xs, double atol, double btol, double ctol, int maxiter,
             int *result /* iter, istop */) {
    __shared__ double s_a, s_b;   // broadcast scalars
    __shared__ int s_flag;
    const int tid = threadIdx.x, m = M.m, n = M.n;
    const bool damped = damp != nullptr;
    // preconditioner and sqrt(damp) (iterative_lsmr.jl:129-141, :252)
    for (int j = tid; j < n; j += EX_NT) {
        double s = colsum[j];
        if (damped) {
            double d = damp[j];
            s += 1.0 * d;
            damp[j] = sqrt(d);
            ux[j] = 0.0;
        }
        P[j] = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
        xs[j] = 0.0;
    }
    for (int i = tid; i < m; i += EX_NT) u[i] = y[i];
    __syncthreads();
    const double *dg = damp;
    // helpers -----------------------------------------------------------------------------
    auto mulT = [&](double scale_v /* v <- scale_v*v + P.*(J'u + ux.*dg) ; 0 => fill */) {
        for (int j = tid; j < n; j += EX_NT) {
            double t = 0.0;
            if (M.dense) {
                const double *col = M.A + (size_t)j * m;
                for (int i = 0; i < m; ++i) t += col[i] * u[i];
            } else {
                for (int k = M.cptr[j]; k < M.cptr[j + 1]; ++k) t += M.cval[k] * u[M.ridx[k]];
            }
            // adjoint mul!: x[j] += t * alpha with alpha = 1, after fill (x = 0) or beta = 1 on a zeroed tmp
            t = 0.0 + t * 1.0;
            if (damped) t = t + 1.0 * ux[j] * dg[j];          // iterative_lsmr.jl:107
            double t2 = t * P[j];                             // :41
            double vj = (scale_v == 0.0) ? 0.0 : v[j] * scale_v;   // :42-48
            v[j] = vj + 1.0 * t2;                             // :49
        }
        __syncthreads();
    };
    auto norm_u = [&]() {  // DampenedVector norm (il:72) / norm(u)
        if (tid == 0) {
            double sy = seq_sumsq(u, m);
            if (damped) {   // sqrt(norm(y)^2 + norm(x)^2), literally (il:72)
                const double ny = sqrt(sy), nx = sqrt(seq_sumsq(ux, n));
                s_a = sqrt(ny * ny + nx * nx);
            } else {
                s_a = sqrt(sy);
            }
        }
        __syncthreads();
        double r = s_a;
        __syncthreads();
        return r;
    };
    auto norm_n = [&](const double *z) {
        if (tid == 0) s_b = sqrt(seq_sumsq(z, n));
        __syncthreads();
        double r = s_b;
        __syncthreads();
        return r;
    };
    auto scale_u = [&](double a) {
        for (int i = tid; i < m; i += EX_NT) u[i] *= a;
        if (damped)
            for (int j = tid; j < n; j += EX_NT) ux[j] *= a;
        __syncthreads();
    };
    // u = b - A*0 = b; beta, u/beta, v = A'u, alpha, v/alpha (lsmr.jl:73-78)
    double beta = norm_u();
    if (beta > 0) scale_u(1.0 / beta);
    mulT(0.0);
    double alpha = norm_n(v);
    if (alpha > 0) {
        const double ia = 1.0 / alpha;
        for (int j = tid; j < n; j += EX_NT) v[j] *= ia;
        __syncthreads();
    }
    // state (every thread keeps an identical copy; only thread-uniform control flow below)
    LsmrState s;
    s.zetabar = alpha * beta; s.alphabar = alpha; s.rho = 1.0; s.rhobar = 1.0; s.cbar = 1.0; s.sbar = 0.0;
    for (int j = tid; j < n; j += EX_NT) { h[j] = v[j]; hbar[j] = 0.0; }
    __syncthreads();
    s.betadd = beta; s.betad = 0.0; s.rhodold = 1.0; s.tautildeold = 0.0; s.thetatilde = 0.0; s.zeta = 0.0; s.d = 0.0;
    s.normA = -1.0; s.condA = -1.0; s.normx = -1.0;
    s.normA2 = alpha * alpha; s.maxrbar = 0.0; s.minrbar = 1e100;
    s.normb = beta; s.normr = beta; s.normAr = alpha * beta;
    s.iter = 0;
    int istop = 0, iter = 0;
    if (s.normAr != 0) {
        while (iter < maxiter) {
            iter += 1;
            // u <- A v - alpha u (lsmr.jl:118): tmp = v.*P; rmul!(u, -alpha); u_y += J tmp; u_x += tmp.*dg
            for (int j = tid; j < n; j += EX_NT) tmp[j] = v[j] * P[j];
            __syncthreads();
            const double na = -alpha;
            for (int i = tid; i < m; i += EX_NT) {
                double acc = u[i] * na;
                if (M.dense) {
                    for (int j = 0; j < n; ++j) acc += M.A[(size_t)j * m + i] * (tmp[j] * 1.0);
                } else {
                    for (int k = M.rptr[i]; k < M.rptr[i + 1]; ++k) acc += M.rval[k] * (tmp[M.cidx[k]] * 1.0);
                }
                u[i] = acc;
            }
            if (damped)
                for (int j = tid; j < n; j += EX_NT) {
                    double b = ux[j] * na;
                    ux[j] = b + 1.0 * tmp[j] * dg[j];
                }
            __syncthreads();
            beta = norm_u();
            if (beta > 0) {
                scale_u(1.0 / beta);
                mulT(-beta);                                   // lsmr.jl:122
                alpha = norm_n(v);
                if (alpha > 0) {
                    const double ia = 1.0 / alpha;
                    for (int j = tid; j < n; j += EX_NT) v[j] *= ia;
                    __syncthreads();
                }
            }
            lsmr_rotate_inline(s, alpha, beta);
            s.iter = iter;
            // lsmr.jl:152-156
            const double c1 = s.c1, c2 = s.c2, c3 = s.c3;
            for (int j = tid; j < n; j += EX_NT) {
                double hb = hbar[j] * c1;
                hb += 1.0 * h[j];
                hbar[j] = hb;
                xs[j] += c2 * hb;
                double hj = h[j] * c3;
                hj += 1.0 * v[j];
                h[j] = hj;
            }
            __syncthreads();
            s.normx = norm_n(xs);
            double test1 = s.normr / s.normb;
            double test2 = s.normAr / (s.normA * s.normr);
            double test3 = 1.0 / s.condA;
            double t1 = test1 / (1.0 + s.normA * s.normx / s.normb);
            double rtol = btol + atol * s.normA * s.normx / s.normb;
            if (iter >= maxiter) { istop = 7; break; }
            if (1.0 + test3 <= 1.0) { istop = 6; break; }
            if (1.0 + test2 <= 1.0) { istop = 5; break; }
            if (1.0 + t1 <= 1.0) { istop = 4; break; }
            if (test3 <= ctol) { istop = 3; break; }
            if (test2 <= atol) { istop = 2; break; }
            if (test1 <= rtol) { istop = 1; break; }
        }
    }
    for (int j = tid; j < n; j += EX_NT) xout[j] = xs[j] * P[j];   // iterative_lsmr.jl:195-196, 256-257
    if (tid == 0) { result[0] = iter; result[1] = istop; }
    (void)s_flag;
}

int lsq_lsmr_exact_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n;
    const bool damped = d_damp != nullptr;
    const double atol = 1e-6, btol = damped ? 0.5 : 1e-6, conlim = 1e8;
    const long long rows = damped ? (long long)m + n : m;
    const int maxiter = (int)std::max<long long>(rows, n);
    if (J->kind == LSQ_MAT_CSC) LSQ_TRY(lsq_ensure_csr(J));
    const double *colsum = lsq_cached_colsum(J);
    if (!colsum) return LSQ_EHIP;
    ExactMat M{J->kind == LSQ_MAT_DENSE, m, n, J->d_dense, J->csr.d_ptr, J->csr.d_idx, J->csr.d_val,
               J->csc.d_ptr, J->csc.d_idx, J->csc.d_val};
    int *res = (int *)(s->d_red);  // reuse the reduction scratch for {iter, istop}
    LSQ_LAUNCH(k_lsmr_exact, dim3(1), dim3(EX_NT), 0, c->stream, M, d_y, colsum, d_damp, d_x, s->d_u, s->d_ux,
                       s->d_v, s->d_h, s->d_hbar, s->d_t, s->d_dg /* tmp2 */, s->d_P, s->d_rhs, atol, btol,
                       1.0 / conlim, maxiter, res);
    LSQ_HIP(hipGetLastError());
    int h[2] = {0, 0};
    LSQ_HIP(hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    s->last_iter = h[0];
    s->last_istop = h[1];
    if (nmul) *nmul = 2 * h[0];
    return LSQ_OK;
}
