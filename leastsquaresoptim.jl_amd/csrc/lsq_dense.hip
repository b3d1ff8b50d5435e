// Dense Jacobian path: GEMVs / colsumabs2, the normal-equation + Cholesky solver
// (dense_cholesky.jl:29-59) and the column-pivoted Householder QR solver with the rank-revealing
// minimum-norm solve (dense_qr.jl:30-88; LinearAlgebra.ldiv!(::QRPivoted, b) [stdlib] = xGELSY).
// Everything here runs on the device; the host only reads back status words.
#include <type_traits>
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "lsq_qr_cholqr.h"
#include "lsq_solver.h"
#include "lsq_spmv.h"

int lsq_cholesky_blocked(lsq_solver *s, lsq_mat *J, const double *d_damp, double *d_x, double *d_dmax);
int lsq_cholesky_blocked_solve(lsq_solver *s, int n, double *d_x);
int lsq_tri_inv_fro2(lsq_solver *s, const double *U, int n, double *fro2_inv);
void lsq_tri_pipe_err_copy(lsq_solver *s, int *h_dst);
void lsq_tri_pipe_disable(lsq_solver *s);  // lsq_dense_mfma.hip

// ---------------------------------------------------------------------------------------------
// generic dense products
// ---------------------------------------------------------------------------------------------
struct EpiAxpbyD {
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double alpha, beta;
    double *y;
    double *partials;
    unsigned *counter;
    __device__ void seg(int s, double dot, double &) const {
        y[s] = (beta == 0.0) ? alpha * dot : alpha * dot + beta * y[s];
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

int lsq_dense_mul(lsq_mat *J, int trans, double alpha, const double *x, double beta, double *y) {
    EpiAxpbyD e{nullptr, 0, alpha, beta, y, nullptr, nullptr};
    return launch_product(J, trans, x, e);
}

struct EpiStoreD {
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int s, double dot, double &) const { out[s] = dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

int lsq_dense_part(lsq_mat *J, int nwin) {
    const int need = nwin * J->n;
    if (J->dpart_cap < need) {
        hipFree(J->d_dpart);
        J->d_dpart = nullptr;
        LSQ_HIP(hipMalloc(&J->d_dpart, ((size_t)need + 8) * sizeof(double)));
        J->dpart_cap = need;
    }
    return LSQ_OK;
}

int lsq_dense_colsumabs2(lsq_mat *J, double *out) {  // utils.jl:139-144
    if (J->n <= 0) return LSQ_OK;
    EpiStoreD e{nullptr, 0, out, nullptr, nullptr};
    if (const int nwin = lsq_dense_t_windows(J->ctx, J->m, J->n)) {   // few columns: (window, column) blocks + combine
        LSQ_TRY(lsq_dense_part(J, nwin));
        const int wrows = ((J->m + nwin - 1) / nwin + 3) / 4 * 4;
        hipLaunchKernelGGL((k_dense_t_win<true>), dim3(nwin * J->n), dim3(LSQ_NT), 0, J->ctx->stream, J->d_dense, J->m, J->n,
                           (const double *)nullptr, wrows, J->d_dpart, (const int *)nullptr);
        const int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
        hipLaunchKernelGGL((k_combine<EpiStoreD>), dim3(nb), dim3(LSQ_NT), 0, J->ctx->stream, J->d_dpart, J->n, nwin, e, nb);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    int grid = J->n > LSQ_MAX_GRID ? LSQ_MAX_GRID : J->n;
    hipLaunchKernelGGL((k_dense_t<EpiStoreD, true>), dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream,
                       J->d_dense, J->m, J->n, nullptr, e);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// normal equations  C = J'J (+ diag damp), upper triangle  (dense_cholesky.jl:31,48,51-53)
// ---------------------------------------------------------------------------------------------
constexpr int SY_T = 32;   // output tile
constexpr int SY_K = 32;   // k-chunk
__global__ void __launch_bounds__(256)
k_syrk_upper(const double *__restrict__ A, int m, int n, double *__restrict__ C, const double *__restrict__ damp) {
    // tile (bi, bj) with bj >= bi, enumerated linearly
    __shared__ double sa[SY_T][SY_K + 1];
    __shared__ double sb[SY_T][SY_K + 1];
    const int nt = (n + SY_T - 1) / SY_T;
    int t = blockIdx.x, bi = 0;
    while (t >= nt - bi) { t -= nt - bi; ++bi; }
    const int bj = bi + t;
    const int i0 = bi * SY_T, j0 = bj * SY_T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16x16 threads, 2x2 outputs each
    double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
    for (int k0 = 0; k0 < m; k0 += SY_K) {
        // 32 columns x 32 k per operand = 1024 elements, 4 per thread, coalesced along k
        for (int e = threadIdx.x; e < SY_T * SY_K; e += 256) {
            int col = e / SY_K, kk = e % SY_K;
            int k = k0 + kk;
            sa[col][kk] = (k < m && i0 + col < n) ? A[(size_t)(i0 + col) * m + k] : 0.0;
            sb[col][kk] = (k < m && j0 + col < n) ? A[(size_t)(j0 + col) * m + k] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < SY_K; ++kk) {
            double a0 = sa[ty][kk], a1 = sa[ty + 16][kk];
            double b0 = sb[tx][kk], b1 = sb[tx + 16][kk];
            c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
        }
        __syncthreads();
    }
    auto put = [&](int i, int j, double v) {
        if (i < n && j < n && i <= j) {
            if (i == j && damp) v += damp[i];
            C[(size_t)j * n + i] = v;
        }
    };
    put(i0 + ty, j0 + tx, c00);
    put(i0 + ty, j0 + tx + 16, c01);
    put(i0 + ty + 16, j0 + tx, c10);
    put(i0 + ty + 16, j0 + tx + 16, c11);
}

// ---------------------------------------------------------------------------------------------
// single-workgroup Cholesky (upper) + solve.  PIVOT = false: dpotf2 (cholesky!(Symmetric(C)),
// dense_cholesky.jl:57); PIVOT = true: dpstf2 with tol = 0 (cholesky!(.., Val(true)), :33).
// info: 0 ok; k > 0 not positive definite at k (unpivoted) / rank deficient with rank k-1... see
// below.  The right-hand side is solved in place (U'U x = b, with the permutation for PIVOT).
// ---------------------------------------------------------------------------------------------
constexpr int CH_NT = 1024;
template <bool PIVOT>
__global__ void __launch_bounds__(CH_NT)
k_chol_solve(double *__restrict__ A, int n, double *__restrict__ b, int *__restrict__ info,
             int *__restrict__ piv, double *__restrict__ work /* 2n */, double *__restrict__ tmp /* n */) {
    __shared__ double sh[CH_NT / 64];
    __shared__ double s_ajj;
    __shared__ int s_pvt;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) s_fail = 0;
    if (PIVOT) {
        for (int i = tid; i < n; i += CH_NT) { piv[i] = i; work[i] = 0.0; }
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double *cj = A + (size_t)j * n;
        double ajj;
        if (PIVOT) {
            // dpstf2: running column dot products, pivot = argmax of the updated diagonal
            for (int i = j + tid; i < n; i += CH_NT) {
                if (j > 0) { double a = A[(size_t)i * n + (j - 1)]; work[i] += a * a; }
                work[n + i] = A[(size_t)i * n + i] - work[i];
            }
            __syncthreads();
            if (tid == 0) {
                int p = j;
                for (int i = j + 1; i < n; ++i)
                    if (work[n + i] > work[n + p]) p = i;
                s_pvt = p;
                s_ajj = work[n + p];
                if (s_ajj <= 0.0 || isnan(s_ajj)) { s_fail = 1; *info = j + 1; }
            }
            __syncthreads();
            if (s_fail) return;
            const int p = s_pvt;
            if (p != j) {
                if (tid == 0) A[(size_t)p * n + p] = A[(size_t)j * n + j];
                for (int i = tid; i < j; i += CH_NT) {
                    double t = A[(size_t)j * n + i]; A[(size_t)j * n + i] = A[(size_t)p * n + i]; A[(size_t)p * n + i] = t;
                }
                for (int k = p + 1 + tid; k < n; k += CH_NT) {
                    double t = A[(size_t)k * n + j]; A[(size_t)k * n + j] = A[(size_t)k * n + p]; A[(size_t)k * n + p] = t;
                }
                for (int i = j + 1 + tid; i < p; i += CH_NT) {
                    double t = A[(size_t)i * n + j]; A[(size_t)i * n + j] = A[(size_t)p * n + i]; A[(size_t)p * n + i] = t;
                }
                if (tid == 0) {
                    double t = work[j]; work[j] = work[p]; work[p] = t;
                    int ti = piv[p]; piv[p] = piv[j]; piv[j] = ti;
                }
            }
            __syncthreads();
            ajj = sqrt(s_ajj);
            if (tid == 0) cj[j] = ajj;
        } else {
            double acc = 0.0;
            for (int i = tid; i < j; i += CH_NT) acc += cj[i] * cj[i];
            double tot = 0.0;
            {
                double v = wave_sum(acc);
                if ((tid & 63) == 0) sh[tid >> 6] = v;
                __syncthreads();
                if (tid == 0) {
                    for (int w = 0; w < CH_NT / 64; ++w) tot += sh[w];
                    double d = cj[j] - tot;
                    s_ajj = d;
                    if (d <= 0.0 || isnan(d)) { s_fail = 1; *info = j + 1; cj[j] = d; }
                    else cj[j] = sqrt(d);
                }
                __syncthreads();
            }
            if (s_fail) return;
            ajj = sqrt(s_ajj);
        }
        // row j of U: A[j,k] = (A[j,k] - sum_{i<j} A[i,j] A[i,k]) / ajj   for k > j
        // one wave per column k so the dot product reads are contiguous
        const int lane = tid & 63, w = tid >> 6;
        for (int k = j + 1 + w; k < n; k += CH_NT / 64) {
            double *ck = A + (size_t)k * n;
            double acc = 0.0;
            for (int i = lane; i < j; i += 64) acc += cj[i] * ck[i];
            acc = wave_sum(acc);
            if (lane == 0) ck[j] = (ck[j] - acc) / ajj;
        }
        __syncthreads();
    }
    // solve U'U x = P'b
    double *z = b;
    if (PIVOT) {
        for (int i = tid; i < n; i += CH_NT) tmp[i] = b[piv[i]];  // permute!(B, piv)
        __syncthreads();
        z = tmp;
    }
    // forward: U' y = z  (row i of U' = column i of U: contiguous)
    for (int i = 0; i < n; ++i) {
        const double *ci = A + (size_t)i * n;
        double acc = 0.0;
        for (int k = tid; k < i; k += CH_NT) acc += ci[k] * z[k];
        double v = wave_sum(acc);
        if ((tid & 63) == 0) sh[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < CH_NT / 64; ++w) tot += sh[w];
            z[i] = (z[i] - tot) / ci[i];
        }
        __syncthreads();
    }
    // backward: U x = y, column-oriented
    for (int i = n - 1; i >= 0; --i) {
        const double *ci = A + (size_t)i * n;
        if (tid == 0) z[i] = z[i] / ci[i];
        __syncthreads();
        const double zi = z[i];
        for (int k = tid; k < i; k += CH_NT) z[k] -= zi * ci[k];
        __syncthreads();
    }
    if (PIVOT) {
        for (int i = tid; i < n; i += CH_NT) b[piv[i]] = tmp[i];  // invpermute!
    }
    if (tid == 0 && !s_fail) *info = 0;
}

// ---------------------------------------------------------------------------------------------
// single-workgroup column-pivoted Householder QR (dgeqp3 semantics via the dlaqp2 recurrence)
// followed by the xGELSY solve.  A is M x n (lda = M), b has length >= max(M, n).
// ---------------------------------------------------------------------------------------------
constexpr int QR_NT = 1024;

__device__ __forceinline__ double blk_sum_qr(double v, double *sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < QR_NT / 64; ++w) r += sh[w];
    __syncthreads();
    return r;  // every thread gets the total
}

// LAPACK dlaic1 (incremental condition estimation); alpha = x'w is supplied by the caller.
__device__ void laic1_dev(int job, double alpha, double sest, double gamma, double *sestpr, double *s,
                          double *c) {
    const double eps = DBL_EPSILON / 2;
    double absalp = fabs(alpha), absgam = fabs(gamma), absest = fabs(sest);
    double s1, s2, tmp, b, cc, t, zeta1, zeta2, sine, cosine;
    if (job == 1) {
        if (sest == 0.0) {
            s1 = fmax(absgam, absalp);
            if (s1 == 0.0) { *s = 0; *c = 1; *sestpr = 0; }
            else { *s = alpha / s1; *c = gamma / s1; tmp = sqrt(*s * *s + *c * *c); *s /= tmp; *c /= tmp; *sestpr = s1 * tmp; }
        } else if (absgam <= eps * absest) {
            *s = 1; *c = 0; tmp = fmax(absest, absalp); s1 = absest / tmp; s2 = absalp / tmp;
            *sestpr = tmp * sqrt(s1 * s1 + s2 * s2);
        } else if (absalp <= eps * absest) {
            s1 = absgam; s2 = absest;
            if (s1 <= s2) { *s = 1; *c = 0; *sestpr = s2; } else { *s = 0; *c = 1; *sestpr = s1; }
        } else if (absest <= eps * absalp || absest <= eps * absgam) {
            s1 = absgam; s2 = absalp;
            if (s1 <= s2) { tmp = s1 / s2; *s = sqrt(1 + tmp * tmp); *sestpr = s2 * *s; *c = (gamma / s2) / *s; *s = copysign(1.0, alpha) / *s; }
            else { tmp = s2 / s1; *c = sqrt(1 + tmp * tmp); *sestpr = s1 * *c; *s = (alpha / s1) / *c; *c = copysign(1.0, gamma) / *c; }
        } else {
            zeta1 = alpha / absest; zeta2 = gamma / absest;
            b = (1 - zeta1 * zeta1 - zeta2 * zeta2) * 0.5; cc = zeta1 * zeta1;
            t = b > 0 ? cc / (b + sqrt(b * b + cc)) : sqrt(b * b + cc) - b;
            sine = -zeta1 / t; cosine = -zeta2 / (1 + t);
            tmp = sqrt(sine * sine + cosine * cosine);
            *s = sine / tmp; *c = cosine / tmp; *sestpr = sqrt(t + 1) * absest;
        }
    } else {
        if (sest == 0.0) {
            *sestpr = 0;
            if (fmax(absgam, absalp) == 0.0) { sine = 1; cosine = 0; } else { sine = -gamma; cosine = alpha; }
            s1 = fmax(fabs(sine), fabs(cosine));
            *s = sine / s1; *c = cosine / s1; tmp = sqrt(*s * *s + *c * *c); *s /= tmp; *c /= tmp;
        } else if (absgam <= eps * absest) {
            *s = 0; *c = 1; *sestpr = absgam;
        } else if (absalp <= eps * absest) {
            s1 = absgam; s2 = absest;
            if (s1 <= s2) { *s = 0; *c = 1; *sestpr = s1; } else { *s = 1; *c = 0; *sestpr = s2; }
        } else if (absest <= eps * absalp || absest <= eps * absgam) {
            s1 = absgam; s2 = absalp;
            if (s1 <= s2) { tmp = s1 / s2; *c = sqrt(1 + tmp * tmp); *sestpr = absest * (tmp / *c); *s = -(gamma / s2) / *c; *c = copysign(1.0, alpha) / *c; }
            else { tmp = s2 / s1; *s = sqrt(1 + tmp * tmp); *sestpr = absest / *s; *c = (alpha / s1) / *s; *s = -copysign(1.0, gamma) / *s; }
        } else {
            zeta1 = alpha / absest; zeta2 = gamma / absest;
            double norma = fmax(1 + zeta1 * zeta1 + fabs(zeta1 * zeta2), fabs(zeta1 * zeta2) + zeta2 * zeta2);
            double test = 1 + 2 * (zeta1 - zeta2) * (zeta1 + zeta2);
            if (test >= 0) {
                b = (zeta1 * zeta1 + zeta2 * zeta2 + 1) * 0.5; cc = zeta2 * zeta2;
                t = cc / (b + sqrt(fabs(b * b - cc)));
                sine = zeta1 / (1 - t); cosine = -zeta2 / t;
                *sestpr = sqrt(t + 4 * eps * eps * norma) * absest;
            } else {
                b = (zeta2 * zeta2 + zeta1 * zeta1 - 1) * 0.5; cc = zeta1 * zeta1;
                t = b >= 0 ? -cc / (b + sqrt(b * b + cc)) : b - sqrt(b * b + cc);
                sine = -zeta1 / t; cosine = -zeta2 / (1 + t);
                *sestpr = sqrt(1 + t + 4 * eps * eps * norma) * absest;
            }
            tmp = sqrt(sine * sine + cosine * cosine);
            *s = sine / tmp; *c = cosine / tmp;
        }
    }
}

// ws layout (doubles): vn1[n] vn2[n] tau[mn] wmin[mn] wmax[mn] tz[n] perm[n] ; jp (ints) separate
__global__ void __launch_bounds__(QR_NT)
k_qrcp_solve(double *__restrict__ A, int M, int n, double *__restrict__ b, int lenb, double *__restrict__ x,
             double *__restrict__ ws, int *__restrict__ jp, double *__restrict__ Cz /* n*n scratch */,
             double rcond, int *__restrict__ rank_out, int phase /* 1: factor, 2: apply Q' to b; solve always */) {
    __shared__ double sh[QR_NT / 64];
    __shared__ double s_val;
    __shared__ double s_val2;
    __shared__ int s_idx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = QR_NT / 64;
    const int mn = M < n ? M : n;
    double *vn1 = ws, *vn2 = ws + n, *tau = ws + 2 * n, *wmin = tau + mn, *wmax = wmin + mn;
    double *tz = wmax + mn, *perm = tz + n;
    const double tol3z = sqrt(DBL_EPSILON / 2);
    if ((phase & 4) && *rank_out == n) return;   // k_qr_rank + k_qr_backsolve already produced x
    if (phase & 1) {
    // column norms
    for (int j = wv; j < n; j += NW) {
        const double *c = A + (size_t)j * M;
        double acc = 0.0;
        for (int k = lane; k < M; k += 64) acc += c[k] * c[k];
        acc = wave_sum(acc);
        if (lane == 0) { double v = sqrt(acc); vn1[j] = v; vn2[j] = v; jp[j] = j; }
    }
    __syncthreads();
    for (int i = 0; i < mn; ++i) {
        if (tid == 0) {  // idamax: first maximum
            int p = i;
            for (int j = i + 1; j < n; ++j)
                if (vn1[j] > vn1[p]) p = j;
            s_idx = p;
        }
        __syncthreads();
        const int p = s_idx;
        double *ci = A + (size_t)i * M;
        if (p != i) {
            double *cp = A + (size_t)p * M;
            for (int k = tid; k < M; k += QR_NT) { double t = cp[k]; cp[k] = ci[k]; ci[k] = t; }
            if (tid == 0) { int t = jp[p]; jp[p] = jp[i]; jp[i] = t; vn1[p] = vn1[i]; vn2[p] = vn2[i]; }
        }
        __syncthreads();
        // dlarfg on A(i:M, i)
        double acc = 0.0;
        for (int k = i + 1 + tid; k < M; k += QR_NT) acc += ci[k] * ci[k];
        double xn = sqrt(blk_sum_qr(acc, sh));
        if (tid == 0) {
            double alpha = ci[i];
            if (xn == 0.0) { s_val = 0.0; s_val2 = 0.0; }
            else {
                double beta = -copysign(hypot(alpha, xn), alpha);
                s_val = (beta - alpha) / beta;       // tau
                s_val2 = 1.0 / (alpha - beta);       // scale
                ci[i] = beta;
            }
            tau[i] = s_val;
        }
        __syncthreads();
        const double ti = s_val, sc = s_val2;
        if (ti != 0.0)
            for (int k = i + 1 + tid; k < M; k += QR_NT) ci[k] *= sc;
        __syncthreads();
        // apply H(i) to the trailing columns (one wave per column), then downdate the norms
        for (int j = i + 1 + wv; j < n; j += NW) {
            double *cj = A + (size_t)j * M;
            double cji = cj[i];                      // same address in every lane
            if (ti != 0.0) {
                double w = 0.0;
                for (int k = i + 1 + lane; k < M; k += 64) w += ci[k] * cj[k];
                w = wave_sum(w);
                w = __shfl(w, 0, 64) + cji;          // v_i = 1
                const double tw = ti * w;
                for (int k = i + 1 + lane; k < M; k += 64) cj[k] -= ci[k] * tw;
                cji -= tw;
                if (lane == 0) cj[i] = cji;
            }
            const double v1 = vn1[j];
            if (v1 != 0.0) {  // wave-uniform; each lane re-reads only elements it wrote itself
                double r = fabs(cji) / v1;
                double temp = fmax(1.0 - r * r, 0.0);
                double q = v1 / vn2[j];
                double temp2 = temp * q * q;
                if (temp2 <= tol3z) {
                    double nv = 0.0;
                    if (i < M - 1) {
                        double a2 = 0.0;
                        for (int k = i + 1 + lane; k < M; k += 64) a2 += cj[k] * cj[k];
                        a2 = wave_sum(a2);
                        nv = sqrt(__shfl(a2, 0, 64));
                    }
                    if (lane == 0) { vn1[j] = nv; vn2[j] = nv; }
                } else if (lane == 0) {
                    vn1[j] = v1 * sqrt(temp);
                }
            }
        }
        __syncthreads();
    }
    }  // phase & 1
    // ---- rank detection (dlaic1), LinearAlgebra.ldiv!(::QRPivoted, B, rcond) [stdlib] ----
    int rnk = 0;
    {
        double smax = fabs(A[0]), smin = smax;
        if (smax == 0.0) {
            for (int k = tid; k < n; k += QR_NT) x[k] = 0.0;
            if (tid == 0) *rank_out = 0;
            return;
        }
        if (tid == 0) { wmin[0] = 1.0; wmax[0] = 1.0; }
        __syncthreads();
        rnk = 1;
        while (rnk < mn) {
            const int i = rnk;
            const double *col = A + (size_t)i * M;
            double a1 = 0.0, a2 = 0.0;
            for (int k = tid; k < rnk; k += QR_NT) { a1 += wmin[k] * col[k]; a2 += wmax[k] * col[k]; }
            a1 = blk_sum_qr(a1, sh);
            a2 = blk_sum_qr(a2, sh);
            double sminpr, s1, c1, smaxpr, s2, c2;
            laic1_dev(2, a1, smin, col[i], &sminpr, &s1, &c1);   // every thread computes the same
            laic1_dev(1, a2, smax, col[i], &smaxpr, &s2, &c2);
            if (smaxpr * rcond > sminpr) break;
            for (int k = tid; k < rnk; k += QR_NT) { wmin[k] *= s1; wmax[k] *= s2; }
            if (tid == 0) { wmin[i] = c1; wmax[i] = c2; }
            smin = sminpr; smax = smaxpr;
            rnk += 1;
            __syncthreads();
        }
        __syncthreads();
    }
    // ---- Q'b (dorm2r 'L','T'): H(0), H(1), ... in order ----
    for (int i = 0; (phase & 2) && i < mn; ++i) {
        const double *ci = A + (size_t)i * M;
        double acc = 0.0;
        for (int k = i + 1 + tid; k < M; k += QR_NT) acc += ci[k] * b[k];
        double s = (blk_sum_qr(acc, sh) + b[i]) * tau[i];
        __syncthreads();
        for (int k = i + 1 + tid; k < M; k += QR_NT) b[k] -= ci[k] * s;
        if (tid == 0) b[i] -= s;
        __syncthreads();
    }
    if (rnk < n) {
        // RZ factorisation of R(0:rnk, :) (dlatrz) into the scratch copy Cz (rnk x n, ld = rnk)
        const int l = n - rnk;
        for (int e = tid; e < rnk * n; e += QR_NT) {
            int r = e % rnk, cidx = e / rnk;
            Cz[e] = (r <= cidx) ? A[(size_t)cidx * M + r] : 0.0;
        }
        __syncthreads();
        for (int i = rnk - 1; i >= 0; --i) {
            double acc = 0.0;
            for (int k = tid; k < l; k += QR_NT) { double v = Cz[(size_t)(n - l + k) * rnk + i]; acc += v * v; }
            double xn = sqrt(blk_sum_qr(acc, sh));
            if (tid == 0) {
                double alpha = Cz[(size_t)i * rnk + i];
                if (xn == 0.0) { s_val = 0.0; s_val2 = 0.0; }
                else {
                    double beta = -copysign(hypot(alpha, xn), alpha);
                    s_val = (beta - alpha) / beta;
                    s_val2 = 1.0 / (alpha - beta);
                    Cz[(size_t)i * rnk + i] = beta;
                }
                tz[i] = s_val;
            }
            __syncthreads();
            const double ti = s_val, sc = s_val2;
            if (ti != 0.0)
                for (int k = tid; k < l; k += QR_NT) Cz[(size_t)(n - l + k) * rnk + i] *= sc;
            __syncthreads();
            if (ti != 0.0)
                for (int r = tid; r < i; r += QR_NT) {  // dlarz 'R' on rows 0..i-1
                    double w = Cz[(size_t)i * rnk + r];
                    for (int k = 0; k < l; ++k) w += Cz[(size_t)(n - l + k) * rnk + r] * Cz[(size_t)(n - l + k) * rnk + i];
                    Cz[(size_t)i * rnk + r] -= ti * w;
                    for (int k = 0; k < l; ++k) Cz[(size_t)(n - l + k) * rnk + r] -= ti * w * Cz[(size_t)(n - l + k) * rnk + i];
                }
            __syncthreads();
        }
        for (int i = rnk - 1; i >= 0; --i) {  // T z = (Q'b)(0:rnk), column-oriented
            if (tid == 0) b[i] = b[i] / Cz[(size_t)i * rnk + i];
            __syncthreads();
            const double bi = b[i];
            for (int k = tid; k < i; k += QR_NT) b[k] -= bi * Cz[(size_t)i * rnk + k];
            __syncthreads();
        }
        for (int k = rnk + tid; k < n; k += QR_NT) b[k] = 0.0;
        __syncthreads();
        for (int i = 0; i < rnk; ++i) {  // Z'b (dormr3 'L','T')
            double acc = 0.0;
            for (int k = tid; k < l; k += QR_NT) acc += Cz[(size_t)(n - l + k) * rnk + i] * b[n - l + k];
            double w = (blk_sum_qr(acc, sh) + b[i]) * tz[i];
            __syncthreads();
            for (int k = tid; k < l; k += QR_NT) b[n - l + k] -= Cz[(size_t)(n - l + k) * rnk + i] * w;
            if (tid == 0) b[i] -= w;
            __syncthreads();
        }
    } else {
        for (int i = n - 1; i >= 0; --i) {
            const double *ci = A + (size_t)i * M;
            if (tid == 0) b[i] = b[i] / ci[i];
            __syncthreads();
            const double bi = b[i];
            for (int k = tid; k < i; k += QR_NT) b[k] -= bi * ci[k];
            __syncthreads();
        }
    }
    for (int k = tid; k < n; k += QR_NT) perm[jp[k]] = b[k];
    __syncthreads();
    for (int k = tid; k < n; k += QR_NT) x[k] = perm[k];
    if (tid == 0) *rank_out = rnk;
    (void)lenb;
}

// ---- fast solve phase for n <= 2048 (after either factorisation) -------------------------------
// (1) rank decision: the dlaic1 recurrence of xGELSY is a chain of n dependent steps; a 1024-thread
//     workgroup pays two barriers per reduction (~4.5 us per step).  ONE wavefront with the two
//     estimate vectors in LDS needs no barrier at all (~0.6 us per step).
constexpr int QRK_MAXN = 2048;
constexpr int QRK_RPL = QRK_MAXN / 64;   // column entries per lane
__global__ void __launch_bounds__(64)
k_qr_rank(const double *__restrict__ A, int ld, int mn, double rcond, int *__restrict__ rank_out) {
    __shared__ double wmin[QRK_MAXN];
    __shared__ double wmax[QRK_MAXN];
    const int lane = threadIdx.x;
    double smax = fabs(A[0]), smin = smax;
    if (smax == 0.0) {
        if (lane == 0) *rank_out = 0;
        return;
    }
    if (lane == 0) { wmin[0] = 1.0; wmax[0] = 1.0; }
    // the column of step rnk+1 is fetched (all loads of a lane at once) while step rnk computes: the chain
    // of n dependent steps then costs the dlaic1 arithmetic, not a memory round trip per step
    auto fetch = [&](double (&ck)[QRK_RPL], double &gamma, int col_idx) {
        const int cc = col_idx < mn ? col_idx : mn - 1;
        const double *col = A + (size_t)cc * ld;
#pragma unroll
        for (int q = 0; q < QRK_RPL; ++q) {
            const int k = lane + 64 * q;
            ck[q] = k < cc ? col[k] : 0.0;
        }
        gamma = col[cc];
    };
    double ca[QRK_RPL], cb[QRK_RPL], ga, gb;
    fetch(ca, ga, 1);
    int rnk = 1;
    bool stop = false;
    auto step = [&](double (&ck)[QRK_RPL], double gamma, double (&nx)[QRK_RPL], double &gnx) {
        fetch(nx, gnx, rnk + 1);
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int q = 0; q < QRK_RPL; ++q) {
            const int k = lane + 64 * q;
            if (k < rnk) {
                a1 += wmin[k] * ck[q];
                a2 += wmax[k] * ck[q];
            }
        }
        a1 = __shfl(wave_sum(a1), 0, 64);
        a2 = __shfl(wave_sum(a2), 0, 64);
        double sminpr, s1, c1, smaxpr, s2, c2;
        laic1_dev(2, a1, smin, gamma, &sminpr, &s1, &c1);
        laic1_dev(1, a2, smax, gamma, &smaxpr, &s2, &c2);
        if (smaxpr * rcond > sminpr) { stop = true; return; }
        for (int k = lane; k < rnk; k += 64) { wmin[k] *= s1; wmax[k] *= s2; }
        if (lane == 0) { wmin[rnk] = c1; wmax[rnk] = c2; }
        smin = sminpr; smax = smaxpr;
        rnk += 1;
        __builtin_amdgcn_wave_barrier();
    };
    while (rnk < mn && !stop) {
        step(ca, ga, cb, gb);
        if (rnk < mn && !stop) step(cb, gb, ca, ga);
    }
    if (lane == 0) *rank_out = rnk;
}

// (2) full rank: R z = Q'b by 64-column blocks (diagonal block solved by one wavefront in LDS, the rows
//     above updated by all threads), then x[jp[k]] = z[k].  Does nothing when rank < n (the general
//     kernel with the minimum-norm completion runs instead).
__global__ void __launch_bounds__(QR_NT)
k_qr_backsolve(const double *__restrict__ A, int ld, int n, const double *__restrict__ b, const int *__restrict__ jp,
               const int *__restrict__ rank, double *__restrict__ x) {
    __shared__ double z[QRK_MAXN];
    __shared__ double D[64][65];
    if (*rank != n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int k = tid; k < n; k += QR_NT) z[k] = b[k];
    __syncthreads();
    for (int c1 = n; c1 > 0; c1 -= 64) {
        const int c0 = max(0, c1 - 64), nb = c1 - c0;
        for (int e = tid; e < 64 * 64; e += QR_NT) {
            const int r = e % 64, cidx = e / 64;
            D[r][cidx] = (r < nb && cidx < nb && r <= cidx) ? A[(size_t)(c0 + cidx) * ld + c0 + r] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {   // back substitution inside the block, column oriented (dtrsv 'U','N'): the lane's row of
                          // the block and its unknown live in registers, shuffles broadcast each solved value
            double drow[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) drow[j] = D[lane][j];
            double v = lane < nb ? z[c0 + lane] : 0.0;
#pragma unroll
            for (int j = 63; j >= 0; --j) {
                const double zj = __shfl(v, j, 64) / __shfl(drow[j], j, 64);
                if (j < nb) {
                    if (lane == j) v = zj;
                    else if (lane < j) v -= zj * drow[j];
                }
            }
            if (lane < nb) z[c0 + lane] = v;
        }
        __syncthreads();
        for (int r = tid; r < c0; r += QR_NT) {
            double s = 0.0;
            const double *row = A + r;
#pragma unroll 8
            for (int j = 0; j < nb; ++j) s += row[(size_t)(c0 + j) * ld] * z[c0 + j];
            z[r] -= s;
        }
        __syncthreads();
    }
    for (int k = tid; k < n; k += QR_NT) x[jp[k]] = z[k];
}

// ---------------------------------------------------------------------------------------------
// multi-CU column-pivoted Householder QR for larger matrices: the dlaqp2 recurrence, one column per
// step, two launches per step -- (a) pivot choice, column swap and reflector in one workgroup,
// (b) reflector applied to all trailing columns AND the right-hand side (column index n) with the
// partial-norm downdate, one wavefront per column.  Same arithmetic as k_qrcp_solve's factor phase.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_qr_norms(const double *__restrict__ A, int M, int n, double *__restrict__ vn1, double *__restrict__ vn2,
           int *__restrict__ jp) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *c = A + (size_t)j * M;
    double acc = 0.0;
    for (int k = lane; k < M; k += 64) acc += c[k] * c[k];
    acc = wave_sum(acc);
    if (lane == 0) { double v = sqrt(acc); vn1[j] = v; vn2[j] = v; jp[j] = j; }
}

__global__ void __launch_bounds__(QR_NT)
k_qr_pivot(double *__restrict__ A, int M, int n, int i, double *__restrict__ vn1, double *__restrict__ vn2,
           int *__restrict__ jp, double *__restrict__ tau) {
    __shared__ double sh[QR_NT / 64];
    __shared__ double s_best[QR_NT / 64];
    __shared__ int s_bidx[QR_NT / 64];
    __shared__ int s_p;
    __shared__ double s_tau, s_scale;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // idamax over vn1[i:n): the FIRST maximum
    double best = -1.0;
    int bidx = n;
    for (int j = i + tid; j < n; j += QR_NT) {
        const double v = vn1[j];
        if (v > best) { best = v; bidx = j; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) { s_best[wv] = best; s_bidx[wv] = bidx; }
    __syncthreads();
    if (tid == 0) {
        double b = s_best[0];
        int p = s_bidx[0];
        for (int w = 1; w < QR_NT / 64; ++w)
            if (s_best[w] > b || (s_best[w] == b && s_bidx[w] < p)) { b = s_best[w]; p = s_bidx[w]; }
        s_p = p < n ? p : i;
    }
    __syncthreads();
    const int p = s_p;
    double *ci = A + (size_t)i * M;
    if (p != i) {
        double *cp = A + (size_t)p * M;
        for (int k = tid; k < M; k += QR_NT) { double t = cp[k]; cp[k] = ci[k]; ci[k] = t; }
        if (tid == 0) { int t = jp[p]; jp[p] = jp[i]; jp[i] = t; vn1[p] = vn1[i]; vn2[p] = vn2[i]; }
    }
    __syncthreads();
    double acc = 0.0;
    for (int k = i + 1 + tid; k < M; k += QR_NT) acc += ci[k] * ci[k];
    const double xn = sqrt(blk_sum_qr(acc, sh));
    if (tid == 0) {
        const double alpha = ci[i];
        if (xn == 0.0) { s_tau = 0.0; s_scale = 0.0; }
        else {
            const double beta = -copysign(hypot(alpha, xn), alpha);
            s_tau = (beta - alpha) / beta;
            s_scale = 1.0 / (alpha - beta);
            ci[i] = beta;
        }
        tau[i] = s_tau;
    }
    __syncthreads();
    if (s_tau != 0.0) {
        const double sc = s_scale;
        for (int k = i + 1 + tid; k < M; k += QR_NT) ci[k] *= sc;
    }
}

__global__ void __launch_bounds__(256)
k_qr_apply(double *__restrict__ A, int M, int n, int i, double *__restrict__ rhs, const double *__restrict__ tau,
           double *__restrict__ vn1, double *__restrict__ vn2) {
    const int lane = threadIdx.x & 63;
    const int j = i + 1 + blockIdx.x * 4 + (threadIdx.x >> 6);   // j == n: the right-hand side
    if (j > n) return;
    const double ti = tau[i];
    const double *ci = A + (size_t)i * M;
    double *cj = j < n ? A + (size_t)j * M : rhs;
    const double tol3z = sqrt(DBL_EPSILON / 2);
    double cji = cj[i];
    if (ti != 0.0) {
        double w = 0.0;
        for (int k = i + 1 + lane; k < M; k += 64) w += ci[k] * cj[k];
        w = wave_sum(w);
        w = __shfl(w, 0, 64) + cji;          // v_i = 1
        const double tw = ti * w;
        for (int k = i + 1 + lane; k < M; k += 64) cj[k] -= ci[k] * tw;
        cji -= tw;
        if (lane == 0) cj[i] = cji;
    }
    if (j >= n) return;
    const double v1 = vn1[j];
    if (v1 != 0.0) {
        const double r = fabs(cji) / v1;
        const double temp = fmax(1.0 - r * r, 0.0);
        const double q = v1 / vn2[j];
        const double temp2 = temp * q * q;
        if (temp2 <= tol3z) {
            double nv = 0.0;
            if (i < M - 1) {
                double a2 = 0.0;
                for (int k = i + 1 + lane; k < M; k += 64) a2 += cj[k] * cj[k];
                a2 = wave_sum(a2);
                nv = sqrt(__shfl(a2, 0, 64));
            }
            if (lane == 0) { vn1[j] = nv; vn2[j] = nv; }
        } else if (lane == 0) {
            vn1[j] = v1 * sqrt(temp);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// helpers: stacked matrix [J; diag(sqrt(damp))] and right-hand side (y, 0)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(LSQ_NT)
k_stack(const double *__restrict__ J, int m, int n, const double *__restrict__ damp, double *__restrict__ Q) {
    const int M = damp ? m + n : m;
    const long long tot = (long long)M * n;
    for (long long e = blockIdx.x * (long long)LSQ_NT + threadIdx.x; e < tot; e += (long long)gridDim.x * LSQ_NT) {
        int r = (int)(e % M), c = (int)(e / M);
        double v;
        if (r < m) v = J[(size_t)c * m + r];
        else v = (r - m == c) ? sqrt(damp[c]) : 0.0;  // dense_qr.jl:72-74
        Q[e] = v;
    }
}
__global__ void __launch_bounds__(LSQ_NT)
k_rhs(const double *__restrict__ y, int m, int len, double *__restrict__ u) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < len; i += gridDim.x * LSQ_NT) u[i] = i < m ? y[i] : 0.0;
}

int lsq_dense_solver_alloc(lsq_solver *s) {
    const int m = s->m, n = s->n;
    const size_t n1 = n > 0 ? n : 1;
    LSQ_HIP(hipMalloc(&s->d_info, 4 * sizeof(int)));
    if (s->kind == LSQ_CHOLESKY) {
        LSQ_HIP(hipMalloc(&s->d_chol, n1 * n1 * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_Ds, ((size_t)((n + 63) / 64) * 4096 + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_rhs, n1 * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_work, 4 * n1 * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_tau, n1 * sizeof(int) + 16));  // pivots
    } else {
        const size_t M = s->for_lm ? (size_t)m + n : (size_t)m;       // dense_qr.jl:25-28, 50-54
        const size_t lu = s->for_lm ? M : (size_t)std::max(m, n);
        // (+32768: the register-resident panel steps fetch whole row slabs unconditionally, up to 20480 rows past a column)
        LSQ_HIP(hipMalloc(&s->d_qr, (M * n1 + 8 + 32768) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_qu, (lu + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_work, (8 * n1 + 3 * M + 64) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->d_tau, n1 * sizeof(int) + 16));         // jpvt
        LSQ_HIP(hipMalloc(&s->d_T, n1 * n1 * sizeof(double)));        // RZ scratch
    }
    return LSQ_OK;
}

void lsq_dense_solver_free(lsq_solver *s) {
    hipFree(s->d_info); hipFree(s->d_chol); hipFree(s->d_Ds); hipFree(s->d_rhs); hipFree(s->d_work); hipFree(s->d_tau);
    hipFree(s->d_qr); hipFree(s->d_qu); hipFree(s->d_T);
    if (s->qr2 && s->qr2_free) s->qr2_free(s->qr2);
    if (s->tripipe && s->tripipe_free) s->tripipe_free(s->tripipe);
    hipFree(s->tri_X); hipFree(s->tri_T); hipFree(s->tri_fro);
    if (s->tri_hfro) hipHostFree(s->tri_hfro);
    s->qr2 = nullptr;
}

// Dogleg's solver is the PIVOTED factorisation cholesky!(Symmetric(J'J), Val(true)) (dense_cholesky.jl:33; tol = 0,
// check = true): dpstrf picks the largest remaining diagonal entry as pivot and gives up (RankDeficientException)
// when that pivot is not positive.  Every such pivot is a diagonal entry of a Schur complement, hence
// >= lambda_min(J'J).  So if lambda_min(J'J) exceeds the rounding error a factorisation can commit
// (16 n eps max_j (J'J)_jj, generous), no pivot can fail: the reference returns the unique solution of the
// normal equations -- which the unpivoted blocked factorisation J'J = U'U delivers as well.
// lambda_min(J'J) = 1 / ||inv(U)||_2^2 >= 1 / ||inv(U)||_F^2, from the explicit inverse of U (the machinery of
// the QR certificate).  Returns true when the solve was done here (x in d_x, *rc = status); false: the caller runs
// the pivoted single-workgroup kernel, which also produces the reference's exception on a deficient matrix.
static bool chol_certified(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_x, int *rc) {
    const int n = J->n;
    *rc = LSQ_OK;
    auto fail = [&](int code) { *rc = code; return true; };
    if (lsq_cholesky_blocked(s, J, nullptr, nullptr, s->d_work) != LSQ_OK) return fail(LSQ_EHIP);
    // the solves run before the decision is known (discarded if the certificate refuses): the one synchronisation
    // below then also shows whether a wait of the pipelined solves gave up
    if (lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x) != LSQ_OK) return fail(LSQ_EHIP);   // mul!(x, J', y)
    if (lsq_cholesky_blocked_solve(s, n, d_x) != LSQ_OK) return fail(LSQ_EHIP);
    double fro2 = 0.0;
    int perr = 0;
    lsq_tri_pipe_err_copy(s, &perr);
    if (lsq_tri_inv_fro2(s, s->d_chol, n, &fro2) != LSQ_OK) return fail(LSQ_EHIP);   // (synchronises the stream)
    int info = 0;
    double dmax = 0.0;
    if (hipMemcpy(&info, s->d_info, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&dmax, s->d_work, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(LSQ_EHIP);
    const bool ok = info == 0 && std::isfinite(fro2) && fro2 > 0.0 && std::isfinite(dmax) &&
                    1.0 / fro2 > 16.0 * n * DBL_EPSILON * dmax;
    if (!ok) return false;
    if (perr) {
        lsq_tri_pipe_disable(s);
        if (lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x) != LSQ_OK) return fail(LSQ_EHIP);
        if (lsq_cholesky_blocked_solve(s, n, d_x) != LSQ_OK) return fail(LSQ_EHIP);
    }
    s->last_chol_path = 3;
    return true;
}

// dense_cholesky.jl:29-35 (d_damp == nullptr: pivoted) and :43-59 (damped, unpivoted)
int lsq_cholesky_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n;
    if (J->kind != LSQ_MAT_DENSE) {
        lsq_set_error("Cholesky() has no method for sparse Jacobians (dense_cholesky.jl:19)");
        return LSQ_EARG;
    }
    if (n != s->n || m != s->m) { lsq_set_error("cholesky: size mismatch"); return LSQ_EDIM; }
    const char *mn_env = getenv("LSQ_CHOL_MIN_N");
    int rc_cert = LSQ_OK;
    // blocked path from n = 32, or earlier when the rows make the SYRK the whole cost (tall and thin: 10^6 x 20 takes
    // 1.8 ms blocked, 90 ms with the one-workgroup kernels; 300 x 8 0.15 vs 0.09)
    const bool blocked = mn_env ? n >= atoi(mn_env) : (n >= 32 || (n >= 2 && (long long)m * n >= 20000));
    if (blocked && d_damp && !getenv("LSQ_NO_MFMA")) {
        // MFMA SYRK + blocked Cholesky + pipelined solves (lsq_dense_mfma.hip); measured crossover against the
        // single-workgroup kernel: 200 x 16 0.15 vs 0.11 ms, 300 x 32 0.15 vs 0.18, 500 x 64 0.16 vs 0.32, 2000 x 127 0.24 vs 0.93
        LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));  // mul!(x, J', y)
        LSQ_TRY(lsq_cholesky_blocked(s, J, d_damp, d_x, nullptr));
        s->last_chol_path = 2;
        int info = 0, perr = 0;
        LSQ_HIP(hipMemcpyAsync(&info, s->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        lsq_tri_pipe_err_copy(s, &perr);
        LSQ_HIP(hipStreamSynchronize(c->stream));
        if (info != 0) {
            lsq_set_error("PosDefException: matrix is not positive definite; Cholesky failed at %d", info);
            return LSQ_ENOTPD;
        }
        if (perr) {                               // (the factor is intact: only the two solves are repeated)
            lsq_tri_pipe_disable(s);
            LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));
            LSQ_TRY(lsq_cholesky_blocked_solve(s, n, d_x));
        }
    } else if (blocked && !d_damp && !getenv("LSQ_NO_MFMA") && !getenv("LSQ_CHOL_ALWAYS_PIVOT") &&
               chol_certified(s, J, d_y, d_x, &rc_cert)) {
        // Dogleg (dense_cholesky.jl:29-35): the unpivoted blocked factorisation gave the solution and the
        // certificate proved that cholesky!(.., Val(true)) would not have stopped early (see chol_certified)
        if (rc_cert != LSQ_OK) return rc_cert;
    } else if (n > 0) {
        if (rc_cert != LSQ_OK) return rc_cert;
        s->last_chol_path = 1;
        const int nt = (n + SY_T - 1) / SY_T;
        hipLaunchKernelGGL(k_syrk_upper, dim3(nt * (nt + 1) / 2), dim3(256), 0, c->stream, J->d_dense, m, n,
                           s->d_chol, d_damp);
        LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));  // mul!(x, J', y)
        int *piv = (int *)s->d_tau;
        if (d_damp)
            hipLaunchKernelGGL((k_chol_solve<false>), dim3(1), dim3(CH_NT), 0, c->stream, s->d_chol, n, d_x,
                               s->d_info, piv, s->d_work, s->d_work + 2 * n);
        else
            hipLaunchKernelGGL((k_chol_solve<true>), dim3(1), dim3(CH_NT), 0, c->stream, s->d_chol, n, d_x,
                               s->d_info, piv, s->d_work, s->d_work + 2 * n);
        LSQ_HIP(hipGetLastError());
        int info = 0;
        LSQ_HIP(hipMemcpyAsync(&info, s->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        LSQ_HIP(hipStreamSynchronize(c->stream));
        if (info != 0) {
            if (d_damp) {
                lsq_set_error("PosDefException: matrix is not positive definite; Cholesky failed at %d", info);
                return LSQ_ENOTPD;
            }
            lsq_set_error("RankDeficientException(%d)", info - 1);
            return LSQ_ERANK;
        }
    }
    if (nmul) *nmul = 1;
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// Two-stage column-pivoted QR: the pivoted sweep of dgeqp3 streams the whole trailing matrix once
// per column (C3: 2.6e11 bytes).  Stage 1 is an UNPIVOTED blocked Householder QR of [A | b]:
//   * 64-column panels; the panel steps carry K pivot columns per launch in registers (k_qr1_step_multi: lazy
//     reflectors, 1-256 row slabs per column with an in-kernel exchange, TSQR levels for tall thin operands; the older
//     per-column kernels k_qr1_step / _reg / _lazy remain selectable for A/B runs);
//   * per panel the trailing matrix is updated once with the compact-WY form on the fp64 MFMA units:
//     W = V'[V | A2 | b] (k_qr1_vtb: split-K, deterministic reduce), T'W through an explicit T' built in LDS from the
//     Gram block (k_qr1_tw_mfma), A2 -= V (T'W) (k_qr1_update); in the last panel b rides through the steps instead.
// Then either the FULL-RANK CERTIFICATE (qr2_certify_full_rank: explicit inverse of the triangle, Frobenius bound on
// cond_2 -- xGELSY's rank decision is provably n, the unpivoted triangle gives the solution: k_tri_bsolve), or stage 2:
// the pivoted sweep (k_qr2_step, the dlaqp2 recurrence with lazy column exchanges) on the n x n triangle R with Q1'b
// riding along: A P = Q1 (R P) = Q1 Q2 R2, so R2, the pivots and Q2'Q1'b are what ldiv!(::QRPivoted, b) needs; the rank
// decision (dlaic1, rcond) and the minimum-norm completion then run unchanged (k_qrcp_solve, phase 0).
// ---------------------------------------------------------------------------------------------
typedef double v4d_qr __attribute__((ext_vector_type(4)));
constexpr int Q2_NB = 64;     // panel width
constexpr int Q2_KC = 32;     // k-rows staged per MFMA step
constexpr int Q2_KS = Q2_KC + 2;

// One column step inside a panel [c0, cend).  first: only the reflector of column c0.  Otherwise
// block b applies H_i to column j = i+1+b of the panel; the block of j == i+1 then forms H_{i+1}
// from the column it has just updated (dlarfg), so a step is ONE launch.
__global__ void __launch_bounds__(QR_NT)
k_qr1_step(double *__restrict__ A, int M, int cend, int i, int first, double *__restrict__ tau) {
    __shared__ double sh[QR_NT / 64];
    __shared__ double s_w;
    const int tid = threadIdx.x;
    const int j = first ? i : i + 1 + blockIdx.x;
    if (j >= cend) return;
    double *cj = A + (size_t)j * M;
    if (!first) {
        const double *ci = A + (size_t)i * M;
        const double ti = tau[i];
        if (ti != 0.0) {
            double w = 0.0;
            for (int k = i + 1 + tid; k < M; k += QR_NT) w += ci[k] * cj[k];
            w = blk_sum_qr(w, sh);
            if (tid == 0) s_w = ti * (w + cj[i]);   // v_i(i) = 1
            __syncthreads();
            const double tw = s_w;
            for (int k = i + 1 + tid; k < M; k += QR_NT) cj[k] -= ci[k] * tw;
            if (tid == 0) cj[i] -= tw;
            __syncthreads();
        }
        if (j != i + 1) return;
    }
    // reflector of column j on rows j..M-1 (dlarfg)
    double acc = 0.0;
    for (int k = j + 1 + tid; k < M; k += QR_NT) acc += cj[k] * cj[k];
    const double xn = sqrt(blk_sum_qr(acc, sh));
    __shared__ double s_tau, s_scale;
    if (tid == 0) {
        const double alpha = j < M ? cj[j] : 0.0;
        if (xn == 0.0 || j >= M) { s_tau = 0.0; s_scale = 0.0; }
        else {
            const double beta = -copysign(hypot(alpha, xn), alpha);
            s_tau = (beta - alpha) / beta;
            s_scale = 1.0 / (alpha - beta);
            cj[j] = beta;
        }
        tau[j] = s_tau;
    }
    __syncthreads();
    if (s_tau != 0.0) {
        const double sc = s_scale;
        for (int k = j + 1 + tid; k < M; k += QR_NT) cj[k] *= sc;
    }
}

// Same step with the column held in registers (<= RPT rows per thread: M - i - 1 <= RPT * QR_NT): one
// load round trip, two block reductions, one store -- the loop version above pays a memory round trip
// per pass and per 4 rows.
template <int RPT>
__global__ void __launch_bounds__(QR_NT)
k_qr1_step_reg(double *__restrict__ A, int M, int cend, int i, int first, double *__restrict__ tau) {
    __shared__ double sh[QR_NT / 64];
    __shared__ double s_w, s_tau, s_scale;
    const int tid = threadIdx.x;
    const int j = first ? i : i + 1 + blockIdx.x;
    if (j >= cend) return;
    double *cj = A + (size_t)j * M;
    const int base = (first ? i : i + 1);       // rows base + tid + q*QR_NT, q < RPT (rows below row i, or from row i when first)
    // element q of a thread is (uniform pointer + q*QR_NT)[t] with ONE unsigned per-thread offset t: the
    // loads take the scalar-base + 32-bit-offset form instead of RPT 64-bit address pairs in VGPRs
    const unsigned t = (unsigned)(base + tid);
    double a[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) a[q] = (int)t + q * QR_NT < M ? (cj + q * QR_NT)[t] : 0.0;
    if (!first) {
        const double *ci = A + (size_t)i * M;
        const double ti = tau[i];
        if (ti != 0.0) {
            // only the column itself stays in registers: v_i is read twice (the second time from L2)
            const double cji = cj[i];
            double w = 0.0;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const double vq = (int)t + q * QR_NT < M ? (ci + q * QR_NT)[t] : 0.0;
                w += vq * a[q];
            }
            w = blk_sum_qr(w, sh);
            if (tid == 0) s_w = ti * (w + cji);   // v_i(i) = 1
            __syncthreads();
            const double tw = s_w;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const double vq = (int)t + q * QR_NT < M ? (ci + q * QR_NT)[t] : 0.0;
                a[q] -= vq * tw;
            }
            if (tid == 0) cj[i] = cji - tw;
        }
        if (j != i + 1) {
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                if ((int)t + q * QR_NT < M) (cj + q * QR_NT)[t] = a[q];
            return;
        }
    }
    // reflector of column j (dlarfg) on rows j..M-1: row j itself is element 0 of thread 0
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q)
        if ((int)t + q * QR_NT > j) acc += a[q] * a[q];
    const double xn = sqrt(blk_sum_qr(acc, sh));
    if (tid == 0) {
        const double alpha = a[0];              // row base == j
        if (xn == 0.0) { s_tau = 0.0; s_scale = 1.0; }
        else {
            const double beta = -copysign(hypot(alpha, xn), alpha);
            s_tau = (beta - alpha) / beta;
            s_scale = 1.0 / (alpha - beta);
            a[0] = beta;
        }
        tau[j] = s_tau;
    }
    __syncthreads();
    const double sc = s_tau != 0.0 ? s_scale : 1.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int k = (int)t + q * QR_NT;
        if (k < M) (cj + q * QR_NT)[t] = (k > j) ? a[q] * sc : a[q];
    }
}

// Lazy-reflector variant of the step: column i is left UNSCALED by the launch that finished it; every
// workgroup of launch i forms H_i itself from it (sum v^2 and v'a share ONE block reduction), so a step has
// a single reduction round on its critical path instead of two (apply, then form the next reflector).
// beta_i, tau_i and the scale 1/(alpha - beta) go to side arrays: R(i,i) is not written while others read it,
// and the V materialisation applies the scale.  bookkeeping-only launch (nblocks == 0 columns): i = cend-1.
template <int RPT>
__global__ void __launch_bounds__(QR_NT)
k_qr1_step_lazy(double *__restrict__ A, int M, int cend, int i, double *__restrict__ tau, double *__restrict__ beta_out,
                double *__restrict__ scale_out) {
    __shared__ double sh2[QR_NT / 64][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = i + 1 + blockIdx.x;
    const bool has_col = j < cend;             // (the last column of a panel has nobody to update: bookkeeping only)
    const double *ci = A + (size_t)i * M;
    double *cj = A + (size_t)(has_col ? j : i) * M;
    const unsigned t = (unsigned)(i + 1 + tid);
    double a[RPT], v[RPT];
    double svv = 0.0, sva = 0.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const bool in = (int)t + q * QR_NT < M;
        v[q] = in ? (ci + q * QR_NT)[t] : 0.0;
        a[q] = (in && has_col) ? (cj + q * QR_NT)[t] : 0.0;
        svv += v[q] * v[q];
        sva += v[q] * a[q];
    }
    const double alpha = ci[i];
    const double aji = has_col ? cj[i] : 0.0;
    svv = wave_sum(svv);
    sva = wave_sum(sva);
    if (lane == 0) { sh2[wv][0] = svv; sh2[wv][1] = sva; }
    __syncthreads();
    double tvv = 0.0, tva = 0.0;
#pragma unroll
    for (int w = 0; w < QR_NT / 64; ++w) { tvv += sh2[w][0]; tva += sh2[w][1]; }
    const double xn = sqrt(tvv);
    double ti = 0.0, beta = alpha, sc = 0.0;
    if (xn != 0.0) {
        beta = -copysign(hypot(alpha, xn), alpha);
        ti = (beta - alpha) / beta;
        sc = 1.0 / (alpha - beta);
    }
    if (blockIdx.x == 0 && tid == 0) { tau[i] = ti; beta_out[i] = beta; scale_out[i] = sc; }
    if (!has_col || ti == 0.0) return;
    const double tw = ti * (sc * tva + aji);      // v_i(i) = 1, v_i(k) = sc * column_i(k)
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        a[q] -= (v[q] * sc) * tw;
        if ((int)t + q * QR_NT < M) (cj + q * QR_NT)[t] = a[q];
    }
    if (tid == 0) cj[i] = aji - tw;
}

// DPP all-reductions (no LDS round trips; helpers of lsq_common.h): after row_allsum every lane of a 16-lane row holds
// the row's sum (rotations pair the same operands in every lane, so the lanes agree bit for bit); wave_allsum = wave_sum
// adds the four row sums in a fixed order.
__device__ __forceinline__ double row_allsum(double x) {
    x += lsq_dpp_mov_f64<0x128>(x);   // row_ror:8
    x += lsq_dpp_mov_f64<0x124>(x);   // row_ror:4
    x += lsq_dpp_mov_f64<0x122>(x);   // row_ror:2
    x += lsq_dpp_mov_f64<0x121>(x);   // row_ror:1
    return x;
}
__device__ __forceinline__ double readlane_f64(double x, int l) { return lsq_readlane_f64(x, l); }
__device__ __forceinline__ double wave_allsum(double x) { return wave_sum(x); }

// K lazy reflectors per launch.  A launch's fixed cost (dispatch + the fetch of the columns) dominates a step,
// so one launch carries the K pivot columns i .. i+K-1 through K reduction rounds: every workgroup fetches
// the K pivot columns and its own target column i+K+blockIdx.x, and in round r forms H_{i+r} from pivot
// column r (as k_qr1_step_lazy does) and applies it to the later pivot columns and to the target, all in
// registers.  The pivot columns are updated redundantly by every workgroup (identical arithmetic); workgroup
// 0 stores them and the side arrays.  Thread rows start at row i so that the pivot element of every round is
// an ordinary masked element: "alpha" and the row-(i+r) entries of the other columns come out of the same
// block reduction as the dot products (sums with a single non-zero term are exact).
//
// S > 1: the rows of a column are cut into S slabs of RPT*NT rows, one workgroup each -- S times more CUs
// stream the panel, a workgroup holds 1/S of a column per array (so K can be larger), and the S workgroups of
// a target column ("group") add their partial sums through memory once per round: each publishes its NS
// partials with agent-scope stores, then picks up all S sets (its own included, so every member adds the
// same values in the same order) and goes on.  The members of a group sit at block
// indices 8 apart -- the same XCD under the round-robin dispatch -- and a launch never has more workgroups
// than the device holds at once (host side), so the members of a group are always co-resident; the wait is
// bounded anyway and reports through *err instead of hanging.
constexpr int QR1_SPIN_LIMIT = 1 << 22;
template <int N, class F>
__device__ __forceinline__ void qr_static_for(F &&f) {
    if constexpr (N > 0) {
        qr_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
//
// TSQR = 1 (tall and thin operands, n <= K; kept for reference, not instantiated): every workgroup is on its own -- its slab of RPT*NT rows of ALL n
// columns and of b sits in registers, the K rounds factor the slab locally (pivot rows = the slab's first rows, no
// exchange), and the only thing written is the slab's n x n triangle and the first n entries of its Q'b, stacked
// for the next level (tsq_S: (slabs*n) x n, tsq_r): ONE pass over the matrix.
// TSQR = 2 (the one in use): the same with one WAVEFRONT per slab (64*RPT rows): the reductions are wave reductions, the
// row-c elements come from their owner lane by v_readlane -- no LDS, no barrier anywhere in the rounds.
template <int NT, int RPT, int K, int S, int TSQR = 0>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_qr1_step_multi(double *__restrict__ A, int M, int cend, int i, int kk /* live pivots, <= K */, double *__restrict__ tau,
                 double *__restrict__ beta_out, double *__restrict__ scale_out, int G /* groups = target columns (>= 1) */,
                 unsigned long long *__restrict__ xslot, unsigned long long epoch,
                 int *__restrict__ err, double *__restrict__ Pn /* side panel: column (col - c0) * M */, int c0,
                 double *__restrict__ rhs_col /* last panel: the right-hand side rides along as target column "cend" */,
                 double *__restrict__ tsq_S = nullptr, int tsq_ld = 0, double *__restrict__ tsq_r = nullptr) {
    constexpr int NW = NT / 64;
    constexpr int NS = 2 * (K + 1);
    __shared__ double sh[NW][NS];
    __shared__ double sx[S][NS];
    __shared__ double sat[NS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int QS = TSQR == 2 ? 64 : NT;      // rows between two elements of a thread
    int g = (int)blockIdx.x, sidx = 0;
    if (TSQR) {                      // one workgroup (or wavefront) per slab, a single "group" whose target is b
        g = 0;
        sidx = TSQR == 2 ? (int)blockIdx.x * NW + wv : (int)blockIdx.x;
    } else if (S > 1 && S < 64) {           // members 8 apart: one XCD; 8 * S workgroups must be resident together
        const int kq = (int)blockIdx.x >> 3;
        sidx = kq % S;
        g = (kq / S) * 8 + ((int)blockIdx.x & 7);
        if (g >= G) return;
    } else if (S >= 64) {            // members consecutive (all XCDs): a group spans S <= 256 indices (the device holds 512)
        g = (int)blockIdx.x / S;
        sidx = (int)blockIdx.x % S;
    }
    const int j = i + kk + g;
    const bool is_rhs = rhs_col != nullptr && j == cend;
    const bool has_col = j < cend || is_rhs;
    // one buffer descriptor per column (scalar base, byte count M*8): every fetch and store is descriptor +
    // ONE shared 32-bit VGPR offset + a scalar offset; rows beyond M read as zero and their stores are dropped by
    // the bounds check, and a dead column (ragged last launch, no target) gets an empty descriptor
    typedef unsigned v2u_qr __attribute__((ext_vector_type(2)));
    const int t = i + sidx * (RPT * QS) + (TSQR == 2 ? lane : tid);    // row of element 0
    const unsigned tb = (unsigned)t * 8u;
    const unsigned colbytes = (unsigned)M * 8u;
    double pv[K][RPT], a[RPT];
    __amdgpu_buffer_rsrc_t rp[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        rp[r] = __builtin_amdgcn_make_buffer_rsrc(A + (size_t)(i + (r < kk ? r : 0)) * M, 0, r < kk ? colbytes : 0u, 0x00020000);
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const v2u_qr w = __builtin_amdgcn_raw_buffer_load_b64(rp[r], tb, q * QS * 8, 0);
            pv[r][q] = __builtin_bit_cast(double, w);
        }
    }
    const __amdgpu_buffer_rsrc_t rj =
        __builtin_amdgcn_make_buffer_rsrc(is_rhs ? rhs_col : A + (size_t)(has_col ? j : i) * M, 0, has_col ? colbytes : 0u, 0x00020000);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const v2u_qr w = __builtin_amdgcn_raw_buffer_load_b64(rj, tb, q * QS * 8, 0);
        a[q] = __builtin_bit_cast(double, w);
    }
    double mybeta = 0.0;   // (TSQR)
    // (rounds as instantiations, not as a loop: the unroller gives up on a body of this size for K >= 6 and the
    // register arrays would land in scratch)
    auto round = [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if (r >= kk) return;                        // (uniform over the whole grid: a dead round of a ragged launch)
        const bool live = true;
        const int c = (TSQR ? i + sidx * (RPT * QS) : i) + r;   // pivot row of this round (TSQR: of this slab)
        // sm: [0] v'v  [1] alpha  then per later column x (pivots r+1.., target): v'x and x(c).  The even entries are
        // sums over all rows; the odd ones are single elements of row c, all owned by ONE thread (slab 0, thread r:
        // only element 0 of a thread can sit at or above the pivot row) which hands them out directly
        double sm[NS];
        constexpr int NSr = 2 * (K - r) + 2;
        {
            const double v = t > c ? pv[r][0] : 0.0;
            sm[0] = v * v;
#pragma unroll
            for (int x = 0; x < K; ++x)       // (constant trip counts: the unroller must not depend on r)
                if (x > r) sm[2 * (x - r)] = v * pv[x][0];
            sm[2 * (K - r)] = v * a[0];
        }
#pragma unroll
        for (int q = 1; q < RPT; ++q) {
            const double v = pv[r][q];
            sm[0] = __builtin_fma(v, v, sm[0]);
#pragma unroll
            for (int x = 0; x < K; ++x)
                if (x > r) sm[2 * (x - r)] = __builtin_fma(v, pv[x][q], sm[2 * (x - r)]);
            sm[2 * (K - r)] = __builtin_fma(v, a[q], sm[2 * (K - r)]);
        }
#pragma unroll
        for (int e = 0; e < NS; e += 2)
            if (e < NSr) sm[e] = wave_allsum(sm[e]);
        if constexpr (TSQR == 2) {
            // wave slab: the sums are complete; the row-c elements sit in lane r
            sm[1] = readlane_f64(pv[r][0], r);
#pragma unroll
            for (int x = 0; x < K; ++x)
                if (x > r) sm[2 * (x - r) + 1] = readlane_f64(pv[x][0], r);
            sm[2 * (K - r) + 1] = readlane_f64(a[0], r);
        } else {
        __syncthreads();                             // (the previous round's readers of sh, sx and sat are done)
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < NS; e += 2)
                if (e < NSr) sh[wv][e] = sm[e];
        }
        if (t == c) {                                // the owner of row c (slab 0, thread r)
            sat[1] = pv[r][0];
#pragma unroll
            for (int x = 0; x < K; ++x)
                if (x > r) sat[2 * (x - r) + 1] = pv[x][0];
            sat[2 * (K - r) + 1] = a[0];
        }
        __syncthreads();
        static_assert(NW <= 16, "one 16-lane row sums the wave partials");
#pragma unroll
        for (int e = 0; e < NS; e += 2)
            if (e < NSr) sm[e] = row_allsum((lane & 15) < NW ? sh[lane & 15][e] : 0.0);
        }
        if constexpr (TSQR == 2) {
        } else if (S == 1 || TSQR) {
#pragma unroll
            for (int e = 1; e < NS; e += 2)
                if (e < NSr) sm[e] = sat[e];
        } else {
            // flag-in-data exchange (the low-latency protocol of the collectives libraries): every 64-bit word
            // carries 32 bits of payload and the 32-bit epoch, so a reader that sees the epoch has the payload --
            // one store and one load on the critical path, no fences, no separate flag.  Sums: every slab publishes
            // its partial; row-c elements: slab 0 alone publishes them.
            const unsigned ep = (unsigned)epoch;
            if (tid < NSr && ((tid & 1) == 0 || sidx == 0)) {
                double val = 0.0;
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (e < NSr && tid == e) val = (e & 1) ? sat[e] : sm[e];
                unsigned long long *mine = xslot + ((((size_t)g * S + sidx) * K + r) * NS + tid) * 2;
                const unsigned long long hi = (unsigned long long)ep << 32;
                __hip_atomic_store(mine, hi | (unsigned)__double2loint(val), RLX_AGENT);
                __hip_atomic_store(mine + 1, hi | (unsigned)__double2hiint(val), RLX_AGENT);
            }
            for (int idx = tid; idx < S * NS; idx += NT) {
                const int sp = idx / NS, e = idx % NS;
                if (e < NSr && ((e & 1) == 0 || sp == 0)) {
                    const unsigned long long *f = xslot + ((((size_t)g * S + sp) * K + r) * NS + e) * 2;
                    unsigned long long w0 = __hip_atomic_load(f, RLX_AGENT), w1 = __hip_atomic_load(f + 1, RLX_AGENT);
                    int spins = 0;
                    while ((unsigned)(w0 >> 32) != ep || (unsigned)(w1 >> 32) != ep) {
                        if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); break; }
                        __builtin_amdgcn_s_sleep(1);
                        w0 = __hip_atomic_load(f, RLX_AGENT);
                        w1 = __hip_atomic_load(f + 1, RLX_AGENT);
                    }
                    sx[sp][e] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                if (e < NSr) {
                    if (e & 1) sm[e] = sx[0][e];
                    else {
                        double tot = 0.0;
#pragma unroll 8
                        for (int sp = 0; sp < S; ++sp) tot += sx[sp][e];   // (fixed order: every member gets the same bits)
                        sm[e] = tot;
                    }
                }
            }
        }
        const double alpha = sm[1];
        double ti = 0.0, beta = alpha, sc = 0.0;
        if (sm[0] != 0.0) {
            // (sum v^2 is already formed unscaled, so dlapy2's overflow guard has nothing left to protect)
            beta = -copysign(sqrt(__builtin_fma(alpha, alpha, sm[0])), alpha);
            ti = (beta - alpha) / beta;
            sc = 1.0 / (alpha - beta);
        }
        if (!live) ti = 0.0;
        if (TSQR) {
            if ((TSQR == 2 ? lane : tid) == r) mybeta = beta;   // R(r, r) of this slab, kept by the owner of the slab's row r
        } else if (live && g == 0 && sidx == 0 && tid == 0) { tau[c] = ti; beta_out[c] = beta; scale_out[c] = sc; }
        if (ti != 0.0) {
            double tw[K + 1];
#pragma unroll
            for (int x = 0; x <= K; ++x)
                if (x > r) tw[x] = ti * (sc * sm[2 * (x - r)] + sm[2 * (x - r) + 1]);
            {
                const double vs = t > c ? pv[r][0] * sc : (t == c ? 1.0 : 0.0);
#pragma unroll
                for (int x = 0; x < K; ++x)
                    if (x > r) pv[x][0] -= vs * tw[x];
                a[0] -= vs * tw[K];
            }
#pragma unroll
            for (int q = 1; q < RPT; ++q) {
                const double vs = pv[r][q] * sc;
#pragma unroll
                for (int x = 0; x < K; ++x)
                    if (x > r) pv[x][q] = __builtin_fma(-vs, tw[x], pv[x][q]);
                a[q] = __builtin_fma(-vs, tw[K], a[q]);
            }
        }
    };
    qr_static_for<K>(round);
    if (TSQR) {
        // rows 0 .. kk-1 of the slab: R(r, x) = element 0 of thread r in column x (x > r), beta on the diagonal, zeros
        // before it; and entry r of the slab's Q'b
        const int own = TSQR == 2 ? lane : tid;
        if (own < kk && sidx < G) {                  // (G: number of slabs; a workgroup's last wavefronts may have none)
            const size_t row = (size_t)sidx * kk + own;
#pragma unroll
            for (int x = 0; x < K; ++x)
                if (x < kk) tsq_S[(size_t)x * tsq_ld + row] = x < own ? 0.0 : (x == own ? mybeta : pv[x][0]);
            tsq_r[row] = a[0];
        }
        return;
    }
    if (g == 0) {
        // pivot columns 1 .. kk-1 are final (unscaled) now.  They go to the SIDE panel, not in place: other groups
        // may not have fetched them yet (a launch can be larger than what the device holds at once), and
        // nobody but k_qr1_vbuf needs them again -- it moves them back while it builds V
#pragma unroll
        for (int r = 1; r < K; ++r) {
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(Pn + (size_t)(i + (r < kk ? r : 0) - c0) * M, 0, r < kk ? colbytes : 0u, 0x00020000);
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_qr, pv[r][q]), rs, tb, q * NT * 8, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_qr, a[q]), rj, tb, q * NT * 8, 0);
}

// V (unit lower trapezoid of the panel, zeros above, zero columns beyond nb) -> Vb[row - c0][col], ld = ldv
__global__ void __launch_bounds__(256)
k_qr1_vbuf(double *__restrict__ A, int M, int c0, int nb, double *__restrict__ Vb, int ldv,
           const double *__restrict__ beta, const double *__restrict__ scale /* lazy reflectors: column c0+c is stored
           unscaled and its R(c,c) = beta is put on the diagonal here, once nobody reads the old pivot element */,
           const double *__restrict__ Pn, int K /* k_qr1_step_multi: the later pivot columns of a launch (c % K != 0)
           were left in the side panel from the launch's first pivot row on; they return to A here */) {
    const int rows = M - c0;
    const long long tot = (long long)rows * Q2_NB;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int r = (int)(e % rows), cidx = (int)(e / rows);
        double v = 0.0;
        if (cidx < nb) {
            double *pa = A + (size_t)(c0 + cidx) * M + c0 + r;
            if (Pn && cidx % K != 0 && r >= cidx / K * K) *pa = Pn[(size_t)cidx * M + c0 + r];
            if (r > cidx) v = *pa * (scale ? scale[c0 + cidx] : 1.0);
            else if (r == cidx) {
                v = 1.0;
                if (beta) *pa = beta[c0 + cidx];
            }
        }
        Vb[(size_t)cidx * ldv + r] = v;
    }
}

// last panel, right-hand side already transformed by the steps: nobody needs V any more, only the panel's part of R --
// the rows c0 .. c0+nb-1 of the side-panel columns go back to A and beta goes on the diagonal
__global__ void __launch_bounds__(256)
k_qr1_fin(double *__restrict__ A, int M, int c0, int nb, const double *__restrict__ beta, const double *__restrict__ Pn, int K) {
    for (int e = threadIdx.x; e < Q2_NB * Q2_NB; e += 256) {
        const int r = e % Q2_NB, cidx = e / Q2_NB;
        if (cidx >= nb || r > cidx || c0 + r >= M) continue;
        double *pa = A + (size_t)(c0 + cidx) * M + c0 + r;
        if (r == cidx) *pa = beta[c0 + cidx];
        else if (Pn && cidx % K != 0 && r >= cidx / K * K) *pa = Pn[(size_t)cidx * M + c0 + r];
    }
}

// column cb of the virtual matrix B = [V | A(:, cend:n) | b] restricted to rows c0..M-1
__device__ __forceinline__ const double *q2_bcol(const double *Vb, int ldv, const double *A, int M, int c0, int cend, int n,
                                                 const double *rhs, int cb) {
    if (cb < Q2_NB) return Vb + (size_t)cb * ldv;
    const int a = cend + (cb - Q2_NB);
    return (a < n ? A + (size_t)a * M : rhs) + c0;
}

// Wp[slice][tile][64 x 64] = V(rows of the slice)' * B(rows of the slice, 64 columns of tile)   (fp64 MFMA)
__global__ void __launch_bounds__(256)
k_qr1_vtb(const double *__restrict__ Vb, int ldv, const double *__restrict__ A, int M, int c0, int cend, int n,
          const double *__restrict__ rhs, int ncolsB, int kslices, double *__restrict__ Wp) {
    __shared__ double sA[Q2_NB * Q2_KS];
    __shared__ double sB[Q2_NB * Q2_KS];
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
    const int tile = blockIdx.x % ntile, slice = blockIdx.x / ntile;
    const int rows = M - c0;
    const int kper = ((rows + kslices - 1) / kslices + Q2_KC - 1) / Q2_KC * Q2_KC;
    const int kb = slice * kper, ke = min(rows, kb + kper);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    v4d_qr acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d_qr){0.0, 0.0, 0.0, 0.0};
    const int lc = tid >> 2, lk = (tid & 3) * 8;
    const int cb = tile * Q2_NB + lc;
    const double *pbcol = cb < ncolsB ? q2_bcol(Vb, ldv, A, M, c0, cend, n, rhs, cb) : nullptr;
    const double *pacol = Vb + (size_t)lc * ldv;
    // the next 32-row slab is fetched into registers while the MFMAs of the current one run
    double ra[8], rb[8];
    const double *pb = pbcol ? pbcol : Vb;     // (a dummy column for tiles past the last one: fetched, then zeroed)
    const bool bok = pbcol != nullptr;
    auto fetch = [&](int k0) {
        if (k0 + Q2_KC <= ke) {                // full slab: unconditional fetches, nothing to branch on
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ra[q] = pacol[k0 + lk + q];
                const double y = pb[k0 + lk + q];
                rb[q] = bok ? y : 0.0;
            }
        } else {                               // the ragged last slab: clamped addresses, zeroed by selection
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + lk + q, ks = min(k, ke - 1);
                const double x = pacol[ks], y = pb[ks];
                ra[q] = k < ke ? x : 0.0;
                rb[q] = (k < ke && bok) ? y : 0.0;
            }
        }
    };
    if (kb < ke) fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += Q2_KC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sA[lc * Q2_KS + lk + q] = ra[q];
            sB[lc * Q2_KS + lk + q] = rb[q];
        }
        __syncthreads();
        if (k0 + Q2_KC < ke) fetch(k0 + Q2_KC);
#pragma unroll
        for (int kk = 0; kk < Q2_KC; kk += 4) {
            const int ko = kk + (lane >> 4);
            const double a0 = sA[(wr + (lane & 15)) * Q2_KS + ko];
            const double a1 = sA[(wr + 16 + (lane & 15)) * Q2_KS + ko];
            const double b0 = sB[(wc + (lane & 15)) * Q2_KS + ko];
            const double b1 = sB[(wc + 16 + (lane & 15)) * Q2_KS + ko];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    double *out = Wp + ((size_t)slice * ntile + tile) * (Q2_NB * Q2_NB);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + (lane >> 4) + 4 * r;   // index into V's columns
                const int col = wc + b * 16 + (lane & 15);           // index into the tile's B columns
                out[(size_t)col * Q2_NB + row] = acc[a][b][r];
            }
}

// W[cb][0:64] = sum over slices (fixed order)
__global__ void __launch_bounds__(256)
k_qr1_wreduce(const double *__restrict__ Wp, int ncolsB, int kslices, double *__restrict__ W) {
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
    const long long tot = (long long)ntile * Q2_NB * Q2_NB;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        double s = 0.0;
        for (int sl = 0; sl < kslices; ++sl) s += Wp[(size_t)sl * tot + e];
        W[e] = s;   // layout [tile][col][row] == [cb][row]
    }
}

// W2 = T' W for the columns of A2 and b without forming T: for H_0 ... H_{nb-1} = I - V T V' the inverse
// of T is the upper triangle of the Gram block G = V'V with 1/tau on the diagonal (tau_j = 2 / v_j'v_j),
// so T' W = W2 solves the unit-structured lower-triangular system
//      W2[j] = tau_j * (W[j] - sum_{k<j} G[j][k] W2[k])
// (tau_j = 0, i.e. H_j = I, gives W2[j] = 0 as dlarft's zero column does).  One thread per column: 2016
// FMAs with G broadcast from LDS -- instead of a 64-step, two-barrier recurrence in every workgroup.
__global__ void __launch_bounds__(256)
k_qr1_tw(const double *__restrict__ W, int ncolsB, const double *__restrict__ tau, int c0, int nb, double *__restrict__ W2) {
    __shared__ double G[Q2_NB][Q2_NB + 1];   // G[j][k] = v_j'v_k (k < j used)
    __shared__ double st[Q2_NB];
    const int tid = threadIdx.x;
    for (int e = tid; e < Q2_NB * Q2_NB; e += 256) {
        const int j = e / Q2_NB, k = e % Q2_NB;
        G[j][k] = W[(size_t)j * Q2_NB + k];    // W[cb = j][row = k] = v_k'v_j (symmetric)
    }
    if (tid < Q2_NB) st[tid] = tid < nb ? tau[c0 + tid] : 0.0;
    __syncthreads();
    const int ncols = ncolsB - Q2_NB;
    for (int cidx = blockIdx.x * 256 + tid; cidx < ncols; cidx += gridDim.x * 256) {
        const double *wc = W + (size_t)(Q2_NB + cidx) * Q2_NB;
        double x[Q2_NB];
#pragma unroll
        for (int j = 0; j < Q2_NB; ++j) {
            double s = wc[j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= G[j][k] * x[k];
            x[j] = st[j] * s;
        }
#pragma unroll
        for (int j = 0; j < Q2_NB; ++j) W2[(size_t)cidx * Q2_NB + j] = x[j];
    }
}

// The same W2 = T'W with the 64-step recurrence taken off the critical path: the recurrence reads
//      (I + D N) x = D w,      D = diag(tau), N = strict lower triangle of G
// so T' = inv(I + D N) D, the inverse of a UNIT lower-triangular matrix (no divisions; tau_j = 0 gives a zero row
// as before).  Every workgroup builds it in LDS -- 16 x 16 diagonal blocks by substitution (one thread per
// column, 120 dependent FMAs instead of 2016), then two levels of  X_BA = -X_BB (Y_BA X_AA)  -- and multiplies
// its 64 columns of W by it on the MFMA units.  One launch, ~10 us instead of 54.
__global__ void __launch_bounds__(256)
k_qr1_tw_mfma(const double *__restrict__ W, int ncolsB, const double *__restrict__ tau, int c0, int nb,
              double *__restrict__ W2) {
    constexpr int LS = Q2_NB + 1;
    __shared__ double Y[Q2_NB * LS];      // D N (strictly lower)
    __shared__ double X[Q2_NB * LS];      // inv(I + Y), then T' = X D
    __shared__ double Tm[32 * 33];        // Y_BA X_AA of the current level
    __shared__ double st[Q2_NB];
    __shared__ double sB[Q2_NB * Q2_KS];
    __shared__ double sA[Q2_NB * Q2_KS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < Q2_NB) st[tid] = tid < nb ? tau[c0 + tid] : 0.0;
    __syncthreads();
    for (int e = tid; e < Q2_NB * Q2_NB; e += 256) {
        const int j = e / Q2_NB, k = e % Q2_NB;
        Y[j * LS + k] = k < j ? st[j] * W[(size_t)j * Q2_NB + k] : 0.0;    // W[cb = j][row = k] = v_k'v_j
        X[j * LS + k] = 0.0;
    }
    __syncthreads();
    if (tid < Q2_NB) {                    // 16 x 16 diagonal blocks: column c of inv(I + Y_bb)
        const int o = (tid >> 4) * 16, cc = tid & 15;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = r == cc ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= Y[(o + r) * LS + o + k] * x[k];
            x[r] = r >= cc ? acc : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(o + r) * LS + o + cc] = x[r];
    }
    __syncthreads();
    for (int sz = 16; sz < Q2_NB; sz *= 2) {
        const int npair = Q2_NB / (2 * sz);
        // Tm(pair)[r][c] = sum_k Y[B r][A k] X[A k][A c]   (X_AA lower triangular: k >= c)
        for (int e = tid; e < npair * sz * sz; e += 256) {
            const int pr = e / (sz * sz), r = (e / sz) % sz, cc = e % sz, oa = pr * 2 * sz, ob = oa + sz;
            double acc = 0.0;
#pragma unroll 8
            for (int k = cc; k < sz; ++k) acc += Y[(ob + r) * LS + oa + k] * X[(oa + k) * LS + oa + cc];
            Tm[(pr * sz + r) * 33 + cc] = acc;     // (two pairs of 16 rows or one of 32: 32 x 32 in all)
        }
        __syncthreads();
        // X_BA[r][c] = -sum_k X[B r][B k] Tm[k][c]       (X_BB lower triangular: k <= r)
        for (int e = tid; e < npair * sz * sz; e += 256) {
            const int pr = e / (sz * sz), r = (e / sz) % sz, cc = e % sz, oa = pr * 2 * sz, ob = oa + sz;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k <= r; ++k) acc += X[(ob + r) * LS + ob + k] * Tm[(pr * sz + k) * 33 + cc];
            X[(ob + r) * LS + oa + cc] = -acc;
        }
        __syncthreads();
    }
    // W2(:, 64 columns of this workgroup) = (X D) W(:, columns)
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int ncols = ncolsB - Q2_NB;
    const int j0 = blockIdx.x * Q2_NB;
    v4d_qr acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d_qr){0.0, 0.0, 0.0, 0.0};
    const int lc = tid >> 2, lk = (tid & 3) * 8;
    const double *wcol = j0 + lc < ncols ? W + (size_t)(Q2_NB + j0 + lc) * Q2_NB : nullptr;
    for (int k0 = 0; k0 < Q2_NB; k0 += Q2_KC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + lk + q;
            sA[lc * Q2_KS + lk + q] = X[lc * LS + k] * st[k];       // T'[m = lc][k]
            sB[lc * Q2_KS + lk + q] = wcol ? wcol[k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < Q2_KC; kk += 4) {
            const int ko = kk + (lane >> 4);
            const double a0 = sA[(wr + (lane & 15)) * Q2_KS + ko];
            const double a1 = sA[(wr + 16 + (lane & 15)) * Q2_KS + ko];
            const double b0 = sB[(wc + (lane & 15)) * Q2_KS + ko];
            const double b1 = sB[(wc + 16 + (lane & 15)) * Q2_KS + ko];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + (lane >> 4) + 4 * r;   // entry of W2's column
                const int col = j0 + wc + b * 16 + (lane & 15);
                if (col < ncols) W2[(size_t)col * Q2_NB + row] = acc[a][b][r];
            }
}

// A2(rows, 64 columns of tile) -= V(rows, :) * W2(:, columns)      (fp64 MFMA, K = 64)
constexpr int Q2_UCT = 4;   // column tiles per workgroup of the update: the V tile is staged once for all of them
__global__ void __launch_bounds__(256)
k_qr1_update(const double *__restrict__ Vb, int ldv, double *__restrict__ A, int M, int c0, int cend, int n,
             double *__restrict__ rhs, int ncols /* n - cend + 1 */, const double *__restrict__ W2) {
    constexpr int CS = Q2_NB + 1;             // product image [col][row], padded
    __shared__ double sV[Q2_NB * Q2_NB];     // [k][row]: row contiguous
    __shared__ double sW[Q2_NB * CS];        // [col][k] (64 x 64 used); after the MFMAs the product image
    const int rows = M - c0;
    const int nrt = (rows + Q2_NB - 1) / Q2_NB;
    const int rt = blockIdx.x % nrt, cg = blockIdx.x / nrt;
    const int r0 = rt * Q2_NB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const bool rin = r0 + lane < rows;
    // V and W2 images are exactly 64 x 64 doubles; bank conflicts are
    // avoided by rotation instead of padding: V column k is stored rotated by 16*(k&3) rows (the four
    // k-groups of an MFMA operand read then hit four different 128-byte segments), W2 column j by
    // 2*(j&15) entries (the 16 columns of an operand read hit 16 different bank pairs)
    if (r0 + Q2_NB <= rows) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = w + 4 * q;                       // V column k, row r0 + lane
            sV[k * Q2_NB + ((lane + 16 * (k & 3)) & 63)] = Vb[(size_t)k * ldv + r0 + lane];
        }
    } else {
        for (int e = tid; e < Q2_NB * Q2_NB; e += 256) {
            const int k = e / Q2_NB, r = e % Q2_NB;        // V column k, row r0 + r
            sV[k * Q2_NB + ((r + 16 * (k & 3)) & 63)] = (r0 + r < rows) ? Vb[(size_t)k * ldv + r0 + r] : 0.0;
        }
    }
    // tile t+1's operands (the A2 tile, row-contiguous: thread = row, 16 columns each; the W2 tile) are fetched
    // into registers while tile t is multiplied and written back: latency hidden, read-modify-write coalesced
    double at[16], wt[16], atn[16];
    auto fetch = [&](int j0, double *ta, double *tw2) {
        if (r0 + Q2_NB <= rows && j0 + Q2_NB < ncols) {      // interior tile of A2: unconditional fetches
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                ta[q] = (A + (size_t)(cend + j0 + w * 16 + q) * M + c0 + r0)[lane];
                const int e = tid + q * 256;
                tw2[q] = W2[(size_t)j0 * Q2_NB + e];                  // (column j0 + e / 64, entry e % 64)
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cidx = j0 + w * 16 + q;
                const int ac = cend + cidx;
                ta[q] = (rin && cidx < ncols) ? ((ac < n ? A + (size_t)ac * M : rhs) + c0 + r0)[lane] : 0.0;
                const int e = tid + q * 256, cw = e / Q2_NB, kk = e % Q2_NB;     // W2 column j0 + cw, entry kk
                tw2[q] = (j0 + cw < ncols) ? W2[(size_t)(j0 + cw) * Q2_NB + kk] : 0.0;
            }
        }
    };
    if (cg * Q2_UCT * Q2_NB < ncols) fetch(cg * Q2_UCT * Q2_NB, at, wt);
    for (int t = 0; t < Q2_UCT; ++t) {
        const int j0 = (cg * Q2_UCT + t) * Q2_NB;
        if (j0 >= ncols) break;                        // (uniform)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + q * 256, cw = e / Q2_NB, kk = e % Q2_NB;
            sW[cw * Q2_NB + ((kk + 2 * (cw & 15)) & 63)] = wt[q];
        }
        __syncthreads();
        const bool more = t + 1 < Q2_UCT && j0 + Q2_NB < ncols;
        if (more) fetch(j0 + Q2_NB, atn, wt);
        v4d_qr acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (v4d_qr){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < Q2_NB; kk += 4) {
            const int ko = kk + (lane >> 4);
            const int rot = 16 * (ko & 3);
            const int c0w = wc + (lane & 15), c1w = wc + 16 + (lane & 15);
            const double a0 = sV[ko * Q2_NB + ((wr + (lane & 15) + rot) & 63)];
            const double a1 = sV[ko * Q2_NB + ((wr + 16 + (lane & 15) + rot) & 63)];
            const double b0 = sW[c0w * Q2_NB + ((ko + 2 * (c0w & 15)) & 63)];
            const double b1 = sW[c1w * Q2_NB + ((ko + 2 * (c1w & 15)) & 63)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();                                 // every wave is done with the W2 image
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr + a * 16 + (lane >> 4) + 4 * r;
                    const int cidx = wc + b * 16 + (lane & 15);
                    sW[cidx * CS + row] = acc[a][b][r];
                }
        __syncthreads();
        if (r0 + Q2_NB <= rows && j0 + Q2_NB < ncols) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cl = w * 16 + q;
                (A + (size_t)(cend + j0 + cl) * M + c0 + r0)[lane] = at[q] - sW[cl * CS + lane];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cl = w * 16 + q, cidx = j0 + cl;
                if (rin && cidx < ncols) {
                    const int ac = cend + cidx;
                    ((ac < n ? A + (size_t)ac * M : rhs) + c0 + r0)[lane] = at[q] - sW[cl * CS + lane];
                }
            }
        }
        __syncthreads();                                 // the product image is consumed before the next W2 tile lands
#pragma unroll
        for (int q = 0; q < 16; ++q) at[q] = atn[q];
    }
}

// R (upper triangle of the factored A, zeros below) and the first n entries of Q1'b -> stage-2 operands
__global__ void __launch_bounds__(256)
k_qr1_extract(const double *__restrict__ A, int M, int n, const double *__restrict__ rhs, double *__restrict__ R,
              double *__restrict__ rhs2) {
    const long long tot = (long long)n * n;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int r = (int)(e % n), cidx = (int)(e / n);
        R[e] = r <= cidx ? A[(size_t)cidx * M + r] : 0.0;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) rhs2[i] = rhs[i];
}

// ---- stage 2, one launch per column ----------------------------------------------------------
// The pivoted sweep on the n x n triangle with LAZY column exchanges: physical columns never move;
// colat[pos] names the column standing at position pos (dgeqp3's idamax runs over positions, so ties
// -- e.g. the all-zero norms of a rank-deficient tail -- resolve exactly as with physical swaps).
// Every workgroup of step i redundantly (a) finds the pivot position, (b) builds H_i from the pivot
// column in registers, then (c) applies it to its own column and downdates that column's norm.
// Norms and the position map are double-buffered (read *_in, write *_out): a block must see the
// norms as they were when the step began.  Block 0 also records beta, tau and the new position map.
constexpr int Q2S_NT = 256;
constexpr int Q2S_RPT = 8;    // rows per thread: n - i <= 2048
__device__ __forceinline__ double blk_sum_256(double v, double *sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
// Q2S_CPB columns per workgroup: the pivot column (fetched and turned into the reflector by every workgroup) is
// shared by that many updates and a step has that many times fewer workgroups (worth it while the trailing
// matrix is wide; near the end one column per workgroup has the shorter critical path)
template <int Q2S_CPB>
__global__ void __launch_bounds__(Q2S_NT)
k_qr2_step(double *__restrict__ R, int n, int i, double *__restrict__ rhs, const double *__restrict__ vn1_in,
           const double *__restrict__ vn2_in, double *__restrict__ vn1_out, double *__restrict__ vn2_out,
           const int *__restrict__ colat_in, int *__restrict__ colat_out, double *__restrict__ tau,
           double *__restrict__ diag, double *__restrict__ ice /* wmin[n] wmax[n] smin smax stopped */, double rcond,
           int *__restrict__ rank_out) {
    __shared__ double sh[4];
    __shared__ double shv[4][Q2S_CPB];
    __shared__ double s_best[4];
    __shared__ int s_bpos[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int npos = n - i - 1;                 // positions i+1 .. n-1; item npos is the right-hand side
    const int nitems = npos + 1;
    const int nblk = (nitems + Q2S_CPB - 1) / Q2S_CPB;
    const bool is_ice = (int)blockIdx.x == nblk;
    // the columns standing at this block's positions are fetched right away (the norms are indexed by POSITION,
    // so the pivot search needs no indirection); only a column sitting at the pivot's position has to be
    // fetched again -- that slot works on the column the exchange brings there
    // every column is reached through a buffer descriptor (uniform base, n*8 bytes; an absent column gets an empty one):
    // the fetches are unconditional -- rows beyond n read as zero, their stores are dropped -- instead of one exec-masked
    // branch per element
    typedef unsigned v2u_q2 __attribute__((ext_vector_type(2)));
    const unsigned tb = (unsigned)(i + 1 + tid) * 8u, ib = (unsigned)i * 8u, colbytes = (unsigned)n * 8u;
    auto col_rsrc = [&](const double *base, bool present) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, present ? colbytes : 0u, 0x00020000);
    };
    auto ld = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
    };
    double a[Q2S_CPB][Q2S_RPT], cji[Q2S_CPB];
    int own[Q2S_CPB];
#pragma unroll
    for (int c = 0; c < Q2S_CPB; ++c) {
        const int item = (int)blockIdx.x * Q2S_CPB + c;
        own[c] = -1;                                            // -1: nothing, -2: rhs
        const double *cg = nullptr;
        if (!is_ice && item < npos) {
            own[c] = __builtin_amdgcn_readfirstlane(colat_in[i + 1 + item]);
            cg = R + (size_t)own[c] * n;
        } else if (!is_ice && item == npos) { own[c] = -2; cg = rhs; }
        const __amdgpu_buffer_rsrc_t rc = col_rsrc(cg ? cg : R, cg != nullptr);
#pragma unroll
        for (int q = 0; q < Q2S_RPT; ++q) a[c][q] = ld(rc, tb, q * Q2S_NT * 8);
        cji[c] = ld(rc, ib, 0);
    }
    // (a) first maximum of the norms over positions i..n-1
    double best = -1.0;
    int bpos = n;
    for (int pos = i + tid; pos < n; pos += Q2S_NT) {
        const double v = vn1_in[pos];
        if (v > best) { best = v; bpos = pos; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int op = __shfl_down(bpos, o, 64);
        if (ob > best || (ob == best && op < bpos)) { best = ob; bpos = op; }
    }
    if (lane == 0) { s_best[wv] = best; s_bpos[wv] = bpos; }
    __syncthreads();
    best = s_best[0]; bpos = s_bpos[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_best[w] > best || (s_best[w] == best && s_bpos[w] < bpos)) { best = s_best[w]; bpos = s_bpos[w]; }
    const int ppos = bpos < n ? bpos : i;
    const int pcol = __builtin_amdgcn_readfirstlane(colat_in[ppos]), icol = __builtin_amdgcn_readfirstlane(colat_in[i]);
    // (b) reflector of the pivot column on rows i..n-1
    const double *cp = R + (size_t)pcol * n;
    const __amdgpu_buffer_rsrc_t rp = col_rsrc(cp, true);
    double v[Q2S_RPT];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < Q2S_RPT; ++q) {
        v[q] = ld(rp, tb, q * Q2S_NT * 8);
        acc += v[q] * v[q];
    }
    const double alpha = ld(rp, ib, 0);
#pragma unroll
    for (int c = 0; c < Q2S_CPB; ++c) {
        const int item = (int)blockIdx.x * Q2S_CPB + c;
        if (own[c] >= 0 && i + 1 + item == ppos) {   // this position receives the displaced column
            own[c] = icol;
            const __amdgpu_buffer_rsrc_t ro = col_rsrc(R + (size_t)icol * n, true);
#pragma unroll
            for (int q = 0; q < Q2S_RPT; ++q) a[c][q] = ld(ro, tb, q * Q2S_NT * 8);
            cji[c] = ld(ro, ib, 0);
        }
    }
    const double xn = sqrt(blk_sum_256(acc, sh));
    double ti = 0.0, beta = alpha;
    if (xn != 0.0) {
        beta = -copysign(hypot(alpha, xn), alpha);
        ti = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
#pragma unroll
        for (int q = 0; q < Q2S_RPT; ++q) v[q] *= sc;
    }
    if (is_ice) {
        // the extra workgroup: step i of xGELSY's incremental condition estimate (dlaic1) on the column that
        // has just become final -- R(0:i-1, i) = rows above the diagonal of the pivot column, R(i,i) = beta --
        // so the rank is known when the sweep ends instead of after n more dependent steps
        double *wmin = ice, *wmax = ice + n, *sc = ice + 2 * n;   // sc: smin, smax, stopped
        if (i == 0) {
            if (tid == 0) {
                const double a0 = fabs(beta);
                sc[0] = a0; sc[1] = a0;
                sc[2] = a0 == 0.0 ? 1.0 : 0.0;
                wmin[0] = 1.0; wmax[0] = 1.0;
                *rank_out = a0 == 0.0 ? 0 : 1;
            }
            return;
        }
        if (sc[2] != 0.0) return;   // rank already decided
        double a1 = 0.0, a2 = 0.0;
        for (int k = tid; k < i; k += Q2S_NT) {
            const double ck = cp[k];
            a1 += wmin[k] * ck;
            a2 += wmax[k] * ck;
        }
        a1 = blk_sum_256(a1, sh);
        a2 = blk_sum_256(a2, sh);
        double sminpr, s1, c1, smaxpr, s2, c2;
        laic1_dev(2, a1, sc[0], beta, &sminpr, &s1, &c1);
        laic1_dev(1, a2, sc[1], beta, &smaxpr, &s2, &c2);
        __syncthreads();   // every thread has read sc[] before thread 0 rewrites it
        if (smaxpr * rcond > sminpr) {
            if (tid == 0) sc[2] = 1.0;
            return;
        }
        for (int k = tid; k < i; k += Q2S_NT) { wmin[k] *= s1; wmax[k] *= s2; }
        if (tid == 0) {
            wmin[i] = c1; wmax[i] = c2;
            sc[0] = sminpr; sc[1] = smaxpr;
            *rank_out = i + 1;
        }
        return;
    }
    // (c) apply H_i to the block's columns: the four dot products share one reduction
    if (ti != 0.0) {
        double w[Q2S_CPB];
#pragma unroll
        for (int c = 0; c < Q2S_CPB; ++c) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < Q2S_RPT; ++q) t += v[q] * a[c][q];
            w[c] = wave_sum(t);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < Q2S_CPB; ++c) shv[wv][c] = w[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < Q2S_CPB; ++c) {
            if (own[c] == -1) continue;
            const double wt = (((shv[0][c] + shv[1][c]) + shv[2][c]) + shv[3][c]) + cji[c];      // v_i = 1
            const double tw = ti * wt;
            double *cj = own[c] == -2 ? rhs : R + (size_t)own[c] * n;
            const __amdgpu_buffer_rsrc_t rj = col_rsrc(cj, true);
#pragma unroll
            for (int q = 0; q < Q2S_RPT; ++q) {
                a[c][q] -= v[q] * tw;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_q2, a[c][q]), rj, tb, q * Q2S_NT * 8, 0);
            }
            cji[c] -= tw;
            if (tid == 0) cj[i] = cji[c];
        }
        __syncthreads();
    }
    // partial-norm downdate (dlaqp2); the norms travel with the POSITION
    const double tol3z = sqrt(DBL_EPSILON / 2);
#pragma unroll
    for (int c = 0; c < Q2S_CPB; ++c) {
        if (own[c] < 0) continue;                            // nothing, or the right-hand side
        const int mypos = i + 1 + (int)blockIdx.x * Q2S_CPB + c;
        const int from = mypos == ppos ? i : mypos;          // where this column stood when the step began
        const double v1 = vn1_in[from], v2 = vn2_in[from];
        double n1 = v1, n2 = v2;
        if (v1 != 0.0) {
            const double r = fabs(cji[c]) / v1;
            const double temp = fmax(1.0 - r * r, 0.0);
            const double qq = v1 / v2;
            const double temp2 = temp * qq * qq;
            if (temp2 <= tol3z) {
                double a2 = 0.0;
#pragma unroll
                for (int q = 0; q < Q2S_RPT; ++q) a2 += a[c][q] * a[c][q];
                a2 = blk_sum_256(a2, sh);
                n1 = i < n - 1 ? sqrt(a2) : 0.0;
                n2 = n1;
            } else {
                n1 = v1 * sqrt(temp);
            }
        }
        if (tid == 0) { vn1_out[mypos] = n1; vn2_out[mypos] = n2; }
    }
    if (blockIdx.x == 0) {   // bookkeeping of the step
        if (tid == 0) {
            diag[i] = beta;   // (not into R: other workgroups of this step still read the pivot column)
            tau[i] = ti;
        }
        for (int pos = tid; pos < n; pos += Q2S_NT) {
            int cidx = colat_in[pos];
            if (pos == i) cidx = pcol;
            else if (pos == ppos) cidx = icol;
            colat_out[pos] = cidx;
        }
    }
}

// R in pivoted order for the solve: G(0:pos, pos) = R(0:pos, colat[pos]); jp[pos] = colat[pos]
__global__ void __launch_bounds__(256)
k_qr2_gather(const double *__restrict__ R, int n, const int *__restrict__ colat, const double *__restrict__ diag,
             double *__restrict__ G, int *__restrict__ jp) {
    const long long tot = (long long)n * n;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int r = (int)(e % n), pos = (int)(e / n);
        G[e] = r < pos ? R[(size_t)colat[pos] * n + r] : (r == pos ? diag[pos] : 0.0);
    }
    for (int pos = blockIdx.x * 256 + threadIdx.x; pos < n; pos += gridDim.x * 256) jp[pos] = colat[pos];
}
__global__ void __launch_bounds__(256)
k_qr2_init(const double *__restrict__ R, int n, double *__restrict__ vn1, double *__restrict__ vn2, int *__restrict__ colat) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *c = R + (size_t)j * n;
    double acc = 0.0;
    for (int k = lane; k < n; k += 64) acc += c[k] * c[k];
    acc = wave_sum(acc);
    if (lane == 0) { double v = sqrt(acc); vn1[j] = v; vn2[j] = v; colat[j] = j; }
}

// ---- full-rank certificate --------------------------------------------------------------------
// The pivoted sweep (stage 2) is n dependent launches; it only matters when xGELSY's rank decision can come out
// below n.  That decision compares dlaic1's estimates on the pivoted triangle: smaxpr is ||R11'x|| for a unit x,
// so smaxpr <= sigma_max(R11) <= sigma_max(A), and sminpr >= sigma_min(R11) >= sigma_min(A) (R11 spans a
// subset of A's columns).  Hence  cond_2(A) * rcond <= 1  PROVES that every step keeps the column, i.e.
// rank = n, and then the solution is the unique least-squares solution, which the unpivoted triangle of
// stage 1 yields just as well.  cond_2(A) = cond_2(R) <= ||R||_F ||inv(R)||_F is computed rigorously from the
// explicit inverse X of the stage-1 triangle: 64 x 64 diagonal blocks inverted one workgroup each, then
// log2(n/64) levels of  X12 = -X11 (R12 X22)  as batched fp64-MFMA tile products (n^3/3 flops in all).
// If the bound (with a safety factor) does not certify full rank -- or is not finite -- stage 2 runs as before.
__global__ void __launch_bounds__(256)
k_tri_diaginv(const double *__restrict__ R, int n, double *__restrict__ X, int ldx, size_t bstride) {
    // block d -> X + d * bstride, element (r, c) at [c * ldx + r]  (in place in an n x n image: ldx = n, bstride = 64 n + 64)
    // 16 x 16 diagonal sub-blocks by back substitution (one thread per column, registers), then two levels of
    // X_AB = -X_AA (R_AB X_BB) with all threads: the dependent chain is 16 steps instead of 64
    constexpr int LS = 65;
    __shared__ double sR[64 * LS];
    __shared__ double sX[64 * LS];
    __shared__ double Tm[32 * 33];
    const int o = blockIdx.x * 64, nb = min(64, n - o), tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e % 64, cidx = e / 64;
        sR[r * LS + cidx] = (r < nb && cidx < nb && r <= cidx) ? R[(size_t)(o + cidx) * n + o + r] : (r == cidx ? 1.0 : 0.0);
        sX[r * LS + cidx] = 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        const int ob = (tid >> 4) * 16, cc = tid & 15;
        double x[16];
#pragma unroll
        for (int r = 15; r >= 0; --r) {
            double acc = r == cc ? 1.0 : 0.0;
#pragma unroll
            for (int k = r + 1; k < 16; ++k) acc -= sR[(ob + r) * LS + ob + k] * x[k];
            x[r] = r <= cc ? acc / sR[(ob + r) * LS + ob + r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sX[(ob + r) * LS + ob + cc] = x[r];
    }
    __syncthreads();
    for (int sz = 16; sz < 64; sz *= 2) {
        const int npair = 64 / (2 * sz);
        // Tm(pair)[r][c] = sum_k R[A r][B k] X[B k][B c]     (X_BB upper triangular: k <= c)
        for (int e = tid; e < npair * sz * sz; e += 256) {
            const int pr = e / (sz * sz), r = (e / sz) % sz, cc = e % sz, oa = pr * 2 * sz, ob = oa + sz;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k <= cc; ++k) acc += sR[(oa + r) * LS + ob + k] * sX[(ob + k) * LS + ob + cc];
            Tm[(pr * sz + r) * 33 + cc] = acc;
        }
        __syncthreads();
        // X_AB[r][c] = -sum_k X[A r][A k] Tm[k][c]            (X_AA upper triangular: k >= r)
        for (int e = tid; e < npair * sz * sz; e += 256) {
            const int pr = e / (sz * sz), r = (e / sz) % sz, cc = e % sz, oa = pr * 2 * sz, ob = oa + sz;
            double acc = 0.0;
#pragma unroll 8
            for (int k = r; k < sz; ++k) acc += sX[(oa + r) * LS + oa + k] * Tm[(pr * sz + k) * 33 + cc];
            sX[(oa + r) * LS + ob + cc] = -acc;
        }
        __syncthreads();
    }
    double *out = X + (size_t)blockIdx.x * bstride;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e % 64, cidx = e / 64;
        if (r < nb && cidx < nb) out[(size_t)cidx * ldx + r] = sX[r * LS + cidx];
    }
}

// one level of the recursion, blocks of size s: phase 0  T12 = R12 * X22,  phase 1  X12 = -X11 * T12
// (64 x 64 output tile per workgroup; the k-range is cut to the non-zero part of the triangular factor)
__global__ void __launch_bounds__(256)
k_tri_level(const double *__restrict__ R, double *__restrict__ X, double *__restrict__ T, int n, int s, int phase) {
    __shared__ double sA[Q2_NB * Q2_KS];
    __shared__ double sB[Q2_NB * Q2_KS];
    const int tps = s / 64, tpp = tps * tps;
    const int p = blockIdx.x / tpp, tt = blockIdx.x % tpp;
    const int tm = tt % tps, tn = tt / tps;
    const int o = 2 * p * s;
    const int N2 = min(s, n - o - s);
    if (N2 <= 0 || tn * 64 >= N2) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    // C(m, c) = alpha * sum_k A(m, k) B(k, c);  A(m, k) = Ap[k * n + m], B(k, c) = Bp[c * n + k]
    const double *Ap, *Bp;
    double *Cp;
    int kb, ke;
    if (phase == 0) {
        Ap = R + (size_t)(o + s) * n + o;          // R12
        Bp = X + (size_t)(o + s) * n + o + s;      // X22 (upper triangular: k <= c)
        Cp = T + (size_t)(o + s) * n + o;
        kb = 0; ke = min(N2, tn * 64 + 64);
    } else {
        Ap = X + (size_t)o * n + o;                // X11 (upper triangular: k >= m)
        Bp = T + (size_t)(o + s) * n + o;
        Cp = X + (size_t)(o + s) * n + o;
        kb = tm * 64; ke = s;
    }
    const double alpha = phase == 0 ? 1.0 : -1.0;
    v4d_qr acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d_qr){0.0, 0.0, 0.0, 0.0};
    const int am = tid & 63, akq = (tid >> 6) * 8;           // A staging: lane = row (coalesced), 8 k's per thread
    const int lc = tid >> 2, lk = (tid & 3) * 8;             // B staging: 8 consecutive k's of one column
    const int m0 = tm * 64, c0 = tn * 64;
    const bool cin = c0 + lc < N2;
    double ra[8], rb[8];
    const double *bcol = Bp + (size_t)(c0 + (cin ? lc : 0)) * n;
    auto fetch = [&](int k0) {
        if (k0 + Q2_KC <= ke) {                // full slab: unconditional fetches
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ra[q] = Ap[(size_t)(k0 + akq + q) * n + m0 + am];
                const double y = bcol[k0 + lk + q];
                rb[q] = cin ? y : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ka = k0 + akq + q;
                ra[q] = ka < ke ? Ap[(size_t)ka * n + m0 + am] : 0.0;
                const int k = k0 + lk + q;
                rb[q] = (cin && k < ke) ? Bp[(size_t)(c0 + lc) * n + k] : 0.0;
            }
        }
    };
    if (kb < ke) fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += Q2_KC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sA[am * Q2_KS + akq + q] = ra[q];
            sB[lc * Q2_KS + lk + q] = rb[q];
        }
        __syncthreads();
        if (k0 + Q2_KC < ke) fetch(k0 + Q2_KC);
#pragma unroll
        for (int kk = 0; kk < Q2_KC; kk += 4) {
            const int ko = kk + (lane >> 4);
            const double a0 = sA[(wr + (lane & 15)) * Q2_KS + ko];
            const double a1 = sA[(wr + 16 + (lane & 15)) * Q2_KS + ko];
            const double b0 = sB[(wc + (lane & 15)) * Q2_KS + ko];
            const double b1 = sB[(wc + 16 + (lane & 15)) * Q2_KS + ko];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr + a * 16 + (lane >> 4) + 4 * r;
                const int col = c0 + wc + b * 16 + (lane & 15);
                if (col < N2) Cp[(size_t)col * n + row] = alpha * acc[a][b][r];
            }
}

// partial sums of squares over the upper triangles of R and X (fixed order; the host adds the partials in order)
__global__ void __launch_bounds__(256)
k_tri_fro(const double *__restrict__ R, const double *__restrict__ X, int n, double *__restrict__ part) {
    __shared__ double sh[4];
    double a = 0.0, b = 0.0;
    for (int cidx = blockIdx.x; cidx < n; cidx += gridDim.x)
        for (int r = threadIdx.x; r <= cidx; r += 256) {
            const double x = R[(size_t)cidx * n + r], y = X[(size_t)cidx * n + r];
            a += x * x;
            b += y * y;
        }
    a = block_sum<256>(a, sh);
    __syncthreads();
    b = block_sum<256>(b, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}

__global__ void k_tri_identity(int *__restrict__ jp, int n, int *__restrict__ rank) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) jp[k] = k;
    if (blockIdx.x == 0 && threadIdx.x == 0) *rank = n;
}

// R z = c with the inverted diagonal blocks of the certificate, ONE launch: workgroup t owns rows 64t .. 64t+63,
// subtracts R(t, e) z_e for e = last .. t+1 as the z_e arrive (flag-in-data slots, as in k_qr1_step_multi),
// then forms z_t = X_tt (c_t - ...) and publishes it.  Workgroups are numbered so that a workgroup only waits
// for workgroups dispatched BEFORE it (no co-residency assumption); the next R tile is fetched before the wait,
// so the chain  z_e -> z_{e-1}  costs one exchange plus two 64 x 64 products from registers and LDS.
__global__ void __launch_bounds__(256)
k_tri_bsolve(const double *__restrict__ R, const double *__restrict__ X, int ldx, size_t bstride /* as k_tri_diaginv */,
             int n, const double *__restrict__ cvec, double *__restrict__ x,
             unsigned long long *__restrict__ slot /* [nblk][64][2] */, unsigned long long epoch, int *__restrict__ err) {
    __shared__ double sc[64], sz[64], sp[4][64];
    const int nblk = (n + 63) / 64;
    const int t = nblk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, row = tid & 63, part = tid >> 6;
    const int r0 = t * 64;
    const bool rin = r0 + row < n;
    const unsigned ep = (unsigned)epoch;
    if (tid < 64) sc[tid] = rin ? cvec[r0 + tid] : 0.0;
    double tile[16];
    auto fetch = [&](const double *Mx, int e) {      // rows r0.., columns 64e + 16 part + q
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cidx = e * 64 + part * 16 + q;
            tile[q] = (rin && cidx < n) ? Mx[(size_t)cidx * n + r0 + row] : 0.0;
        }
    };
    auto fetch_diag = [&]() {
        const double *Xt = X + (size_t)t * bstride;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cl = part * 16 + q;
            tile[q] = (rin && r0 + cl < n) ? Xt[(size_t)cl * ldx + row] : 0.0;
        }
    };
    auto apply = [&](double sign) {                  // sc += sign * tile * sz   (sz: 64 entries in LDS)
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += tile[q] * sz[part * 16 + q];
        sp[part][row] = acc;
        __syncthreads();
        if (tid < 64) sc[tid] += sign * (((sp[0][tid] + sp[1][tid]) + sp[2][tid]) + sp[3][tid]);
        __syncthreads();
    };
    for (int e = nblk - 1; e > t; --e) {
        fetch(R, e);
        if (tid < 64) {
            const unsigned long long *f = slot + ((size_t)e * 64 + tid) * 2;
            unsigned long long w0 = __hip_atomic_load(f, RLX_AGENT), w1 = __hip_atomic_load(f + 1, RLX_AGENT);
            int spins = 0;
            while ((unsigned)(w0 >> 32) != ep || (unsigned)(w1 >> 32) != ep) {
                if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); break; }
                __builtin_amdgcn_s_sleep(1);
                w0 = __hip_atomic_load(f, RLX_AGENT);
                w1 = __hip_atomic_load(f + 1, RLX_AGENT);
            }
            sz[tid] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
        }
        __syncthreads();
        apply(-1.0);
    }
    // z_t = X_tt c_t (X_tt upper triangular with zeros below: the certificate wrote the full block)
    fetch_diag();
    if (tid < 64) { sz[tid] = sc[tid]; sc[tid] = 0.0; }
    __syncthreads();
    apply(1.0);
    if (tid < 64) {
        const double z = sc[tid];
        unsigned long long *mine = slot + ((size_t)t * 64 + tid) * 2;
        const unsigned long long hi = (unsigned long long)ep << 32;
        __hip_atomic_store(mine, hi | (unsigned)__double2loint(z), RLX_AGENT);
        __hip_atomic_store(mine + 1, hi | (unsigned)__double2hiint(z), RLX_AGENT);
        if (rin) x[r0 + tid] = z;
    }
}

// U'z = b, the forward half of a Cholesky solve, same scheme: workgroup t owns unknowns 64t .., subtracts
// U(e, t)' z_e for e = 0 .. t-1 as they arrive and applies X_tt' (thread = column of the tile, 16 rows each:
// the transposed product needs no cross-lane sums; the tiles are L2-resident)
__global__ void __launch_bounds__(256)
k_tri_fsolve_t(const double *__restrict__ U, const double *__restrict__ X, int ldx, size_t bstride, int n,
               const double *__restrict__ bvec, double *__restrict__ z, unsigned long long *__restrict__ slot,
               unsigned long long epoch, int *__restrict__ err) {
    __shared__ double sc[64], sz[64], sp[4][64];
    const int t = (int)blockIdx.x;
    const int tid = threadIdx.x, col = tid & 63, part = tid >> 6;
    const int c0 = t * 64;
    const bool cin = c0 + col < n;
    const unsigned ep = (unsigned)epoch;
    if (tid < 64) sc[tid] = cin ? bvec[c0 + tid] : 0.0;
    double tile[16];
    auto apply = [&](double sign) {                  // sc += sign * tile' * sz
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += tile[q] * sz[part * 16 + q];
        sp[part][col] = acc;
        __syncthreads();
        if (tid < 64) sc[tid] += sign * (((sp[0][tid] + sp[1][tid]) + sp[2][tid]) + sp[3][tid]);
        __syncthreads();
    };
    for (int e = 0; e < t; ++e) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[q] = cin ? U[(size_t)(c0 + col) * n + e * 64 + part * 16 + q] : 0.0;
        if (tid < 64) {
            const unsigned long long *f = slot + ((size_t)e * 64 + tid) * 2;
            unsigned long long w0 = __hip_atomic_load(f, RLX_AGENT), w1 = __hip_atomic_load(f + 1, RLX_AGENT);
            int spins = 0;
            while ((unsigned)(w0 >> 32) != ep || (unsigned)(w1 >> 32) != ep) {
                if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); break; }
                __builtin_amdgcn_s_sleep(1);
                w0 = __hip_atomic_load(f, RLX_AGENT);
                w1 = __hip_atomic_load(f + 1, RLX_AGENT);
            }
            sz[tid] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
        }
        __syncthreads();
        apply(-1.0);
    }
    {
        const double *Xt = X + (size_t)t * bstride;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = part * 16 + q;
            tile[q] = (cin && c0 + r < n) ? Xt[(size_t)col * ldx + r] : 0.0;     // X_tt(r, col)
        }
    }
    if (tid < 64) { sz[tid] = sc[tid]; sc[tid] = 0.0; }
    __syncthreads();
    apply(1.0);
    if (tid < 64) {
        const double v = sc[tid];
        unsigned long long *mine = slot + ((size_t)t * 64 + tid) * 2;
        const unsigned long long hi = (unsigned long long)ep << 32;
        __hip_atomic_store(mine, hi | (unsigned)__double2loint(v), RLX_AGENT);
        __hip_atomic_store(mine + 1, hi | (unsigned)__double2hiint(v), RLX_AGENT);
        if (cin) z[c0 + tid] = v;
    }
}

// x = X c for the upper-triangular inverse (n beyond the single-workgroup substitution): one wavefront per row
__global__ void __launch_bounds__(256)
k_tri_matvec(const double *__restrict__ X, int n, const double *__restrict__ cvec, double *__restrict__ x) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    double acc = 0.0;
    for (int k = r + lane; k < n; k += 64) acc += X[(size_t)k * n + r] * cvec[k];
    acc = wave_sum(acc);
    if (lane == 0) x[r] = acc;
}

struct Qr2Work {
    double *Vb = nullptr, *Wp = nullptr, *W = nullptr, *W2 = nullptr, *R = nullptr, *rhs2 = nullptr, *tau1 = nullptr;
    double *vn = nullptr;     // stage 2: vn1/vn2 double-buffered (4n)
    double *ice = nullptr;    // stage 2: condition-estimate vectors + scalars (2n + 8)
    double *lazy = nullptr;   // stage 1, lazy reflectors: beta[n] | scale[n]
    double *Xinv = nullptr, *T2 = nullptr, *fro = nullptr, *h_fro = nullptr;   // full-rank certificate (h_fro pinned)
    unsigned long long *bslot = nullptr;   // certified solve: z blocks in flight [256][64][2 words]
    unsigned long long *xslot = nullptr;   // stage 1, slab exchange: [64 groups][8 slabs][8 rounds][18 sums][2 words]
    unsigned long long epoch = 0;
    int *d_err = nullptr;                  //   set when a slab wait gave up
    bool no_exchange = false;              //   ... after which this solver uses neither slabs nor the pipelined solve
    double *Pn = nullptr;                  // stage 1: side panel (M x 64) for the later pivot columns of a launch
    double *tsS[2] = {nullptr, nullptr}, *tsr[2] = {nullptr, nullptr};   // TSQR levels (ping-pong): stacked slab triangles ((slabs*n) x n) and Q'b entries
    int *colat = nullptr;     // stage 2: position map, double-buffered (2n)
    int kslices = 0, M = 0, n = 0;
    CqrWork cq;               // stage 1: CholeskyQR2 panel (lsq_qr_cholqr.hip)
    bool no_cholqr = false;   //   ... off for this solver after a breakdown (ill-conditioned / rank-deficient panels)
    bool cholqr_used = false; //   the current factorisation took it at least once
};
static void qr2_free(void *p) {
    Qr2Work *q = (Qr2Work *)p;
    if (!q) return;
    hipFree(q->Vb); hipFree(q->Wp); hipFree(q->W); hipFree(q->W2); hipFree(q->R); hipFree(q->rhs2); hipFree(q->tau1);
    hipFree(q->vn); hipFree(q->colat); hipFree(q->ice); hipFree(q->lazy);
    hipFree(q->Xinv); hipFree(q->T2); hipFree(q->fro); hipFree(q->xslot); hipFree(q->bslot); hipFree(q->d_err); hipFree(q->Pn); hipFree(q->tsS[0]); hipFree(q->tsS[1]); hipFree(q->tsr[0]); hipFree(q->tsr[1]);
    if (q->h_fro) hipHostFree(q->h_fro);
    lsq_cqr_free(&q->cq);
    delete q;
}

static bool qr2_applies(int M, int n) {
    const bool off = getenv("LSQ_QR_ONE_STAGE") != nullptr;     // (read per call: the tests flip them)
    const bool force = getenv("LSQ_QR_TWO_STAGE") != nullptr;
    if (off || M < n || n < 2) return false;
    // (measured crossover against the one-workgroup / two-launch pivoted kernels: 20 x 5 0.06 vs 0.13 ms,
    //  100 x 20 0.24 vs 0.17, 200 x 50 0.78 vs 0.25, 2000 x 400 10.1 vs 1.7)
    return force || (n >= 16 && (long long)M * n >= 1600) || (n >= 2 && (long long)M * n >= 20000);   // (tall and thin: 100000 x 10 1.2 vs 8.0 ms)
}

// factors [A | b] (s->d_qr, s->d_qu) and leaves R2, pivots (jp) and Q'b for k_qrcp_solve(phase 0) in the
// stage-2 buffers; returns them through the out parameters
static int qr2_factor_core(lsq_solver *s, double *A, double *rhs, int M, int n, double **R_out, double **rhs_out);
// tall and thin operands (n <= 32, many rows): level 0 of a TSQR -- one pass over the matrix, every workgroup
// factors its own slab in registers -- then the stacked triangles go through the regular panel machinery
static int qr2_tsqr_slab_rows(int n) {   // rows of a 256-thread slab ((K + 1) * RPT doubles of registers per thread); a wave slab is a quarter
    return 256 * (n <= 8 ? 8 : n <= 12 ? 6 : n <= 16 ? 4 : n <= 24 ? 3 : 2);
}
static bool qr2_tsqr_applies(int M, int n) {
    if (getenv("LSQ_QR_NO_TSQR") || n > 32 || n < 2) return false;
    return getenv("LSQ_QR_TSQR") ? M >= 2 * qr2_tsqr_slab_rows(n) : M >= 32768;
}
static int qr2_workspace(lsq_solver *s, int M, int n);

static int qr2_factor(lsq_solver *s, int M, int n, double **R_out, double **rhs_out) {
    lsq_ctx *c = s->ctx;
    LSQ_TRY(qr2_workspace(s, M, n));
    Qr2Work *q = (Qr2Work *)s->qr2;
    if (qr2_tsqr_applies(M, n)) {
        // levels of wave slabs (64 * RPT rows each) until the stack is short enough for the panel machinery
        const int L = qr2_tsqr_slab_rows(n) / 4;
        double *Acur = s->d_qr, *bcur = s->d_qu;
        int Mcur = M;
        for (int level = 0; level < 6 && qr2_tsqr_applies(Mcur, n); ++level) {
            const int S = lsq_div_up(Mcur, L), Ms = S * n;
            if (!q->tsS[0]) {
                for (int u = 0; u < 2; ++u) {
                    LSQ_HIP(hipMalloc(&q->tsS[u], ((size_t)Ms * n + 32768) * sizeof(double)));
                    LSQ_HIP(hipMalloc(&q->tsr[u], ((size_t)Ms + 32768) * sizeof(double)));
                }
            }
            double *So = q->tsS[level & 1], *ro = q->tsr[level & 1];
            auto go = [&](auto kern, int slabs_per_block) {
                hipLaunchKernelGGL(kern, dim3(lsq_div_up(S, slabs_per_block)), dim3(256), 0, c->stream, Acur, Mcur, n, 0, n, q->tau1,
                                   q->lazy, q->lazy + n, S, q->xslot, ++q->epoch, q->d_err, q->Pn, 0, bcur, So, Ms, ro);
            };
            if (n <= 8) go(k_qr1_step_multi<256, 8, 8, 1, 2>, 4);
            else if (n <= 12) go(k_qr1_step_multi<256, 6, 12, 1, 2>, 4);
            else if (n <= 16) go(k_qr1_step_multi<256, 4, 16, 1, 2>, 4);
            else if (n <= 20) go(k_qr1_step_multi<256, 3, 20, 1, 2>, 4);
            else if (n <= 24) go(k_qr1_step_multi<256, 3, 24, 1, 2>, 4);
            else if (n <= 28) go(k_qr1_step_multi<256, 2, 28, 1, 2>, 4);
            else go(k_qr1_step_multi<256, 2, 32, 1, 2>, 4);
            Acur = So; bcur = ro; Mcur = Ms;
        }
        LSQ_HIP(hipGetLastError());
        return qr2_factor_core(s, Acur, bcur, Mcur, n, R_out, rhs_out);
    }
    return qr2_factor_core(s, s->d_qr, s->d_qu, M, n, R_out, rhs_out);
}

static int qr2_workspace(lsq_solver *s, int M, int n) {
    lsq_ctx *c = s->ctx;
    Qr2Work *q = (Qr2Work *)s->qr2;
    if (!q || q->M != M || q->n != n) {
        qr2_free(q);
        q = new Qr2Work();
        q->M = M; q->n = n;
        const int ncolsB = Q2_NB + n + 1;
        const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
        q->kslices = std::max(1, std::min(64, (4 * c->num_cus + ntile - 1) / ntile));   // (4 workgroups per CU: their barriers and LDS phases interleave)
        LSQ_HIP(hipMalloc(&q->Vb, ((size_t)M * Q2_NB + 64) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->Wp, (size_t)q->kslices * ntile * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->W, (size_t)ntile * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->W2, (size_t)ntile * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->R, ((size_t)n * n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->rhs2, ((size_t)n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->tau1, ((size_t)n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->vn, (4 * (size_t)n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->ice, (2 * (size_t)n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->lazy, (2 * (size_t)n + 8) * sizeof(double)));
        const size_t smax = M > 64 * 32 * 256 ? 256 : M > 8 * 10 * 256 ? 64 : 8;
        const size_t xs = (size_t)64 * smax * 8 * 18 * 2 * sizeof(unsigned long long);
        LSQ_HIP(hipMalloc(&q->xslot, xs));
        LSQ_HIP(hipMalloc(&q->d_err, sizeof(int)));
        LSQ_HIP(hipMalloc(&q->Pn, ((size_t)M * Q2_NB + 32768) * sizeof(double)));
        LSQ_ZERO(q->xslot, 0, xs);
        LSQ_ZERO(q->d_err, 0, sizeof(int));
        LSQ_HIP(hipMalloc(&q->colat, (2 * (size_t)n + 8) * sizeof(int)));
        s->qr2 = q;
        s->qr2_free = qr2_free;
    }
    return LSQ_OK;
}

// the factorisation proper on [A | rhs] (A: M rows, column stride M); the workspace was sized for at least M rows
static int qr2_factor_core(lsq_solver *s, double *A, double *rhs, int M, int n, double **R_out, double **rhs_out) {
    lsq_ctx *c = s->ctx;
    Qr2Work *q = (Qr2Work *)s->qr2;
    // CholeskyQR2 panels (lsq_qr_cholqr.hip) need the error word to travel back with the certificate's copy
    const bool cq_ok = !q->no_cholqr && !getenv("LSQ_QR1_NO_CHOLQR") && !getenv("LSQ_QR_ALWAYS_PIVOT");
    q->cholqr_used = false;
    for (int c0 = 0; c0 < n; c0 += Q2_NB) {
        const int nb = std::min(Q2_NB, n - c0), cend = c0 + nb;
        if (cq_ok && nb == Q2_NB && M - c0 >= 256) {
            // panel: two Gram / Cholesky passes, no column-by-column chain; block reflector in basis-kernel form
            if (!q->cq.ready) LSQ_TRY(lsq_cqr_alloc(c, &q->cq, q->M));
            q->cholqr_used = true;
            const int rows = M - c0, ldv = rows;
            LSQ_TRY(lsq_cqr_panel(c, &q->cq, A, M, c0, q->Vb, ldv, q->d_err));
            const int ncols = n - cend + 1, ncolsB = Q2_NB + ncols;
            const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
            int ks = std::max(1, std::min(q->kslices, (rows + 4 * Q2_KC - 1) / (4 * Q2_KC)));
            hipLaunchKernelGGL(k_qr1_vtb, dim3(ntile * ks), dim3(256), 0, c->stream, q->Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks,
                               q->Wp);
            {
                long long tot = (long long)ntile * Q2_NB * Q2_NB;
                int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 4);
                hipLaunchKernelGGL(k_qr1_wreduce, dim3(g), dim3(256), 0, c->stream, q->Wp, ncolsB, ks, q->W);
            }
            LSQ_TRY(lsq_cqr_tw(c, &q->cq, q->W, ncolsB, A, M, c0, cend, n, rhs, q->Vb, ldv, q->W2));
            {
                const int nrt = (rows + Q2_NB - 1) / Q2_NB, nct = (ncols + Q2_NB - 1) / Q2_NB;
                hipLaunchKernelGGL(k_qr1_update, dim3(nrt * ((nct + Q2_UCT - 1) / Q2_UCT)), dim3(256), 0, c->stream, q->Vb, ldv, A, M,
                                   c0, cend, n, rhs, ncols, q->W2);
            }
            continue;
        }
        bool lazy = false;
        int side_k = 0;   // > 0: later pivot columns of a launch sit in the side panel
        // last panel: b rides through the steps as one more target column, so no block update is left to do
        const bool ride = cend == n && !getenv("LSQ_QR1_NO_RIDE");
        bool rode = false;
        auto steps = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(QR_NT), 0, c->stream, A, M, cend, c0, 1, q->tau1);
            for (int i = c0; i + 1 < cend; ++i)
                hipLaunchKernelGGL(kern, dim3(cend - i - 1), dim3(QR_NT), 0, c->stream, A, M, cend, i, 0, q->tau1);
        };
        auto steps_lazy = [&](auto kern) {
            for (int i = c0; i < cend; ++i)
                hipLaunchKernelGGL(kern, dim3(std::max(1, cend - i - 1)), dim3(QR_NT), 0, c->stream, A, M, cend, i, q->tau1,
                                   q->lazy, q->lazy + n);
            lazy = true;
        };
        const bool want_lazy = !getenv("LSQ_QR1_EAGER");
        const bool want_multi = want_lazy && !getenv("LSQ_QR1_SINGLE");
        auto steps_multi = [&](auto kern, int nt, int K, int S) {
            for (int i = c0; i < cend; i += K) {
                const int kk = std::min(K, cend - i), G = std::max(1, cend - i - kk + (ride ? 1 : 0));
                const int grid = S >= 64 ? G * S : S > 1 ? 8 * S * ((G + 7) / 8) : G;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 0, c->stream, A, M, cend, i, kk, q->tau1, q->lazy, q->lazy + n, G,
                                   q->xslot, ++q->epoch, q->d_err, q->Pn, c0, ride ? rhs : (double *)nullptr,
                                   (double *)nullptr, 0, (double *)nullptr);
            }
            lazy = true;
            side_k = K;
            rode = ride;
        };
        const int prow = M - c0;
        // slabs (S > 1) need every CU of an unpartitioned device; LSQ_QR1_COOP=0 keeps one workgroup per column
        const char *cv = getenv("LSQ_QR1_COOP");
        const int coop = !want_multi || c->num_cus < 256 || q->no_exchange ? 0 : cv ? atoi(cv) : 1;
        if (getenv("LSQ_QR1_LOOP")) steps(k_qr1_step);
        else if (coop && prow > 2 * 8 * 256 && prow <= 4 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 4>, 256, 4, 4);
        else if (coop && prow > 4 * 8 * 256 && prow <= 8 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 8>, 256, 4, 8);
        else if (coop && prow > 8 * 8 * 256 && prow <= 8 * 10 * 256) steps_multi(k_qr1_step_multi<256, 10, 4, 8>, 256, 4, 8);
        // tall operands: more slabs (the exchange of a round grows with S; there are few columns to pay it)
        else if (coop && prow > 8 * 10 * 256 && prow <= 8 * 16 * 256) steps_multi(k_qr1_step_multi<256, 16, 4, 8>, 256, 4, 8);
        else if (coop && prow > 8 * 16 * 256 && prow <= 32 * 16 * 256) steps_multi(k_qr1_step_multi<256, 16, 4, 32>, 256, 4, 32);
        else if (coop && prow > 32 * 16 * 256 && prow <= 64 * 32 * 256) steps_multi(k_qr1_step_multi<256, 32, 2, 64>, 256, 2, 64);
        else if (coop && prow > 64 * 32 * 256 && prow <= 256 * 32 * 256) steps_multi(k_qr1_step_multi<256, 32, 2, 256>, 256, 2, 256);
        else if (coop && prow > 8 * 256 && prow <= 2 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 2>, 256, 4, 2);
        else if (want_multi && prow <= 8 * 512) steps_multi(k_qr1_step_multi<512, 8, 4, 1>, 512, 4, 1);
        else if (want_multi && prow <= 8 * 1024) steps_multi(k_qr1_step_multi<512, 16, 4, 1>, 512, 4, 1);
        else if (want_multi && prow <= 32 * 512) steps_multi(k_qr1_step_multi<512, 32, 2, 1>, 512, 2, 1);
        else if (want_multi && prow <= 40 * 512) steps_multi(k_qr1_step_multi<512, 40, 2, 1>, 512, 2, 1);
        else if (want_lazy && M - c0 <= 8 * QR_NT) steps_lazy(k_qr1_step_lazy<8>);
        else if (want_lazy && M - c0 <= 16 * QR_NT) steps_lazy(k_qr1_step_lazy<16>);
        else if (want_lazy && M - c0 <= 20 * QR_NT) steps_lazy(k_qr1_step_lazy<20>);
        else if (M - c0 <= 8 * QR_NT) steps(k_qr1_step_reg<8>);          // the column fits the registers of one workgroup
        else if (M - c0 <= 16 * QR_NT) steps(k_qr1_step_reg<16>);
        else if (M - c0 <= 24 * QR_NT) steps(k_qr1_step_reg<24>);
        else steps(k_qr1_step);
        if (rode) {
            hipLaunchKernelGGL(k_qr1_fin, dim3(1), dim3(256), 0, c->stream, A, M, c0, nb, (const double *)q->lazy,
                               side_k ? (const double *)q->Pn : (const double *)nullptr, std::max(1, side_k));
            continue;
        }
        // block update of the trailing columns and of b
        const int ncols = n - cend + 1, ncolsB = Q2_NB + ncols;
        const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
        const int rows = M - c0, ldv = rows;
        {
            long long tot = (long long)rows * Q2_NB;
            int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 8);
            hipLaunchKernelGGL(k_qr1_vbuf, dim3(g), dim3(256), 0, c->stream, A, M, c0, nb, q->Vb, ldv,
                               lazy ? (const double *)q->lazy : (const double *)nullptr,
                               lazy ? (const double *)(q->lazy + n) : (const double *)nullptr,
                               side_k ? (const double *)q->Pn : (const double *)nullptr, std::max(1, side_k));
        }
        int ks = std::max(1, std::min(q->kslices, (rows + 4 * Q2_KC - 1) / (4 * Q2_KC)));
        hipLaunchKernelGGL(k_qr1_vtb, dim3(ntile * ks), dim3(256), 0, c->stream, q->Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks,
                           q->Wp);
        {
            long long tot = (long long)ntile * Q2_NB * Q2_NB;
            int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 4);
            hipLaunchKernelGGL(k_qr1_wreduce, dim3(g), dim3(256), 0, c->stream, q->Wp, ncolsB, ks, q->W);
        }
        if (getenv("LSQ_QR1_TW_SUBST"))
            hipLaunchKernelGGL(k_qr1_tw, dim3(std::max(1, lsq_div_up(ncols, 256))), dim3(256), 0, c->stream, q->W, ncolsB,
                               q->tau1, c0, nb, q->W2);
        else
            hipLaunchKernelGGL(k_qr1_tw_mfma, dim3(std::max(1, lsq_div_up(ncols, Q2_NB))), dim3(256), 0, c->stream, q->W, ncolsB,
                               q->tau1, c0, nb, q->W2);
        {
            const int nrt = (rows + Q2_NB - 1) / Q2_NB, nct = (ncols + Q2_NB - 1) / Q2_NB;
            hipLaunchKernelGGL(k_qr1_update, dim3(nrt * ((nct + Q2_UCT - 1) / Q2_UCT)), dim3(256), 0, c->stream, q->Vb, ldv, A, M, c0, cend, n, rhs, ncols,
                               q->W2);
        }
    }
    {
        long long tot = (long long)n * n;
        int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 8);
        hipLaunchKernelGGL(k_qr1_extract, dim3(g), dim3(256), 0, c->stream, A, M, n, rhs, q->R, q->rhs2);
    }
    LSQ_HIP(hipGetLastError());
    s->last_qr_panel = q->cholqr_used ? 2 : 1;
    *R_out = q->R;
    *rhs_out = q->rhs2;
    return LSQ_OK;
}

// true when ||R||_F ||inv(R)||_F certifies that xGELSY would keep all n columns (see k_tri_diaginv); X = inv(R)
// is left in q->Xinv.  One small device-to-host copy: the caller picks its launch sequence from the answer.
// The certified solve (pipelined back-substitution with the inverted diagonal blocks) is launched BEFORE the copy that
// carries the decision, so the same synchronisation also tells whether one of the in-kernel exchanges gave up
// (*timed_out: the caller repeats the solve without them; the result of this attempt is discarded).
static int qr2_certify_full_rank(lsq_solver *s, const double *R2, int n, double rcond, bool *certified, const double *rhs2,
                                 int *jp, double *d_x, bool *solved, bool *timed_out) {
    lsq_ctx *c = s->ctx;
    Qr2Work *q = (Qr2Work *)s->qr2;
    *certified = false;
    *solved = false;
    *timed_out = false;
    constexpr int FRO_BLOCKS = 256;
    if (!q->Xinv) {
        LSQ_HIP(hipMalloc(&q->Xinv, ((size_t)n * n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->T2, ((size_t)n * n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->fro, 2 * FRO_BLOCKS * sizeof(double)));
        LSQ_HIP(hipHostMalloc(&q->h_fro, (2 * FRO_BLOCKS + 1) * sizeof(double)));
    }
    hipLaunchKernelGGL(k_tri_diaginv, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, R2, n, q->Xinv, n, (size_t)64 * n + 64);
    for (long long sz = 64; sz < n; sz *= 2) {
        const int sb = (int)sz, npairs = (int)((n + 2 * sz - 1) / (2 * sz)), tps = sb / 64;
        const int grid = npairs * tps * tps;
        hipLaunchKernelGGL(k_tri_level, dim3(grid), dim3(256), 0, c->stream, R2, q->Xinv, q->T2, n, sb, 0);
        hipLaunchKernelGGL(k_tri_level, dim3(grid), dim3(256), 0, c->stream, R2, q->Xinv, q->T2, n, sb, 1);
    }
    hipLaunchKernelGGL(k_tri_fro, dim3(FRO_BLOCKS), dim3(256), 0, c->stream, R2, q->Xinv, n, q->fro);
    if (lsq_div_up(n, 64) <= 256 && !q->no_exchange && !getenv("LSQ_QR_SUBST_SOLVE")) {   // speculative: used if certified
        if (!q->bslot) {
            LSQ_HIP(hipMalloc(&q->bslot, (size_t)256 * 64 * 2 * sizeof(unsigned long long)));
            LSQ_ZERO(q->bslot, 0, (size_t)256 * 64 * 2 * sizeof(unsigned long long));
        }
        hipLaunchKernelGGL(k_tri_identity, dim3(lsq_div_up(n, 256)), dim3(256), 0, c->stream, jp, n, s->d_info);
        hipLaunchKernelGGL(k_tri_bsolve, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, R2, q->Xinv, n, (size_t)64 * n + 64, n,
                           rhs2, d_x, q->bslot, ++q->epoch, q->d_err);
        *solved = true;
    }
    if (!q->no_exchange && getenv("LSQ_TEST_EXCHANGE_TIMEOUT")) {   // test hook: pretend a wait gave up
        static const int one = 1;
        LSQ_HIP(hipMemcpyAsync(q->d_err, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    LSQ_HIP(hipMemcpyAsync(q->h_fro, q->fro, 2 * FRO_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LSQ_HIP(hipMemcpyAsync(q->h_fro + 2 * FRO_BLOCKS, q->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    if (const int ev = *(const int *)(q->h_fro + 2 * FRO_BLOCKS)) {
        // bit 0: a bounded wait of an in-kernel exchange gave up (workgroups not dispatched in index order, or the device
        // shared with other work) -- this solver stops using them;  bit 1: a CholeskyQR2 panel broke down (ill-conditioned
        // or rank-deficient panel) -- this solver goes back to the column-by-column panel.  Either way: once more.
        LSQ_ZERO(q->d_err, 0, sizeof(int));
        if (ev & 1) q->no_exchange = true;
        if (ev & 2) q->no_cholqr = true;
        *timed_out = true;
        return LSQ_OK;
    }
    double fr = 0.0, fx = 0.0;
    for (int b = 0; b < FRO_BLOCKS; ++b) { fr += q->h_fro[2 * b]; fx += q->h_fro[2 * b + 1]; }
    const double bound = sqrt(fr) * sqrt(fx);       // >= cond_2(R); NaN/Inf (singular or overflowing R) fail the test
    // safety factor 16: rounding in the computed inverse and in dlaic1's own estimates
    *certified = std::isfinite(bound) && bound * rcond * 16.0 <= 1.0;
    return LSQ_OK;
}

// U'U x = b for the blocked Cholesky (dense_cholesky.jl:56-57): inverted diagonal blocks, then the two pipelined
// block solves; b is overwritten by x.  Returns LSQ_EARG when the scheme does not apply (caller falls back).
struct TriPipe {
    double *Xd = nullptr, *z = nullptr;          // [nblk][64][64] inverted diagonal blocks; intermediate z
    unsigned long long *slot_f = nullptr, *slot_b = nullptr;
    unsigned long long epoch = 0;
    int *d_err = nullptr;
    int n = 0;
};
static void tripipe_free(void *p) {
    TriPipe *t = (TriPipe *)p;
    if (!t) return;
    hipFree(t->Xd); hipFree(t->z); hipFree(t->slot_f); hipFree(t->slot_b); hipFree(t->d_err);
    delete t;
}
static int tri_chol_pipe(lsq_solver *s, int n, TriPipe **out) {
    const int nblk = lsq_div_up(n, 64);
    *out = nullptr;
    if (nblk > 256 || s->pipe_off || getenv("LSQ_CHOL_SUBST_SOLVE")) return LSQ_EARG;
    TriPipe *t = (TriPipe *)s->tripipe;
    if (!t || t->n != n) {
        if (t) tripipe_free(t);
        t = new TriPipe();
        t->n = n;
        const size_t sl = (size_t)nblk * 64 * 2 * sizeof(unsigned long long);
        LSQ_HIP(hipMalloc(&t->Xd, (size_t)nblk * 4096 * sizeof(double)));
        LSQ_HIP(hipMalloc(&t->z, ((size_t)n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&t->slot_f, sl));
        LSQ_HIP(hipMalloc(&t->slot_b, sl));
        LSQ_HIP(hipMalloc(&t->d_err, sizeof(int)));
        LSQ_ZERO(t->slot_f, 0, sl);
        LSQ_ZERO(t->slot_b, 0, sl);
        LSQ_ZERO(t->d_err, 0, sizeof(int));
        LSQ_ZERO(t->Xd, 0, (size_t)nblk * 4096 * sizeof(double));
        s->tripipe = t;
        s->tripipe_free = tripipe_free;
    }
    *out = t;
    return LSQ_OK;
}
double *lsq_tri_chol_diagbuf(lsq_solver *s, int n) {
    TriPipe *t = nullptr;
    return tri_chol_pipe(s, n, &t) == LSQ_OK ? t->Xd : nullptr;
}
int lsq_tri_chol_solve(lsq_solver *s, const double *U, int n, double *d_bx) {
    lsq_ctx *c = s->ctx;
    const int nblk = lsq_div_up(n, 64);
    TriPipe *t = nullptr;
    if (tri_chol_pipe(s, n, &t) != LSQ_OK) return LSQ_EARG;
    ++t->epoch;
    if (!s->chol_have_diaginv)     // (the MFMA panel kernel of the blocked factorisation has already left inv(U_kk) in Xd)
        hipLaunchKernelGGL(k_tri_diaginv, dim3(nblk), dim3(256), 0, c->stream, U, n, t->Xd, 64, (size_t)4096);
    hipLaunchKernelGGL(k_tri_fsolve_t, dim3(nblk), dim3(256), 0, c->stream, U, t->Xd, 64, (size_t)4096, n, d_bx, t->z, t->slot_f,
                       t->epoch, t->d_err);
    hipLaunchKernelGGL(k_tri_bsolve, dim3(nblk), dim3(256), 0, c->stream, U, t->Xd, 64, (size_t)4096, n, t->z, d_bx, t->slot_b,
                       t->epoch, t->d_err);
    if (getenv("LSQ_TEST_EXCHANGE_TIMEOUT")) {   // test hook: pretend a wait gave up (and spoil the result it would have spoilt)
        static const int one = 1;
        LSQ_HIP(hipMemcpyAsync(t->d_err, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
        LSQ_HIP(hipMemsetAsync(d_bx, 0xff, (size_t)n * sizeof(double), c->stream));
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// did a wait of the pipelined solves give up?  lsq_tri_pipe_err_copy enqueues the copy of the flag next to the caller's own
// status copy (one synchronisation for both); lsq_tri_pipe_disable acts on it: the solver stops using the pipelined
// solves and the caller repeats them with the single-workgroup kernel.
void lsq_tri_pipe_err_copy(lsq_solver *s, int *h_dst) {
    TriPipe *t = (TriPipe *)s->tripipe;
    *h_dst = 0;
    if (t && !s->pipe_off) (void)hipMemcpyAsync(h_dst, t->d_err, sizeof(int), hipMemcpyDeviceToHost, s->ctx->stream);
}
void lsq_tri_pipe_disable(lsq_solver *s) {
    TriPipe *t = (TriPipe *)s->tripipe;
    if (t) (void)hipMemsetAsync(t->d_err, 0, sizeof(int), s->ctx->stream);
    s->pipe_off = 1;
}

// sum of squares of inv(U) for the n x n upper triangle U (explicit inverse: k_tri_diaginv + k_tri_level levels);
// synchronises the stream.  NaN / Inf when U is singular.
int lsq_tri_inv_fro2(lsq_solver *s, const double *U, int n, double *fro2_inv) {
    lsq_ctx *c = s->ctx;
    constexpr int FRO_BLOCKS = 256;
    if (!s->tri_X) {
        LSQ_HIP(hipMalloc(&s->tri_X, ((size_t)n * n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->tri_T, ((size_t)n * n + 8) * sizeof(double)));
        LSQ_HIP(hipMalloc(&s->tri_fro, 2 * FRO_BLOCKS * sizeof(double)));
        LSQ_HIP(hipHostMalloc(&s->tri_hfro, 2 * FRO_BLOCKS * sizeof(double)));
    }
    hipLaunchKernelGGL(k_tri_diaginv, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, U, n, s->tri_X, n, (size_t)64 * n + 64);
    for (long long sz = 64; sz < n; sz *= 2) {
        const int sb = (int)sz, npairs = (int)((n + 2 * sz - 1) / (2 * sz)), tps = sb / 64;
        const int grid = npairs * tps * tps;
        hipLaunchKernelGGL(k_tri_level, dim3(grid), dim3(256), 0, c->stream, U, s->tri_X, s->tri_T, n, sb, 0);
        hipLaunchKernelGGL(k_tri_level, dim3(grid), dim3(256), 0, c->stream, U, s->tri_X, s->tri_T, n, sb, 1);
    }
    hipLaunchKernelGGL(k_tri_fro, dim3(FRO_BLOCKS), dim3(256), 0, c->stream, U, s->tri_X, n, s->tri_fro);
    LSQ_HIP(hipMemcpyAsync(s->tri_hfro, s->tri_fro, 2 * FRO_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    double fx = 0.0;
    for (int b = 0; b < FRO_BLOCKS; ++b) fx += s->tri_hfro[2 * b + 1];
    *fro2_inv = fx;
    return LSQ_OK;
}

// dense_qr.jl:30-42 (d_damp == nullptr) and :56-88
int lsq_qr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n;
    if (J->kind != LSQ_MAT_DENSE) {
        lsq_set_error("solver QR() is not available for sparse Jacobians. Choose between Cholesky() and LSMR()");
        return LSQ_EARG;  // types.jl:115-117
    }
    if (n != s->n || m != s->m || (d_damp != nullptr) != (s->for_lm != 0)) {
        lsq_set_error("qr: solver/Jacobian mismatch (length(u) should equal length(x) + length(y))");
        return LSQ_EDIM;
    }
    const int M = d_damp ? m + n : m;
    const int lu = d_damp ? M : std::max(m, n);
    if (n > 0 && M > 0)
    for (int attempt = 0; attempt < 3; ++attempt) {     // (again only after an in-kernel exchange timed out / a CholeskyQR2 panel broke down)
        long long tot = (long long)M * n;
        int grid = (int)std::min<long long>((tot + LSQ_NT - 1) / LSQ_NT, (long long)c->num_cus * 16);
        hipLaunchKernelGGL(k_stack, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, m, n, d_damp, s->d_qr);
        hipLaunchKernelGGL(k_rhs, dim3(lsq_div_up(lu, LSQ_NT)), dim3(LSQ_NT), 0, c->stream, d_y, m, lu, s->d_qu);
        const int mn = std::min(M, n);
        s->last_qr_path = 1;
        if (qr2_applies(M, n)) {
            double *R2 = nullptr, *rhs2 = nullptr;
            LSQ_TRY(qr2_factor(s, M, n, &R2, &rhs2));
            // stage 2: the pivoted sweep on the n x n triangle, Q1'b riding along
            double *ws = s->d_work;
            double *tau = ws + 2 * n;
            int *jp = (int *)s->d_tau;
            Qr2Work *q = (Qr2Work *)s->qr2;
            bool have_rank = false;
            bool full_rank = false, solved = false, timed_out = false;
            if (!getenv("LSQ_QR_ALWAYS_PIVOT"))
                LSQ_TRY(qr2_certify_full_rank(s, R2, n, (double)mn * DBL_EPSILON, &full_rank, rhs2, jp, d_x, &solved, &timed_out));
            if (timed_out) {
                if (attempt < 2) continue;       // once more from the stacked operand, without what gave up
                lsq_set_error("qr: the fast paths gave up three times");
                return LSQ_EHIP;
            }
            if (full_rank) {
                // rank = n is certain: the unpivoted triangle gives the same (unique) solution, jp = identity
                if (!solved) {
                    hipLaunchKernelGGL(k_tri_identity, dim3(lsq_div_up(n, 256)), dim3(256), 0, c->stream, jp, n, s->d_info);
                    if (n <= QRK_MAXN)
                        hipLaunchKernelGGL(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, jp, s->d_info, d_x);
                    else
                        hipLaunchKernelGGL(k_tri_matvec, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, q->Xinv, n, rhs2, d_x);
                }
                LSQ_HIP(hipGetLastError());
                s->last_rank = -1;
                s->last_qr_path = 3;
                if (nmul) *nmul = 1;
                return LSQ_OK;
            }
            s->last_qr_path = 2;
            if (n <= Q2S_NT * Q2S_RPT && !getenv("LSQ_QR2_TWO_LAUNCH")) {
                // one launch per column, lazy exchanges (k_qr2_step)
                double *vn1[2] = {q->vn, q->vn + 2 * n}, *vn2[2] = {q->vn + n, q->vn + 3 * n};
                int *colat[2] = {q->colat, q->colat + n};
                hipLaunchKernelGGL(k_qr2_init, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, R2, n, vn1[0], vn2[0], colat[0]);
                for (int i = 0; i < n; ++i) {
                    const int a = i & 1, b = a ^ 1;
                    auto go = [&](auto kern, int cpb) {
                        hipLaunchKernelGGL(kern, dim3((n - i + cpb - 1) / cpb + 1), dim3(Q2S_NT), 0, c->stream, R2, n, i, rhs2,
                                           vn1[a], vn2[a], vn1[b], vn2[b], colat[a], colat[b], tau, ws + 7 * n, q->ice,
                                           (double)mn * DBL_EPSILON, s->d_info);
                    };
                    if (n - i >= 768) go(k_qr2_step<4>, 4);
                    else if (n - i >= 320) go(k_qr2_step<2>, 2);
                    else go(k_qr2_step<1>, 1);
                }
                // the solve wants R in pivoted order: gather it into the (now free) factor buffer
                long long tot = (long long)n * n;
                int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 8);
                hipLaunchKernelGGL(k_qr2_gather, dim3(g), dim3(256), 0, c->stream, R2, n, colat[n & 1], ws + 7 * n, s->d_qr, jp);
                R2 = s->d_qr;
                have_rank = true;
            } else {
                double *vn1 = ws, *vn2 = ws + n;
                hipLaunchKernelGGL(k_qr_norms, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, R2, n, n, vn1, vn2, jp);
                for (int i = 0; i < n; ++i) {
                    hipLaunchKernelGGL(k_qr_pivot, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, i, vn1, vn2, jp, tau);
                    hipLaunchKernelGGL(k_qr_apply, dim3(lsq_div_up(n - i, 4)), dim3(256), 0, c->stream, R2, n, n, i, rhs2, tau,
                                       vn1, vn2);
                }
            }
            int ph = 0;
            if (n <= QRK_MAXN && !getenv("LSQ_QR_SLOW_SOLVE")) {
                if (!have_rank)
                    hipLaunchKernelGGL(k_qr_rank, dim3(1), dim3(64), 0, c->stream, R2, n, n, (double)mn * DBL_EPSILON, s->d_info);
                hipLaunchKernelGGL(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, jp, s->d_info, d_x);
                ph = 4;
            }
            hipLaunchKernelGGL(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, n, d_x, s->d_work, jp,
                               s->d_T, (double)mn * DBL_EPSILON, s->d_info, ph);
        } else if (n >= 64 && (long long)M * n >= 65536 && !getenv("LSQ_QR_SMALL")) {
            // multi-CU column-pivoted Householder (BLAS-2 per column, the rhs rides along as column n)
            double *ws = s->d_work;
            double *vn1 = ws, *vn2 = ws + n, *tau = ws + 2 * n;
            int *jp = (int *)s->d_tau;
            hipLaunchKernelGGL(k_qr_norms, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, s->d_qr, M, n, vn1, vn2, jp);
            for (int i = 0; i < mn; ++i) {
                hipLaunchKernelGGL(k_qr_pivot, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, i, vn1, vn2, jp, tau);
                hipLaunchKernelGGL(k_qr_apply, dim3(lsq_div_up(n - i, 4)), dim3(256), 0, c->stream, s->d_qr, M, n, i,
                                   s->d_qu, tau, vn1, vn2);
            }
            int ph = 0;
            if (n <= QRK_MAXN && M >= n && !getenv("LSQ_QR_SLOW_SOLVE")) {
                hipLaunchKernelGGL(k_qr_rank, dim3(1), dim3(64), 0, c->stream, s->d_qr, M, mn, (double)mn * DBL_EPSILON, s->d_info);
                hipLaunchKernelGGL(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, jp, s->d_info,
                                   d_x);
                ph = 4;
            }
            hipLaunchKernelGGL(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, lu, d_x,
                               s->d_work, jp, s->d_T, (double)mn * DBL_EPSILON, s->d_info, ph);
        } else {
            hipLaunchKernelGGL(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, lu, d_x,
                               s->d_work, (int *)s->d_tau, s->d_T, (double)mn * DBL_EPSILON, s->d_info, 3);
        }
        LSQ_HIP(hipGetLastError());
        break;
    }
    s->last_rank = -1;  // fetched lazily by lsq_solver_info
    if (nmul) *nmul = 1;
    return LSQ_OK;
}
