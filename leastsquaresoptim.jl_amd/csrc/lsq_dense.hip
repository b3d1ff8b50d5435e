// Dense Jacobian path, part 1: GEMVs / colsumabs2 and the normal-equation + Cholesky solver (dense_cholesky.jl:29-59;
// the blocked MFMA factorisation it drives lives in lsq_dense_mfma.hip).  The QR solver (dense_qr.jl:30-88) and the
// triangular-solve / triangular-inverse kernels both factorisations share are in lsq_qr.hip.
// Everything here runs on the device; the host only reads back status words.
#include <type_traits>
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "lsq_solver.h"
#include "lsq_spmv.h"

// allow_tiles: the one-launch factorisation (k_chol_tiles) may be used; the caller then handles info == -1 (a bounded wait gave up)
int lsq_cholesky_blocked(lsq_solver *s, lsq_mat *J, const double *d_damp, double *d_x, double *d_dmax, bool allow_tiles = false,
                         const double *d_y = nullptr);
int lsq_cholesky_blocked_solve(lsq_solver *s, int n, double *d_x);
int lsq_tri_inv_fro2(lsq_solver *s, const double *U, int n, double *fro2_inv);   // lsq_qr.hip
void lsq_tri_pipe_err_copy(lsq_solver *s, int *h_dst);                           // lsq_qr.hip
void lsq_tri_pipe_disable(lsq_solver *s);                                        // lsq_qr.hip
const int *lsq_tri_pipe_err_ptr(lsq_solver *s);                                  // lsq_qr.hip (null: the pipeline is off)


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

int lsq_dense_part(lsq_mat *J, int nwin) { return lsq_dense_part_elems(J, (size_t)nwin * J->n); }
int lsq_dense_part_elems(lsq_mat *J, size_t need) {
    if ((size_t)J->dpart_cap < need) {
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
        LSQ_LAUNCH((k_dense_t_win<true>), dim3(nwin * J->n), dim3(LSQ_NT), 0, J->ctx->stream, J->d_dense, J->m, J->n,
                           (const double *)nullptr, wrows, J->d_dpart, (const int *)nullptr);
        const int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
        LSQ_LAUNCH((k_combine<EpiStoreD>), dim3(nb), dim3(LSQ_NT), 0, J->ctx->stream, J->d_dpart, J->n, nwin, e, nb);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    int grid = J->n > LSQ_MAX_GRID ? LSQ_MAX_GRID : J->n;
    LSQ_LAUNCH((k_dense_t<EpiStoreD, true>), dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream,
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
    hipFree(s->d_qr); hipFree(s->d_qu); hipFree(s->d_T); hipFree(s->d_chol_flags);
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
    if (lsq_cholesky_blocked(s, J, nullptr, nullptr, s->d_work, true) != LSQ_OK) return fail(LSQ_EHIP);
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
    if (info == -1 && !s->fb_tiles.off()) {       // the one-launch factorisation gave up on a wait: panel launches for a while
        s->fb_tiles.gave_up(s->ctx, LSQ_FB_CHOL_TILES);
        return chol_certified(s, J, d_y, d_x, rc);
    }
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
    int rc_cert = LSQ_OK;
    // blocked path from n = 32, or earlier when the rows make the SYRK the whole cost (tall and thin: 10^6 x 20 takes
    // 1.8 ms blocked, 90 ms with the one-workgroup kernels; 300 x 8 0.15 vs 0.09)
    const bool blocked = n >= 32 || (n >= 2 && (long long)m * n >= 20000);
    if (blocked && d_damp) {
        // MFMA SYRK + blocked Cholesky + pipelined solves (lsq_dense_mfma.hip); measured crossover against the
        // single-workgroup kernel: 200 x 16 0.15 vs 0.11 ms, 300 x 32 0.15 vs 0.18, 500 x 64 0.16 vs 0.32, 2000 x 127 0.24 vs 0.93
        // mul!(x, J', y) rides in the SYRK launch of lsq_cholesky_blocked (d_y), or is launched by it
        // the two status words (PosDefException position / "a wait gave up", and the solve pipeline's flag) reach the host
        // through the pinned slot mirror and a spin, as the loops' scalars do: a hipStreamSynchronize wake-up here cost
        // ~50 us of every 0.35 ms solve; the backward solve's last block stores them itself (pub_want)
        int st4[4] = {0, 0, 0, 0};
        auto factor_and_solve = [&](bool tiles) {
            s->pub_want = true;
            s->pub_seq = 0;
            const int rc = lsq_cholesky_blocked(s, J, d_damp, d_x, nullptr, tiles, d_y);
            s->pub_want = false;     // (whatever happened: no later solve of this solver may publish on this call's behalf)
            if (rc != LSQ_OK) return rc;
            if (s->pub_seq) return lsq_wait_ints(c, s->pub_seq, s->d_info, lsq_tri_pipe_err_ptr(s), nullptr, nullptr, st4);
            return lsq_read_ints(c, s->d_info, lsq_tri_pipe_err_ptr(s), nullptr, nullptr, st4);
        };
        LSQ_TRY(factor_and_solve(true));
        s->last_chol_path = s->last_chol_tiles ? 4 : 2;
        int info = st4[0], perr = st4[1];
        if (info == -1) {                         // k_chol_tiles gave up on a wait: panel launches for a while, and redo
            s->fb_tiles.gave_up(c, LSQ_FB_CHOL_TILES);
            s->last_chol_path = 2;
            LSQ_TRY(factor_and_solve(false));
            info = st4[0];
            perr = st4[1];
        }
        if (info != 0) {
            lsq_set_error("PosDefException: matrix is not positive definite; Cholesky failed at %d", info);
            return LSQ_ENOTPD;
        }
        if (perr) {                               // (the factor is intact: only the two solves are repeated)
            lsq_tri_pipe_disable(s);
            LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));
            LSQ_TRY(lsq_cholesky_blocked_solve(s, n, d_x));
        }
    } else if (blocked && !d_damp &&
               chol_certified(s, J, d_y, d_x, &rc_cert)) {
        // Dogleg (dense_cholesky.jl:29-35): the unpivoted blocked factorisation gave the solution and the
        // certificate proved that cholesky!(.., Val(true)) would not have stopped early (see chol_certified)
        if (rc_cert != LSQ_OK) return rc_cert;
    } else if (n > 0) {
        if (rc_cert != LSQ_OK) return rc_cert;
        s->last_chol_path = 1;
        const int nt = (n + SY_T - 1) / SY_T;
        LSQ_LAUNCH(k_syrk_upper, dim3(nt * (nt + 1) / 2), dim3(256), 0, c->stream, J->d_dense, m, n,
                           s->d_chol, d_damp);
        LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));  // mul!(x, J', y)
        int *piv = (int *)s->d_tau;
        if (d_damp)
            LSQ_LAUNCH((k_chol_solve<false>), dim3(1), dim3(CH_NT), 0, c->stream, s->d_chol, n, d_x,
                               s->d_info, piv, s->d_work, s->d_work + 2 * n);
        else
            LSQ_LAUNCH((k_chol_solve<true>), dim3(1), dim3(CH_NT), 0, c->stream, s->d_chol, n, d_x,
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
    // (fast paths that gave up on a bounded wait are armed again after a pause: LsqFallback)
    s->fb_tiles.solve_done(s->last_chol_path == 4);
    s->fb_pipe.solve_done(s->last_chol_path >= 2 && s->tripipe != nullptr);
    if (nmul) *nmul = 1;
    return LSQ_OK;
}
