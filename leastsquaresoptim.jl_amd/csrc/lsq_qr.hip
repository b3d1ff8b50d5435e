// Dense Jacobian path, part 2: the QR solver (dense_qr.jl:30-88; LinearAlgebra.ldiv!(::QRPivoted, b) [stdlib] = xGELSY):
// the one-workgroup column-pivoted Householder kernel for tiny operands, the two-launch-per-column pivoted sweep, and the
// two-stage factorisation (unpivoted blocked Householder QR with CholeskyQR2 panels, lsq_qr_cholqr.hip -> full-rank
// certificate or pivoted sweep on R).  Also here, because both factorisations use them: the triangular inverse
// (k_tri_diaginv / k_tri_level) and the pipelined triangular solves (k_tri_fsolve_t / k_tri_bsolve) with their driver
// for the blocked Cholesky (lsq_tri_chol_solve).
// Everything runs on the device; the host only reads back status words.
#include <type_traits>
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "lsq_qr_work.h"
#include "lsq_spmv.h"

// ---------------------------------------------------------------------------------------------
// single-workgroup column-pivoted Householder QR (dgeqp3 semantics via the dlaqp2 recurrence)
// followed by the xGELSY solve.  A is M x n (lda = M), b has length >= max(M, n).
// ---------------------------------------------------------------------------------------------


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
             unsigned long long *__restrict__ slot /* [nblk][64][2] */, unsigned long long epoch, int *__restrict__ err,
             const int *pub_info = nullptr, double *pub_dst = nullptr, unsigned long long *pub_seq_word = nullptr,
             unsigned long long pub_seq = 0) {
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
                if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); break; }
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
        // block 0 is the last to finish (it has waited for every other block, and cannot finish cleanly if one of them gave
        // up): it hands the solve's status words -- the factorisation's `info` and the pipeline's flag -- to the host itself,
        // through the pinned mirror (k_publish_ints: one launch less at the end of every Cholesky solve)
        if (pub_dst && t == 0 && tid == 0) {
            // (a workgroup whose wait gave up raised the flag, fenced, and only then published the slots this workgroup has
            //  consumed: with the acquire below the flag read here cannot be older than those slots)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(pub_dst + 0, pub_info ? (double)*pub_info : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub_dst + 1, (double)__hip_atomic_load(err, RLX_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub_dst + 2, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub_dst + 3, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub_seq_word, pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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
                if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); break; }
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
    LSQ_LAUNCH(k_tri_diaginv, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, R2, n, q->Xinv, n, (size_t)64 * n + 64);
    for (long long sz = 64; sz < n; sz *= 2) {
        const int sb = (int)sz, npairs = (int)((n + 2 * sz - 1) / (2 * sz)), tps = sb / 64;
        const int grid = npairs * tps * tps;
        LSQ_LAUNCH(k_tri_level, dim3(grid), dim3(256), 0, c->stream, R2, q->Xinv, q->T2, n, sb, 0);
        LSQ_LAUNCH(k_tri_level, dim3(grid), dim3(256), 0, c->stream, R2, q->Xinv, q->T2, n, sb, 1);
    }
    LSQ_LAUNCH(k_tri_fro, dim3(FRO_BLOCKS), dim3(256), 0, c->stream, R2, q->Xinv, n, q->fro);
    if (lsq_div_up(n, 64) <= 256 && !q->no_exchange) {   // speculative: used if certified
        if (!q->bslot) {
            LSQ_HIP(hipMalloc(&q->bslot, (size_t)256 * 64 * 2 * sizeof(unsigned long long)));
            LSQ_ZERO(q->bslot, 0, (size_t)256 * 64 * 2 * sizeof(unsigned long long));
        }
        LSQ_LAUNCH(k_tri_identity, dim3(lsq_div_up(n, 256)), dim3(256), 0, c->stream, jp, n, s->d_info);
        LSQ_LAUNCH(k_tri_bsolve, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, R2, q->Xinv, n, (size_t)64 * n + 64, n,
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
        if (ev & 1) { q->no_exchange = true; s->fb_qrx.gave_up(c, LSQ_FB_QR_EXCHANGE); }
        if (ev & 2) { q->no_cholqr = true; s->fb_cholqr.gave_up(c, LSQ_FB_CHOLQR); }
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
    bool fwd_fused = false;                      // the forward half of the coming solve was run by the factorisation kernel
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
    if (nblk > 256 || s->fb_pipe.off()) return LSQ_EARG;
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
// k_chol_chain runs U'z = b itself, block by block behind the factorisation (lsq_dense_mfma.hip): the operands of
// k_tri_fsolve_t for the coming solve, whose epoch starts here; lsq_tri_chol_solve then launches the backward half only
int lsq_tri_chol_fwd_operands(lsq_solver *s, int n, double **z, unsigned long long **slot, unsigned long long *epoch, int **err) {
    TriPipe *t = nullptr;
    if (tri_chol_pipe(s, n, &t) != LSQ_OK) return LSQ_EARG;
    ++t->epoch;
    t->fwd_fused = true;
    *z = t->z;
    *slot = t->slot_f;
    *epoch = t->epoch;
    *err = t->d_err;
    return LSQ_OK;
}
int lsq_tri_chol_solve(lsq_solver *s, const double *U, int n, double *d_bx) {
    lsq_ctx *c = s->ctx;
    const int nblk = lsq_div_up(n, 64);
    TriPipe *t = nullptr;
    if (tri_chol_pipe(s, n, &t) != LSQ_OK) return LSQ_EARG;
    if (t->fwd_fused) {
        t->fwd_fused = false;                    // z is in t->z already
    } else {
        ++t->epoch;
        if (!s->chol_have_diaginv)     // (the MFMA panel kernel of the blocked factorisation has already left inv(U_kk) in Xd)
            LSQ_LAUNCH(k_tri_diaginv, dim3(nblk), dim3(256), 0, c->stream, U, n, t->Xd, 64, (size_t)4096);
        LSQ_LAUNCH(k_tri_fsolve_t, dim3(nblk), dim3(256), 0, c->stream, U, t->Xd, 64, (size_t)4096, n, d_bx, t->z, t->slot_f,
                           t->epoch, t->d_err);
    }
    // the caller wants the status words on the host after this solve: the last block publishes them (lsq_cholesky_solve)
    const bool hook = getenv("LSQ_TEST_EXCHANGE_TIMEOUT") != nullptr;
    LsqSlotPublish pub;
    if (s->pub_want && !hook) {
        pub = lsq_ints_ticket(c);
        s->pub_seq = pub.seq;
    }
    s->pub_want = false;
    LSQ_LAUNCH(k_tri_bsolve, dim3(nblk), dim3(256), 0, c->stream, U, t->Xd, 64, (size_t)4096, n, t->z, d_bx, t->slot_b,
                       t->epoch, t->d_err, (const int *)s->d_info, pub.dst, pub.seq_word, pub.seq);
    if (hook) {   // test hook: pretend a wait gave up (and spoil the result it would have spoilt)
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
    if (t && !s->fb_pipe.off()) (void)hipMemcpyAsync(h_dst, t->d_err, sizeof(int), hipMemcpyDeviceToHost, s->ctx->stream);
}
const int *lsq_tri_pipe_err_ptr(lsq_solver *s) {
    TriPipe *t = (TriPipe *)s->tripipe;
    return t && !s->fb_pipe.off() ? t->d_err : nullptr;
}
void lsq_tri_pipe_disable(lsq_solver *s) {
    TriPipe *t = (TriPipe *)s->tripipe;
    if (t) (void)hipMemsetAsync(t->d_err, 0, sizeof(int), s->ctx->stream);
    s->fb_pipe.gave_up(s->ctx, LSQ_FB_TRI_PIPE);
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
    LSQ_LAUNCH(k_tri_diaginv, dim3(lsq_div_up(n, 64)), dim3(256), 0, c->stream, U, n, s->tri_X, n, (size_t)64 * n + 64);
    for (long long sz = 64; sz < n; sz *= 2) {
        const int sb = (int)sz, npairs = (int)((n + 2 * sz - 1) / (2 * sz)), tps = sb / 64;
        const int grid = npairs * tps * tps;
        LSQ_LAUNCH(k_tri_level, dim3(grid), dim3(256), 0, c->stream, U, s->tri_X, s->tri_T, n, sb, 0);
        LSQ_LAUNCH(k_tri_level, dim3(grid), dim3(256), 0, c->stream, U, s->tri_X, s->tri_T, n, sb, 1);
    }
    LSQ_LAUNCH(k_tri_fro, dim3(FRO_BLOCKS), dim3(256), 0, c->stream, U, s->tri_X, n, s->tri_fro);
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
        if (s->qr2) {   // the paths that gave up are armed again after a pause (LsqFallback)
            ((Qr2Work *)s->qr2)->no_exchange = s->fb_qrx.off();
            ((Qr2Work *)s->qr2)->no_cholqr = s->fb_cholqr.off();
        }
        long long tot = (long long)M * n;
        int grid = (int)std::min<long long>((tot + LSQ_NT - 1) / LSQ_NT, (long long)c->num_cus * 16);
        LSQ_LAUNCH(k_stack, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, m, n, d_damp, s->d_qr);
        LSQ_LAUNCH(k_rhs, dim3(lsq_div_up(lu, LSQ_NT)), dim3(LSQ_NT), 0, c->stream, d_y, m, lu, s->d_qu);
        const int mn = std::min(M, n);
        s->last_qr_path = 1;
        if (lsq_qr2_applies(M, n)) {
            double *R2 = nullptr, *rhs2 = nullptr;
            LSQ_TRY(lsq_qr2_factor(s, M, n, &R2, &rhs2));
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
                    LSQ_LAUNCH(k_tri_identity, dim3(lsq_div_up(n, 256)), dim3(256), 0, c->stream, jp, n, s->d_info);
                    if (n <= QRK_MAXN)
                        LSQ_LAUNCH(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, jp, s->d_info, d_x);
                    else
                        LSQ_LAUNCH(k_tri_matvec, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, q->Xinv, n, rhs2, d_x);
                }
                LSQ_HIP(hipGetLastError());
                s->last_rank = -1;
                s->last_qr_path = 3;
                s->fb_qrx.solve_done(!q->no_exchange);
                s->fb_cholqr.solve_done(q->cholqr_used);
                if (nmul) *nmul = 1;
                return LSQ_OK;
            }
            s->last_qr_path = 2;
            if (n <= Q2S_NT * Q2S_RPT) {
                // one launch per column, lazy exchanges (k_qr2_step)
                double *vn1[2] = {q->vn, q->vn + 2 * n}, *vn2[2] = {q->vn + n, q->vn + 3 * n};
                int *colat[2] = {q->colat, q->colat + n};
                LSQ_LAUNCH(k_qr2_init, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, R2, n, vn1[0], vn2[0], colat[0]);
                for (int i = 0; i < n; ++i) {
                    const int a = i & 1, b = a ^ 1;
                    auto go = [&](auto kern, int cpb) {
                        LSQ_LAUNCH(kern, dim3((n - i + cpb - 1) / cpb + 1), dim3(Q2S_NT), 0, c->stream, R2, n, i, rhs2,
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
                LSQ_LAUNCH(k_qr2_gather, dim3(g), dim3(256), 0, c->stream, R2, n, colat[n & 1], ws + 7 * n, s->d_qr, jp);
                R2 = s->d_qr;
                have_rank = true;
            } else {
                double *vn1 = ws, *vn2 = ws + n;
                LSQ_LAUNCH(k_qr_norms, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, R2, n, n, vn1, vn2, jp);
                for (int i = 0; i < n; ++i) {
                    LSQ_LAUNCH(k_qr_pivot, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, i, vn1, vn2, jp, tau);
                    LSQ_LAUNCH(k_qr_apply, dim3(lsq_div_up(n - i, 4)), dim3(256), 0, c->stream, R2, n, n, i, rhs2, tau,
                                       vn1, vn2);
                }
            }
            int ph = 0;
            if (n <= QRK_MAXN) {
                if (!have_rank)
                    LSQ_LAUNCH(k_qr_rank, dim3(1), dim3(64), 0, c->stream, R2, n, n, (double)mn * DBL_EPSILON, s->d_info);
                LSQ_LAUNCH(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, jp, s->d_info, d_x);
                ph = 4;
            }
            LSQ_LAUNCH(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, R2, n, n, rhs2, n, d_x, s->d_work, jp,
                               s->d_T, (double)mn * DBL_EPSILON, s->d_info, ph);
        } else if (n >= 64 && (long long)M * n >= 65536) {
            // multi-CU column-pivoted Householder (BLAS-2 per column, the rhs rides along as column n)
            double *ws = s->d_work;
            double *vn1 = ws, *vn2 = ws + n, *tau = ws + 2 * n;
            int *jp = (int *)s->d_tau;
            LSQ_LAUNCH(k_qr_norms, dim3(lsq_div_up(n, 4)), dim3(256), 0, c->stream, s->d_qr, M, n, vn1, vn2, jp);
            for (int i = 0; i < mn; ++i) {
                LSQ_LAUNCH(k_qr_pivot, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, i, vn1, vn2, jp, tau);
                LSQ_LAUNCH(k_qr_apply, dim3(lsq_div_up(n - i, 4)), dim3(256), 0, c->stream, s->d_qr, M, n, i,
                                   s->d_qu, tau, vn1, vn2);
            }
            int ph = 0;
            if (n <= QRK_MAXN && M >= n) {
                LSQ_LAUNCH(k_qr_rank, dim3(1), dim3(64), 0, c->stream, s->d_qr, M, mn, (double)mn * DBL_EPSILON, s->d_info);
                LSQ_LAUNCH(k_qr_backsolve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, jp, s->d_info,
                                   d_x);
                ph = 4;
            }
            LSQ_LAUNCH(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, lu, d_x,
                               s->d_work, jp, s->d_T, (double)mn * DBL_EPSILON, s->d_info, ph);
        } else {
            LSQ_LAUNCH(k_qrcp_solve, dim3(1), dim3(QR_NT), 0, c->stream, s->d_qr, M, n, s->d_qu, lu, d_x,
                               s->d_work, (int *)s->d_tau, s->d_T, (double)mn * DBL_EPSILON, s->d_info, 3);
        }
        LSQ_HIP(hipGetLastError());
        break;
    }
    s->last_rank = -1;  // fetched lazily by lsq_solver_info
    s->fb_qrx.solve_done(false);
    s->fb_cholqr.solve_done(false);
    if (nmul) *nmul = 1;
    return LSQ_OK;
}
