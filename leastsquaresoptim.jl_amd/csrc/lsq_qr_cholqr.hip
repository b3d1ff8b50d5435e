// Communication-avoiding panel of the blocked Householder QR (stage 1 of lsq_qr_solve; dense_qr.jl:30-88 = geqp3 + the
// xGELSY solve in the reference, restated as unpivoted blocked QR + certificate / pivoted sweep on R, lsq_dense.hip).
//
// The BLAS-2 panel (k_qr1_step_multi) is a chain of dependent launches: 16 per 64 columns, 14 us each, 55 % of the C3
// solve (16384 x 2048).  Here a 64-column panel P (rows x 64) is orthogonalised with TWO passes over it and no
// column-by-column communication:
//
//   pass 1   G1 = P'P (fp64 MFMA, split over 64-row slabs, fixed-order reduce);  R1 = chol(G1);  Q1 = P inv(R1)
//   pass 2   G2 = Q1'Q1 = I + E;  R2 = chol(G2) -- or, when ||E||_F <= 1e-5, its second-order expansion
//            R2 = I + X, X = Phi(E) - Phi(Phi(E)'Phi(E)), Phi = strict upper + half the diagonal, with no dependent chain at
//            all;  Q = Q1 inv(R2)  (CholeskyQR2: ||Q'Q - I|| = O(eps) whenever cond(P) <~ 1e7; beyond that pass 1 breaks
//            down or ||E|| > 1/2, a device flag is raised and the caller repeats the factorisation with the BLAS-2 panel)
//
// and the orthogonal transformation of the block step is taken in the basis-kernel (compact WY) form built from Q itself
// (Yamamoto; Ballard et al., "Reconstructing Householder vectors from TSQR"):
//
//   S = diag(+-1) from the modified LU of Q_top (S_j = -sign of the running diagonal, pivots >= 1 in magnitude),
//   V = Q - [S; 0],   B = Q_top - S = L U,   Qfull = I - V T V',  T' = -inv(B) S,  Qfull' P = [S R2 R1; 0]
//
// so the existing block update applies unchanged:  W = V'[A2 | b] (k_qr1_vtb),  W2 = T'W = inv(B)(A2_top - S Q'[A2|b])
// (k_cqr_tw, one 64 x 64 x N MFMA product),  A2 -= V W2 (k_qr1_update).  The 64-step LU runs in ONE workgroup on a side
// stream (k_cqr_top: it needs the top 64 rows only, so it starts as soon as G2 is known and runs beside pass 2 and the
// V'A2 product), off the critical path for all but the narrowest trailing matrices.  Small-matrix work: lsq_small64.h.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "lsq_qr_cholqr.h"
#include "lsq_small64.h"

// (CQ_RS, CQ_QST and cq_slab_gram live in lsq_qr_cholqr.h: the trailing update of the previous panel forms the Gram partials too)
// pass kernels: inv(R) | R | scratch; the slab image (64 * CQ_QST doubles) lives over R + scratch once the factor is done:
// 75 KB instead of 109, i.e. two slab workgroups per CU (LM's stacked 18432-row operand has 288 slabs: 9.3 -> 8.5 ms).
// (Look-ahead -- panel k+1's passes on a high-priority stream beside the update of panel k -- was built on top of this in
//  round 3 and measured: the passes do start beside the LDS-staged update, but both stretch; 8.0 ms against 7.55 at C3.
//  Round 5 built it again on the LDS-free update, lsq_qr_stage1.hip: qr2_factor_core, and with ONE factor workgroup per pass,
//  k_cqr_factor below: 7.20 -> 7.06 ms at C3, taken while >= 1024 other columns remain.)
constexpr int CQ_LDS_DOUBLES = 2 * S64_MAT + S64_TMP;
static_assert(S64_MAT + S64_TMP >= 64 * CQ_QST, "slab image over R + scratch");
constexpr size_t CQ_LDS = (size_t)CQ_LDS_DOUBLES * sizeof(double);
constexpr size_t CQ_LDS_LU = (size_t)(4 * S64_MAT + S64_TMP) * sizeof(double);
constexpr size_t CQ_LDS_TW = (size_t)(2 * S64_MAT) * sizeof(double);
#ifdef CQ_TIMING   // phase time stamps of workgroup 0 (tools/micro/cqr_bench.hip only)
__device__ unsigned long long cq_tbuf[64];
#define CQ_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) cq_tbuf[k] = wall_clock64(); } while (0)
#else
#define CQ_T(k) do { } while (0)
#endif
#ifndef CQ_NO_FIRST_ORDER
#define CQ_NO_FIRST_ORDER 0                // (A/B builds: -DCQ_NO_FIRST_ORDER=1 keeps the second-order expansion for every panel)
#endif
constexpr int CQ_FAIL = 2;                 // bit of the solver's error word (bit 0: an in-kernel exchange timed out)

// 64 x 64 row-major matrix in global memory -> LDS image (stride S64_LS): all 16 loads of a thread in flight at once
// (one wavefront per SIMD here: a load per loop trip would cost a full memory latency each)
__device__ __forceinline__ void cq_load64(double *__restrict__ dst, const double *__restrict__ src, int tid) {
    double t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = src[tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q;
        dst[(e >> 6) * S64_LS + (e & 63)] = t[q];
    }
}

// The 64 x 64 factor of a pass from its reduced Gram matrix G (row-major, upper triangle valid):
//   on return  M1 = R (upper, zeros below),  M2 = inv(R).
// PASS 1: R1 = chol(G).   PASS 2: G = Q1'Q1 = I + E -- ||E||_F <= 1e-5: second-order expansion (no dependent chain: two
// structured MFMA products, the element-wise steps on 16 register-resident entries per thread); larger: Cholesky again;
// ||E||_F > 1/2: cond(Q1)^2 eps is no longer O(eps) -- *bad is set (as for a Cholesky breakdown).
// NOINV (pass 1, round 6): M2 keeps what s64_chol leaves there -- the four inv(U_kk)' diagonal blocks -- instead of the explicit
// inverse; the caller substitutes (cq_rows_trsm).
template <int PASS, bool NOINV = false>
__device__ __forceinline__ void cq_factor(const double *__restrict__ G, double *__restrict__ M1, double *__restrict__ M2,
                                          double *__restrict__ T, int *s_fail, double *s_red, int *bad, int tid, int ngroups = 0) {
    const int lane = tid & 63, wv = tid >> 6;
    double g[16];
    cq_gram_entries(G, ngroups, g, tid);                          // entry e = tid + 256 q: row e >> 6, column e & 63
    *bad = 0;
    bool series = false, first_order = false;
    if (PASS == 2) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            const double v = g[q] - (r == c ? 1.0 : 0.0);
            acc += r < c ? 2.0 * v * v : (r == c ? v * v : 0.0);      // (only the upper triangle of G is formed)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) s_red[wv] = acc;
        __syncthreads();
        const double fro2 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (!(fro2 <= 0.25)) *bad = 1;                                 // (NaN lands here too)
        series = fro2 <= 1e-10;
        first_order = fro2 <= 1e-22 && !CQ_NO_FIRST_ORDER;
    }
    if (series && first_order) {
        // ||E||_F <= 1e-11 (the usual case: Q1 of a well-conditioned panel is orthonormal to a few ulps): the second-order terms
        // X1'X1 and X2 X2 are below 1e-22 beside entries of size 1 -- R2 = I + Phi(E), inv(R2) = I - Phi(E) to the last bit
        // that matters, and two 64^3 products + three barriers leave k_cqr_top's chain (round 6, late: that chain decides most
        // panels since V'B reads its operands in fragment order)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            const double x = r < c ? g[q] : (r == c ? 0.5 * (g[q] - 1.0) : 0.0), d = r == c ? 1.0 : 0.0;
            M2[r * S64_LS + c] = d - x;
            M1[r * S64_LS + c] = d + x;
        }
        __syncthreads();
    } else if (series) {
        // X1 = Phi(E);  X2 = X1 - Phi(X1'X1);  inv(R2) = I - X2 + X2 X2;  R2 = I + X2      (Phi: strict upper + half diagonal)
        double x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            x[q] = r < c ? g[q] : (r == c ? 0.5 * (g[q] - 1.0) : 0.0);
            M1[r * S64_LS + c] = x[q];
        }
        __syncthreads();
        s64_gemm<true, false, S64_LtU>(M2, M1, M1, 1.0, tid);      // upper tiles of X1'X1
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            const double v = r <= c ? M2[r * S64_LS + c] : 0.0;
            x[q] -= r < c ? v : (r == c ? 0.5 * v : 0.0);
        }
        __syncthreads();                                           // (everyone has read M2 / the product has read M1)
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; M1[(e >> 6) * S64_LS + (e & 63)] = x[q]; }
        __syncthreads();
        s64_gemm<false, false, S64_UU>(M2, M1, M1, 1.0, tid);      // X2 X2 (upper; zero tiles below)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            const double d = r == c ? 1.0 : 0.0;
            M2[r * S64_LS + c] = d - x[q] + M2[r * S64_LS + c];
            M1[r * S64_LS + c] = d + x[q];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; M1[(e >> 6) * S64_LS + (e & 63)] = g[q]; }
        __syncthreads();
        if (s64_chol(M1, M2, s_fail, tid)) *bad = 1;
        if (!NOINV) s64_chol_inverse(M1, M2, T, tid);
    }
}

// Qslab = Pslab inv(U) WITHOUT the explicit inverse (round 6): the wavefront's 16 rows of the slab, fetched in the A-operand
// layout (cq_rows_fetch: a[4 r + rr] = P[row][16 r + 4 rr + kq]), are exactly the B-operand fragments of X = P' that
// s64_trsm_blocks (lsq_dense_mfma.hip) loads from LDS: X <- inv(U)' X by forward substitution over the four 16-column blocks,
// the four stages carried in registers (the accumulator layout of one product is the B-operand layout of the next).  Needs U and
// the inv(U_kk)' blocks s64_chol leaves in W; saves s64_chol_inverse (two levels of tile products behind five barriers: ~5 us of
// the 18 us every pass-1 workgroup spends on its factor).  x[r][rr] = Q[row 16 wv + ij][column 16 r + kq + 4 rr].
__device__ __forceinline__ void cq_rows_trsm(const double (&a)[16], const double *__restrict__ U, const double *__restrict__ W,
                                             s64_v4d x[4], int lane) {
    const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s64_v4d t = {a[4 * r], a[4 * r + 1], a[4 * r + 2], a[4 * r + 3]};
        if (r > 0) {
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < r; ++q)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)      // U(q, r)' X_q : A[i][k] = U[16 q + k][16 r + i]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[(16 * q + 4 * kk + kq) * S64_LS + 16 * r + ij], x[q][kk], acc, 0, 0, 0);
            t -= acc;
        }
        s64_v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(16 * r + ij) * S64_LS + 16 * r + 4 * kk + kq], t[kk], y, 0, 0, 0);
        x[r] = y;
    }
}

// rows r0.. of a 64-column block times the upper triangular inv(R) in M2: one 16-row tile per wavefront, K = 64 (zero
// blocks skipped); the lane's part of the product -> acc[4] (MFMA D layout: column 16 j + (lane & 15), rows (lane >> 4) + 4 r).
// In two halves, so that the rows can be requested BEFORE the factor that produces inv(R) (a memory round trip of ~2 us that
// would otherwise follow the 18 us factor on every pass-1 workgroup's path).
__device__ __forceinline__ void cq_rows_fetch(const double *__restrict__ P, size_t lda, int nr, double (&a)[16], int lane, int wv) {
    const int ij = lane & 15, kq = lane >> 4;
    const int row = wv * 16 + ij;
    const bool rin = row < nr;
    const double *prow = P + (rin ? row : 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const double v = prow[(size_t)(s * 4 + kq) * lda];
        a[s] = rin ? v : 0.0;
    }
}
__device__ __forceinline__ void cq_rows_mma(const double (&a)[16], const double *__restrict__ M2, s64_v4d acc[4], int lane) {
    const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (s64_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = s * 4 + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j >= (s >> 2))       // block (k-tile, j) of an upper triangular matrix is zero for j < k-tile
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], M2[k * S64_LS + 16 * j + ij], acc[j], 0, 0, 0);
    }
}
__device__ __forceinline__ void cq_rows_times_inv(const double *__restrict__ P, size_t lda, int nr, const double *__restrict__ M2,
                                                  s64_v4d acc[4], int lane, int wv) {
    double a[16];
    cq_rows_fetch(P, lda, nr, a, lane, wv);
    cq_rows_mma(a, M2, acc, lane);
}

// PASS 0: Gram partials of the raw panel.   PASS 1: R1 = chol(G), Q1 = P inv(R1) in place, Gram partials of Q1.
// PASS 2: R2 from G2 = Q1'Q1, Q = Q1 inv(R2) -> Vb.      P = A(c0 + [0, rows), c0 + [0, 64)), column-major, ld lda.
// PRE: inv(R) was computed once by k_cqr_factor (G points at it) instead of by every workgroup -- the look-ahead panel, whose
// passes share the CUs with the trailing update: 256 redundant 64 x 64 factorisations are 18 us of every CU's ALU and LDS time.
template <int PASS, bool PRE = false>
__global__ void __launch_bounds__(256)
k_cqr_pass(double *__restrict__ A, int lda, int c0, int rows, const double *__restrict__ G, double *__restrict__ Gp,
           double *__restrict__ R1g, double *__restrict__ Vb, int ldv, int *__restrict__ err, int q1vb = 0 /* PASS 1: Q1 -> Vb
           instead of in place (the Q1 form of the block reflector, round 6: no pass 2) */,
           int ngroups_in = 0 /* > 0: G holds that many group sums (cq_group_reduce), added here */,
           double *Gq = nullptr /* non-null: this launch's own partials are summed per group into Gq */, unsigned *gcnt = nullptr,
           double *__restrict__ Vs = nullptr /* PASS 1, q1vb: Q1 also in V'B's fragment order (lsq_cqr_vs_index) */) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *M2 = sm, *M1 = sm + S64_MAT, *T = sm + 2 * S64_MAT, *Qs = M1;   // (Qs aliases R and the scratch: used after them)
    __shared__ int s_fail;
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slab = blockIdx.x, r0 = slab * CQ_RS, nr = min(CQ_RS, rows - r0);
    double *P = A + (size_t)c0 * lda + c0 + r0;         // element (row, col) at P[col * lda + row]
    CQ_T(PASS * 16 + 0);
    if (PASS == 0) {
        // thread: row (tid & 63), columns (tid >> 6) + 4 q -- 16 loads in flight
        const int row = tid & (CQ_RS - 1), cq = tid >> 6;
        const bool rin = row < nr;
        const double *src = P + (rin ? row : 0);
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = src[(size_t)(cq + 4 * q) * lda];
#pragma unroll
        for (int q = 0; q < 16; ++q) Qs[(cq + 4 * q) * CQ_QST + row] = rin ? t[q] : 0.0;
        __syncthreads();
        CQ_T(1);
        if (Gq) {
            cq_slab_gram<true>(Qs, Gp + (size_t)slab * 4096, tid);
            cq_group_reduce(Gp, Gq, gcnt, slab, (int)gridDim.x, tid);
        } else
            cq_slab_gram(Qs, Gp + (size_t)slab * 4096, tid);
        CQ_T(2);
        return;
    }
    double prow[16];                                              // this wavefront's 16 rows of the slab: requested before the factor
    cq_rows_fetch(P, (size_t)lda, nr, prow, lane, wv);
    if (PRE) {
        cq_load64(M2, G, tid);                                    // inv(R), row-major 64 x 64
    } else {
        int bad;
        cq_factor<PASS, PASS == 1>(G, M1, M2, T, &s_fail, s_red, &bad, tid, ngroups_in);     // every workgroup, identically
        if (bad && slab == 0 && tid == 0) atomicOr(err, CQ_FAIL);
        CQ_T(PASS * 16 + 2);
        if (PASS == 1 && slab == 0)
            for (int e = tid; e < 4096; e += 256) R1g[e] = M1[(e >> 6) * S64_LS + (e & 63)];
    }
    __syncthreads();                                              // (PRE / pass 2: R is dead from here on: its LDS becomes the slab image)
    // ---- slab product  Qslab = Pslab * inv(R) ------------------------------------------------------------------
    CQ_T(PASS * 16 + 3);
    if (PASS == 1 && !PRE) {
        // by substitution with R itself and its inverted 16 x 16 diagonal blocks (cq_rows_trsm), from the prefetched rows
        s64_v4d x[4];
        cq_rows_trsm(prow, M1, M2, x, lane);
        __syncthreads();                                          // every wavefront has read R: the slab image may overwrite it
        const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Qs[(16 * r + kq + 4 * rr) * CQ_QST + wv * 16 + ij] = x[r][rr];
    } else {
        s64_v4d acc[4];
        cq_rows_mma(prow, M2, acc, lane);
        const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Qs[(16 * j + ij) * CQ_QST + wv * 16 + kq + 4 * r] = acc[j][r];
    }
    __syncthreads();
    CQ_T(PASS * 16 + 4);
    const bool inplace = PASS == 1 && !q1vb;
    double *dst = inplace ? P : Vb + r0;
    const size_t ldd = inplace ? (size_t)lda : (size_t)ldv;
    {
        const int row = tid & (CQ_RS - 1), cq = tid >> 6;
        if (row < nr) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dst[(cq + 4 * q) * ldd + row] = Qs[(cq + 4 * q) * CQ_QST + row];
        }
    }
    if (PASS == 1 && Vs && q1vb) {
        // the slab's full 16-row chunks once more, in the order in which k_qr1_vtb_w's lanes consume them (1 KB contiguous per
        // load instruction there): block = (chunk, 16-column tile it, half h), lane = ij + 16 kq, two doubles per lane
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q, ln = idx & 63, blk = idx >> 6, h = blk & 1, it = (blk >> 1) & 3, cc = blk >> 3;
            if (16 * (cc + 1) <= nr) {
                const int ij = ln & 15, kq = ln >> 4;
                const double *src = Qs + (16 * it + ij) * CQ_QST + 16 * cc + 4 * kq + 2 * h;
                double2 v;
                v.x = src[0];
                v.y = src[1];
                *reinterpret_cast<double2 *>(Vs + lsq_cqr_vs_index(slab * 4 + cc, it, h, ln)) = v;
            }
        }
    }
    CQ_T(PASS * 16 + 5);
    if (PASS == 1) {
        if (Gq) {
            cq_slab_gram<true>(Qs, Gp + (size_t)slab * 4096, tid);
            cq_group_reduce(Gp, Gq, gcnt, slab, (int)gridDim.x, tid);
        } else
            cq_slab_gram(Qs, Gp + (size_t)slab * 4096, tid);
    }
    CQ_T(PASS * 16 + 6);
}

// ONE workgroup: the factor of a pass from its reduced Gram matrix, inv(R) -> Minv (row-major), R1 -> R1g (pass 1)
template <int PASS>
__global__ void __launch_bounds__(256)
k_cqr_factor(const double *__restrict__ G, double *__restrict__ Minv, double *__restrict__ R1g, int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *M2 = sm, *M1 = sm + S64_MAT, *T = sm + 2 * S64_MAT;
    __shared__ int s_fail;
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    int bad;
    cq_factor<PASS>(G, M1, M2, T, &s_fail, s_red, &bad, tid);
    if (bad && tid == 0) atomicOr(err, CQ_FAIL);
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) {
        Minv[e] = M2[(e >> 6) * S64_LS + (e & 63)];
        if (PASS == 1) R1g[e] = M1[(e >> 6) * S64_LS + (e & 63)];
    }
}

// G = sum of the slab partials (fixed order): 16 outputs per workgroup, 16 partial sums per output (one batch of loads
// per thread for up to 256 slabs), combined in index order
__global__ void __launch_bounds__(256) k_cqr_reduce(const double *__restrict__ Gp, int nslab, double *__restrict__ G) {
    __shared__ double part[16][17];
    const int tid = threadIdx.x, o = tid & 15, p = tid >> 4;
    const int e = blockIdx.x * 16 + o;
    // (a workgroup's 16 outputs are one row of one 16 x 16 tile; the strictly LOWER tiles are never formed by cq_slab_gram nor read
    //  by cq_factor: 6 of 16 -- 3 MB of the 8 MB of partials at 256 slabs -- are not fetched)
    if (((blockIdx.x * 16) >> 10) > ((blockIdx.x * 16 & 63) >> 4)) {
        if (p == 0) G[e] = 0.0;
        return;
    }
    const int per = (nslab + 15) / 16, s0 = p * per, s1 = min(nslab, s0 + per);
    double acc = 0.0;
    int s = s0;
    for (; s + 16 <= s1; s += 16) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = Gp[(size_t)(s + u) * 4096 + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u];
    }
    for (; s < s1; ++s) acc += Gp[(size_t)s * 4096 + e];
    part[p][o] = acc;
    __syncthreads();
    if (p == 0) {
        double tot = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) tot += part[u][o];
        G[e] = tot;
    }
}

// ONE workgroup, on the side stream, beside pass 2 and the caller's V'[A2 | b] product: everything that hangs on the TOP
// 64 rows only.  It repeats pass 2's factor and forms Q_top = Q1_top inv(R2) itself (so it needs nothing from pass 2),
// then: modified LU of B = Q_top - S (S on the fly), inv(B) = inv(U) inv(L) and S -> global;
// R = R2 R1 and the panel's part of the factor, S R, -> SRg (64 x 64, [col][row]).  It does NOT store into A: pass 2 runs
// beside it on the main stream and its slab 0 reads Q1's top rows from exactly the elements S R belongs in (a write
// after read with nothing ordering it -- the round-3 defect: a host stall between the two launches let the store win and
// slab 0 built V's top rows from S R).  k_cqr_tw, behind both kernels, moves SRg into A's triangle.
// Q1 form (round 6, R2inv != null): there is no pass 2 at all -- the block reflector is applied with Q1 and the small factor
// inv(R2) is folded into the 64 x N matrices (k_cqr_tw_q1), so this kernel also hands out inv(R2), reads Q1's top rows from
// where pass 1 left them (Vb: Q1top / ldq) and raises the breakdown flag that pass 2 used to raise.
__global__ void __launch_bounds__(256)
k_cqr_top(const double *__restrict__ G2, const double *__restrict__ R1g, const double *__restrict__ Q1top, int ldq,
          double *__restrict__ Binv, double *__restrict__ Sg, double *__restrict__ SRg, int *__restrict__ err,
          double *__restrict__ R2inv, int lu_only, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *B0 = sm, *B1 = sm + S64_MAT, *B2 = sm + 2 * S64_MAT, *B3 = sm + 3 * S64_MAT, *T = sm + 4 * S64_MAT;
    __shared__ double sS[64], sR[64], s_red[4];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    CQ_T(48);
    // (this workgroup is a chain of latencies on every panel's path: what comes from memory -- Q1's top rows, R1 -- is requested
    //  before the factor, together with G2)
    double qtop[16], r1v[16];
    cq_rows_fetch(Q1top, (size_t)ldq, 64, qtop, lane, wv);         // Q1's top rows (pass 1: in place in A, or in Vb)
#pragma unroll
    for (int q = 0; q < 16; ++q) r1v[q] = R1g[tid + 256 * q];
    int bad;
    cq_factor<2>(G2, B0, B1, T, &s_fail, s_red, &bad, tid, ngroups);   // B0 = R2, B1 = inv(R2)  (old form: the flag is raised by pass 2)
    if (R2inv) {
        if (bad && tid == 0) atomicOr(err, CQ_FAIL);
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; R2inv[e] = B1[(e >> 6) * S64_LS + (e & 63)]; }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; B2[(e >> 6) * S64_LS + (e & 63)] = r1v[q]; }
    __syncthreads();
    s64_gemm<false, false, S64_UU>(B3, B0, B2, 1.0, tid);          // B3 = R = R2 R1
    {   // Q_top (row-major) -> B2
        s64_v4d acc[4];
        cq_rows_mma(qtop, B1, acc, lane);
        __syncthreads();                                           // (the product above has read B2 = R1)
        const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) B2[(wv * 16 + kq + 4 * r) * S64_LS + 16 * j + ij] = acc[j][r];
    }
    __syncthreads();
    CQ_T(49);
    // ---- TALL PANELS (round 6): the kernel of the block reflector WITHOUT a dependent chain -------------------------------
    // Q has orthonormal columns, so ||Q_top||_2 <= 1, and for a panel of many rows it is small (~2 sqrt(64 / rows)).  The
    // basis-kernel form  Qfull = I - V T V',  V = Q - [S; 0],  T' = inv(I - S Q_top)  is an orthogonal matrix with Qfull [S; 0]
    // = Q for ANY diagonal sign matrix S for which N = I - S Q_top is regular (T^-1 + T^-T = V'V = N + N') -- the sequential
    // sign choice of the modified LU only serves stability (pivots >= 1 when Q_top is not small).  So with E = S Q_top,
    // S_j = -sign(Q_top[j][j]) taken up front (diag(N) >= 1), and rho(E) <= 1/2 PROVEN by ||E^2||_F <= 1/4:
    //      inv(N) = (I + E)(I + E^2)(I + E^4) ...       inv(B) = inv(Q_top - S) = -inv(N) S
    // by repeated squaring on the MFMA unit, until the measured ||E^(2^k)||_F^2 says that the next factor is below 1e-34
    // (cond(N) <= (1 + 1)(4/3)): ~8 products of 64^3 at C3 instead of the 64-step LU + two triangular inversions + a product.
    // Otherwise (short panels: Q_top is a sizeable part of Q) the modified LU below.  LSQ_QR_TOP_LU=1 (lu_only) forces it.
    if (!lu_only) {
        double *E = B0, *P = B1, *F = B3, *Tm = B2;                // (B3 = R until S R is stored; B2 = Q_top until the decision)
        double ev[16];
        if (tid < 64) sS[tid] = B2[tid * S64_LS + tid] >= 0.0 ? -1.0 : 1.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            ev[q] = (B2[r * S64_LS + r] >= 0.0 ? -1.0 : 1.0) * B2[r * S64_LS + c];      // E = S Q_top
            E[r * S64_LS + c] = ev[q];
        }
        __syncthreads();
        s64_gemm<false, false, S64_FULL>(P, E, E, 1.0, tid);       // E^2 -> P for now (R still sits in F's buffer)
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; const double v = P[(e >> 6) * S64_LS + (e & 63)]; acc += v * v; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) s_red[wv] = acc;
        __syncthreads();
        double fro2 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (fro2 <= 0.0625) {                                      // (uniform; NaN fails the test and takes the LU path, which reports)
            // the panel's part of the factor, S R, and S -- with the signs chosen above
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + 256 * q, col = e >> 6, row = e & 63;
                SRg[e] = row <= col ? sS[row] * B3[row * S64_LS + col] : 0.0;
            }
            if (tid < 64) Sg[tid] = sS[tid];
            __syncthreads();                                       // (R and Q_top are dead from here on)
#pragma unroll
            for (int q = 0; q < 16; ++q) {                         // F = E^2 (moved out of P), P = I + E
                const int e = tid + 256 * q, r = e >> 6, c = e & 63;
                F[r * S64_LS + c] = P[r * S64_LS + c];
                P[r * S64_LS + c] = (r == c ? 1.0 : 0.0) + ev[q];
            }
            __syncthreads();
            double *Fc = F, *Fn = E;                               // (E itself is no longer needed)
            CQ_T(53);
            for (int it = 0; it < 8; ++it) {
                s64_gemm<false, false, S64_FULL>(Tm, P, Fc, 1.0, tid);     // P <- P (I + F)
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q, o = (e >> 6) * S64_LS + (e & 63); P[o] += Tm[o]; }
                // ||F^2||_F <= ||F||_F^2: once that is below 1e-34 the next factor would not change a bit of P
                if (fro2 <= 1e-17) break;                           // (uniform)
                __syncthreads();
                s64_gemm<false, false, S64_FULL>(Fn, Fc, Fc, 1.0, tid);    // F <- F^2
                acc = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int e = tid + 256 * q; const double v = Fn[(e >> 6) * S64_LS + (e & 63)]; acc += v * v; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                if (lane == 0) s_red[wv] = acc;                     // (the previous round's readers are past two barriers)
                __syncthreads();
                fro2 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
                double *sw = Fc; Fc = Fn; Fn = sw;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {                         // inv(B) = -inv(N) S: column c scaled by -S_c (row-major, to global)
                const int e = tid + 256 * q, r = e >> 6, c = e & 63;
                Binv[e] = -P[r * S64_LS + c] * sS[c];
            }
            CQ_T(52);
            return;
        }
        __syncthreads();                                           // (P, E are scratch again; B2 = Q_top and B3 = R are intact)
    }
    double *M = B2, *Li = B0;
    s64_lu_modified(M, Li, sS, sR, tid);
    CQ_T(50);
    // the panel's part of the factor, S R (upper triangle; zeros below), for k_cqr_tw to put into A
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, col = e >> 6, row = e & 63;
        SRg[e] = row <= col ? sS[row] * B3[row * S64_LS + col] : 0.0;
    }
    if (tid < 64) Sg[tid] = sS[tid];
    __syncthreads();
    // inv(L) = (inv(L'))': Z = L' (upper, unit diagonal) -> B1, X <- inv(Z) -> B3, diagonal blocks inv(L_kk)' from Li
    double *Z = B1, *X = B3;
    for (int e = tid; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        Z[r * S64_LS + c] = r < c ? M[c * S64_LS + r] : (r == c ? 1.0 : 0.0);
        X[r * S64_LS + c] = ((r >> 4) == (c >> 4)) ? Li[c * S64_LS + r] : 0.0;
    }
    __syncthreads();
    s64_triinv_levels(Z, X, T, tid);           // X = inv(L') = inv(L)'
    // Uc = clean upper part of M -> B0 (Li is done); inv(U) -> B1 (Z is done)
    double *Uc = B0, *Xu = B1;
    for (int e = tid; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        Uc[r * S64_LS + c] = r <= c ? M[r * S64_LS + c] : 0.0;
    }
    __syncthreads();
    s64_diaginv_upper(Uc, Xu, tid);
    s64_triinv_levels(Uc, Xu, T, tid);
    CQ_T(51);
    s64_gemm<false, true, S64_UL>(Binv, Xu, X, 1.0, tid, 64);     // inv(B) = inv(U) inv(L) = Xu * X' -> global (row-major)
    CQ_T(52);
}

// W2(:, 64 columns of the workgroup) = inv(B) (A2_top - S W_Q)   [W = V'[V | A2 | b] with V = Q, from k_qr1_vtb + wreduce]
// B-matrix column cb >= 64 is A(:, cend + cb - 64) or the right-hand side (last column); rows from c0.
__global__ void __launch_bounds__(256)
k_cqr_tw(const double *__restrict__ W, int ncolsB, const double *__restrict__ Binv, const double *__restrict__ Sg,
         const double *__restrict__ SRg, double *A /* read: trailing columns' top rows; written: the panel's triangle */, int lda,
         int c0, int cend, int n, const double *__restrict__ rhs, double *__restrict__ Vb, int ldv, double *__restrict__ W2) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sBi = sm;                      // inv(B), row-major
    double *sRh = sm + S64_MAT;            // RHS[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ncols = ncolsB - 64, j0 = blockIdx.x * 64;
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        // ONE EXTRA workgroup, beside the ones that form W2: V = Q - [S; 0] for the update, and the panel's part of R -> A.
        // (This launch is ordered behind pass 2 -- the last reader of Q1's top rows -- by the stream and behind k_cqr_top by
        //  ev_lu; nothing in this kernel reads the panel's columns of A.  In workgroup 0, as first written in round 4, the
        //  copy put a memory round trip in front of that workgroup's product: 11.2 instead of 7.7 us per launch.)
        if (tid < 64) Vb[(size_t)tid * ldv + tid] -= Sg[tid];
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = SRg[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, col = e >> 6, row = e & 63;
            if (row <= col) A[(size_t)(c0 + col) * lda + c0 + row] = t[q];
        }
        return;
    }
    cq_load64(sBi, Binv, tid);
    {
        double top[16], wq[16];
        const double sk = Sg[tid & 63];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, j = e >> 6, k = e & 63, col = min(j0 + j, ncols - 1), a = cend + col;
            top[q] = a < n ? A[(size_t)a * lda + c0 + k] : rhs[c0 + k];
            wq[q] = W[(size_t)(64 + col) * 64 + k];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, j = e >> 6, k = e & 63;
            sRh[k * S64_LS + j] = j0 + j < ncols ? top[q] - sk * wq[q] : 0.0;
        }
    }
    __syncthreads();
    const int ij = lane & 15, kq = lane >> 4;
    for (int q = wv; q < 16; q += 4) {
        const int ti = q >> 2, tj = q & 3;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = s * 4 + kq;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sBi[(16 * ti + ij) * S64_LS + k], sRh[k * S64_LS + 16 * tj + ij], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + kq + 4 * r, col = j0 + 16 * tj + ij;
            if (col < ncols) W2[(size_t)col * 64 + row] = acc[r];
        }
    }
}

// Q1 FORM of the block reflector (round 6).  With Q = Q1 inv(R2), V = Q - [S; 0]:
//      V'[A2 | b] = inv(R2)' (Q1'[A2 | b]) - S A2_top          A2 - V W2 = A2 - Q1 (inv(R2) W2) + [S W2; 0]
// so pass 2 over the panel (Q = Q1 inv(R2) -> Vb: a launch of 13 us on the critical path of every panel, plus the Gram reduce
// in front of it) is not needed: the trailing kernels take Q1 as it leaves pass 1, and the 64 x 64 factor inv(R2) = I + O(eps
// cond(P)^2) is applied to the 64 x N matrices here.  W comes in as Q1'[A2 | b] (k_qr1_vtb on Q1 + k_qr1_wreduce);
//      WQ = inv(R2)' W;   W2 = inv(B)(A2_top - S WQ);   A2_top += S W2  (written back: every workgroup owns its 64 columns);
//      W3 = inv(R2) W2 -> the update kernel's operand (it then subtracts Q1 W3 from ALL rows of A2, the top ones included).
// R2inv, Binv, S, SR: k_cqr_top on the side stream (ev_lu).  The extra workgroup puts the panel's part of R into A.
// SIXTEEN columns per workgroup (the first form took 64, like k_cqr_tw: three dependent 64^3 products by one CU -- 144 MFMAs per
// wavefront at one per ~100 clocks -- made the launch 15 us on every panel's path; with 16 columns a wavefront owns ONE 16 x 16 tile
// of each product, at most 16 MFMAs, and four times as many CUs share the work).
constexpr int TQ_NC = 16;                  // columns of [A2 | b] per workgroup
constexpr int TQ_XS = TQ_NC + 1;           // row stride of the 64 x 16 images
constexpr size_t CQ_LDS_TW_Q1 = (size_t)(2 * S64_MAT + 2 * 64 * TQ_XS) * sizeof(double);
// Y (64 x 16) = op(A) (64 x 64, LDS, stride S64_LS) * X (64 x 16): wavefront wv forms rows 16 wv .. 16 wv + 15.
//   SHAPE S64_LF: op(A) = A' with A upper triangular (k tiles 0 .. wv);  S64_FULL;  S64_UF: A upper triangular (k tiles wv .. 3)
template <bool TA, int SHAPE>
__device__ __forceinline__ void tq_gemm16(double *__restrict__ Y, const double *__restrict__ A, const double *__restrict__ X, int tid) {
    const int lane = tid & 63, wv = tid >> 6, ij = lane & 15, kq = lane >> 4;
    const int k0 = SHAPE == S64_UF ? wv : 0, k1 = SHAPE == S64_LF ? wv + 1 : 4;
    s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int kt = k0; kt < k1; ++kt) {
        double a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kt * 16 + kk * 4 + kq;
            a[kk] = TA ? A[k * S64_LS + 16 * wv + ij] : A[(16 * wv + ij) * S64_LS + k];
            b[kk] = X[k * TQ_XS + ij];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[(16 * wv + kq + 4 * r) * TQ_XS + ij] = acc[r];
    __syncthreads();
}
__global__ void __launch_bounds__(256)
k_cqr_tw_q1(const double *__restrict__ W, int ncolsB, const double *__restrict__ Binv, const double *__restrict__ Sg,
            const double *__restrict__ SRg, const double *__restrict__ R2inv, double *A, int lda, int c0, int cend, int n,
            double *rhs, double *__restrict__ W2, double *__restrict__ W2s /* or null: lsq_cqr_w2s_index */) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sBi = sm, *sRi = sm + S64_MAT, *X = sm + 2 * S64_MAT, *Y = X + 64 * TQ_XS;
    const int tid = threadIdx.x;
    const int ncols = ncolsB - 64, j0 = blockIdx.x * TQ_NC;
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        double t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = SRg[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, col = e >> 6, row = e & 63;
            if (row <= col) A[(size_t)(c0 + col) * lda + c0 + row] = t[q];
        }
        return;
    }
    const int k = tid & 63;                       // (e = tid + 256 q: row k = e & 63 is the thread's own, column j = e >> 6)
    double top[4], wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // (requested with the two 64 x 64 factors: one memory round trip in all)
        const int j = (tid >> 6) + 4 * q, col = min(j0 + j, ncols - 1), a = cend + col;
        top[q] = a < n ? A[(size_t)a * lda + c0 + k] : rhs[c0 + k];
        wq[q] = W[(size_t)(64 + col) * 64 + k];
    }
    const double sk = Sg[k];
    {
        double tb[16], tr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { tb[q] = Binv[tid + 256 * q]; tr[q] = R2inv[tid + 256 * q]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, o = (e >> 6) * S64_LS + (e & 63);
            sBi[o] = tb[q];
            sRi[o] = tr[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = (tid >> 6) + 4 * q;
        X[k * TQ_XS + j] = j0 + j < ncols ? wq[q] : 0.0;
    }
    __syncthreads();
    tq_gemm16<true, S64_LF>(Y, sRi, X, tid);                       // Y = WQ = inv(R2)' (Q1'[A2 | b])
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = (tid >> 6) + 4 * q;
        Y[k * TQ_XS + j] = j0 + j < ncols ? top[q] - sk * Y[k * TQ_XS + j] : 0.0;
    }
    __syncthreads();
    tq_gemm16<false, S64_FULL>(X, sBi, Y, tid);                    // X = W2 = inv(B)(A2_top - S WQ)
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                  // A2_top += S W2: the [S W2; 0] part of V W2
        const int j = (tid >> 6) + 4 * q, col = j0 + j, a = cend + col;
        if (col < ncols) {
            const double v = top[q] + sk * X[k * TQ_XS + j];
            if (a < n) A[(size_t)a * lda + c0 + k] = v;
            else rhs[c0 + k] = v;
        }
    }
    tq_gemm16<false, S64_UF>(Y, sRi, X, tid);                      // Y = W3 = inv(R2) W2
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = (tid >> 6) + 4 * q;
        if (j0 + j < ncols) W2[(size_t)(j0 + j) * 64 + k] = Y[k * TQ_XS + j];
        // the same values in the trailing update's FRAGMENT order (this workgroup's 16 columns are one of its 16-column tiles;
        // columns past the end are zeros here)
        if (W2s) W2s[lsq_cqr_w2s_index((int)blockIdx.x, j, k)] = Y[k * TQ_XS + j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
int lsq_cqr_alloc(lsq_ctx *c, CqrWork *w, int M) {
    w->max_slabs = (M + CQ_RS - 1) / CQ_RS;
    LSQ_HIP(hipMalloc(&w->Gp, (size_t)w->max_slabs * 4096 * sizeof(double)));
    LSQ_ZERO(w->Gp, 0, (size_t)w->max_slabs * 4096 * sizeof(double));   // (the strictly lower tiles are never written; LSQ_ZERO waits:
    // a bare hipMemset runs on the null stream, which the context's non-blocking stream does not order itself after)
    LSQ_HIP(hipMalloc(&w->G, 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->R1, 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->G2, 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->Binv, 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->Minv, 2 * 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->R2inv, 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->Gq1, (size_t)CQ_HIER_MAX_GROUPS * 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->Gq2, (size_t)CQ_HIER_MAX_GROUPS * 4096 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->gcnt, CQ_HIER_MAX_GROUPS * sizeof(unsigned)));
    LSQ_ZERO(w->Gq1, 0, (size_t)CQ_HIER_MAX_GROUPS * 4096 * sizeof(double));     // (the strictly lower tiles are never written)
    LSQ_ZERO(w->Gq2, 0, (size_t)CQ_HIER_MAX_GROUPS * 4096 * sizeof(double));
    LSQ_ZERO(w->gcnt, 0, CQ_HIER_MAX_GROUPS * sizeof(unsigned));
    LSQ_HIP(hipMalloc(&w->S, 64 * sizeof(double)));
    LSQ_HIP(hipMalloc(&w->SR, 4096 * sizeof(double)));
    {   // highest priority: k_cqr_top's single workgroup (141 KB of LDS) must get a CU before the caller's V'[A2 | b] grid fills
        // them.  The two streams belong to the CONTEXT (lsq_ctx::helper_stream): solvers of one context run one after the other
        int lo = 0, hi = 0;
        LSQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int k = 0; k < 2; ++k)
            if (!c->helper_stream[k]) LSQ_HIP(hipStreamCreateWithPriority(&c->helper_stream[k], hipStreamNonBlocking, hi));
        w->side = c->helper_stream[0];
        w->ahead = c->helper_stream[1];
    }
    LSQ_HIP(hipEventCreateWithFlags(&w->ev_first, hipEventDisableTiming));
    LSQ_HIP(hipEventCreateWithFlags(&w->ev_panel, hipEventDisableTiming));
    LSQ_HIP(hipEventCreateWithFlags(&w->ev_q, hipEventDisableTiming));
    LSQ_HIP(hipEventCreateWithFlags(&w->ev_lu, hipEventDisableTiming));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_pass<0>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_pass<1>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_pass<2>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_pass<1, true>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_pass<2, true>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_factor<1>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_factor<2>, CQ_LDS));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_top, CQ_LDS_LU));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_tw, CQ_LDS_TW));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_tw_q1, CQ_LDS_TW_Q1));
    w->ready = true;
    return LSQ_OK;
}

void lsq_cqr_free(CqrWork *w) {
    if (!w || !w->ready) return;
    hipFree(w->Gp); hipFree(w->G); hipFree(w->G2); hipFree(w->R1); hipFree(w->Binv); hipFree(w->Minv); hipFree(w->R2inv); hipFree(w->Gq1); hipFree(w->Gq2); hipFree(w->gcnt); hipFree(w->S); hipFree(w->SR);
    hipEventDestroy(w->ev_q); hipEventDestroy(w->ev_lu); hipEventDestroy(w->ev_first); hipEventDestroy(w->ev_panel);
    w->side = w->ahead = nullptr;          // (the context's)
    w->ready = false;
}

// The Q1 form (default since round 6; LSQ_QR_CQR_PASS2=1 restores the three-pass panel of rounds 2-5 for A/B): read per call.
bool lsq_cqr_q1form() { return getenv("LSQ_QR_CQR_PASS2") == nullptr; }

// MEASURED AND NOT TAKEN (round 6, profiles/r06/ab_c3_hier.txt): with the group sums k_cqr_pass<1> takes 42 instead of 24 us,
// k_cqr_top 69 instead of 49, the Gram-forming update 68 instead of 62 -- the agent-scope stores of the partials, the last
// arrivers' uncached re-reads and four rounds of loads in every consumer cost more than the two 9-19 us reduce launches they
// replace: C3 7.38 ms against 6.83.  Kept behind LSQ_QR_HIER=1 (the GPU suite passes with it).
bool lsq_cqr_hier(int nslab) {
    return lsq_cqr_q1form() && nslab <= CQ_GS * CQ_HIER_MAX_GROUPS && getenv("LSQ_QR_HIER") != nullptr;
}

int lsq_cqr_panel(lsq_ctx *c, CqrWork *w, double *A, int M, int c0, double *Vb, int ldv, int *d_err, hipStream_t ps,
                  bool gram_ready, bool hier, double *Vs) {
    const int rows = M - c0, nslab = (rows + CQ_RS - 1) / CQ_RS;
    const bool pre = ps == w->ahead && !getenv("LSQ_QR_AHEAD_REDUNDANT");     // one factor kernel instead of one factor per workgroup
    const bool q1 = lsq_cqr_q1form();
    w->q1form = q1;
    const int lu_only = getenv("LSQ_QR_TOP_LU") ? 1 : 0;      // (A/B: the modified LU for every panel, as in rounds 2-5)
    if (hier && q1 && !pre) {
        // NO reduce launches (cq_group_reduce): the producers of the Gram partials leave group sums, the consumers add them
        const int ng = (nslab + CQ_GS - 1) / CQ_GS;
        if (!gram_ready)
            LSQ_LAUNCH(k_cqr_pass<0>, dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)nullptr, w->Gp,
                               w->R1, Vb, ldv, d_err, 0, 0, w->Gq1, w->gcnt);
        LSQ_LAUNCH(k_cqr_pass<1>, dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)w->Gq1, w->Gp,
                           w->R1, Vb, ldv, d_err, 1, ng, w->Gq2, w->gcnt, Vs);
        LSQ_HIP(hipGetLastError());
        LSQ_HIP(hipEventRecord(w->ev_q, ps));
        LSQ_HIP(hipStreamWaitEvent(w->side, w->ev_q, 0));
        LSQ_LAUNCH(k_cqr_top, dim3(1), dim3(256), CQ_LDS_LU, w->side, (const double *)w->Gq2, (const double *)w->R1,
                           (const double *)Vb, ldv, w->Binv, w->S, w->SR, d_err, w->R2inv, lu_only, ng);
        LSQ_HIP(hipGetLastError());
        LSQ_HIP(hipEventRecord(w->ev_lu, w->side));
        return LSQ_OK;
    }
    // (measured and not kept, round 6 late: group sums for pass 1's Gram ALONE -- no reduce launch on the side chain, k_cqr_top adds
    //  the <= 32 group sums -- 6.82 against 6.43 ms at C3: the last arrivers' sums at pass 1's tail cost more than the launch)
    // (gram_ready: the update of the previous panel left the Gram partials of this panel's slabs in w->Gp itself)
    if (!gram_ready)
        LSQ_LAUNCH(k_cqr_pass<0>, dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)nullptr, w->Gp,
                           w->R1, Vb, ldv, d_err, 0);
    LSQ_LAUNCH(k_cqr_reduce, dim3(256), dim3(256), 0, ps, (const double *)w->Gp, nslab, w->G);
    if (pre) {
        // (LSQ_QR_FACTOR_LDS: an LDS reservation beyond what the kernel uses -- with more than 160 - 84 KB it cannot share a CU
        //  with a workgroup of the look-ahead's update (84 KB reserved) and is placed on a free one)
        static const size_t flds = [] { const char *e = getenv("LSQ_QR_FACTOR_LDS"); return e ? (size_t)atoi(e) : (size_t)0; }();
        const size_t fl = std::max(CQ_LDS, flds);
        if (fl > CQ_LDS) LSQ_TRY(lsq_set_lds(c, (const void *)k_cqr_factor<1>, fl));
        LSQ_LAUNCH(k_cqr_factor<1>, dim3(1), dim3(256), fl, ps, (const double *)w->G, w->Minv, w->R1, d_err);
        LSQ_LAUNCH((k_cqr_pass<1, true>), dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)w->Minv, w->Gp,
                           w->R1, Vb, ldv, d_err, q1 ? 1 : 0, 0, (double *)nullptr, (unsigned *)nullptr, q1 ? Vs : nullptr);
    } else
        LSQ_LAUNCH(k_cqr_pass<1>, dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)w->G, w->Gp,
                           w->R1, Vb, ldv, d_err, q1 ? 1 : 0, 0, (double *)nullptr, (unsigned *)nullptr, q1 ? Vs : nullptr);
    if (q1) {
        // Q1 form: the panel is DONE on this stream -- Q1 is in Vb, the caller's V'[A2 | b] product may start.  The Gram reduce of
        // pass 1's partials and everything that hangs on G2 (R2, the 64-step LU of Q_top, inv(B), S R) run on the side stream
        // beside that product; k_cqr_tw_q1 waits for them (ev_lu).
        LSQ_HIP(hipGetLastError());
        LSQ_HIP(hipEventRecord(w->ev_q, ps));
        LSQ_HIP(hipStreamWaitEvent(w->side, w->ev_q, 0));
        LSQ_LAUNCH(k_cqr_reduce, dim3(256), dim3(256), 0, w->side, (const double *)w->Gp, nslab, w->G2);
        LSQ_LAUNCH(k_cqr_top, dim3(1), dim3(256), CQ_LDS_LU, w->side, (const double *)w->G2, (const double *)w->R1,
                           (const double *)Vb, ldv, w->Binv, w->S, w->SR, d_err, w->R2inv, lu_only, 0);
        LSQ_HIP(hipGetLastError());
        LSQ_HIP(hipEventRecord(w->ev_lu, w->side));
        return LSQ_OK;
    }
    LSQ_LAUNCH(k_cqr_reduce, dim3(256), dim3(256), 0, ps, (const double *)w->Gp, nslab, w->G2);
    LSQ_HIP(hipGetLastError());
    // everything that hangs on the top 64 rows (the 64-step LU among it) runs on the side stream from here on, beside
    // pass 2 and the caller's V'[A2 | b] product
    LSQ_HIP(hipEventRecord(w->ev_q, ps));
    LSQ_HIP(hipStreamWaitEvent(w->side, w->ev_q, 0));
    LSQ_LAUNCH(k_cqr_top, dim3(1), dim3(256), CQ_LDS_LU, w->side, (const double *)w->G2, (const double *)w->R1,
                       (const double *)(A + (size_t)c0 * M + c0), M, w->Binv, w->S, w->SR, d_err, (double *)nullptr, lu_only, 0);
    LSQ_HIP(hipEventRecord(w->ev_lu, w->side));
    if (pre) {
        LSQ_LAUNCH(k_cqr_factor<2>, dim3(1), dim3(256), CQ_LDS, ps, (const double *)w->G2, w->Minv + 4096, w->R1, d_err);
        LSQ_LAUNCH((k_cqr_pass<2, true>), dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)(w->Minv + 4096), w->Gp,
                           w->R1, Vb, ldv, d_err, 0);
    } else
        LSQ_LAUNCH(k_cqr_pass<2>, dim3(nslab), dim3(256), CQ_LDS, ps, A, M, c0, rows, (const double *)w->G2, w->Gp,
                           w->R1, Vb, ldv, d_err, 0);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_cqr_tw(lsq_ctx *c, CqrWork *w, const double *W, int ncolsB, double *A, int M, int c0, int cend, int n,
               double *rhs, double *Vb, int ldv, double *W2, double *W2s) {
    LSQ_HIP(hipStreamWaitEvent(c->stream, w->ev_lu, 0));
    const int ncols = ncolsB - 64;
    if (w->q1form) {
        LSQ_LAUNCH(k_cqr_tw_q1, dim3(std::max(1, (ncols + TQ_NC - 1) / TQ_NC) + 1), dim3(256), CQ_LDS_TW_Q1, c->stream, W, ncolsB,
                           (const double *)w->Binv, (const double *)w->S, (const double *)w->SR, (const double *)w->R2inv, A, M, c0, cend,
                           n, rhs, W2, W2s);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    LSQ_LAUNCH(k_cqr_tw, dim3(std::max(1, (ncols + 63) / 64) + 1), dim3(256), CQ_LDS_TW, c->stream, W, ncolsB,
                       (const double *)w->Binv, (const double *)w->S, (const double *)w->SR, A, M, c0, cend, n, rhs, Vb, ldv, W2);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
