// Stage 1 of the two-stage QR (lsq_qr.hip: dense_qr.jl:30-88): UNPIVOTED blocked Householder QR of [A | b] with 64-column
// panels -- CholeskyQR2 panels (lsq_qr_cholqr.hip) with the column-by-column Householder steps (k_qr1_step_multi) as the
// fallback for ill-conditioned panels and narrow / short operands, TSQR levels for tall-and-thin operands -- and the
// compact-WY block update on the fp64 MFMA unit (k_qr1_vtb, k_qr1_tw_mfma, k_qr1_update).
#include <type_traits>
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "lsq_qr_work.h"
#include "lsq_spmv.h"

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
// column r -- left UNSCALED by the round that finished it ("lazy" reflectors: sum v^2 and v'a share ONE block reduction; beta,
// tau and the scale 1/(alpha - beta) go to side arrays, the V materialisation applies the scale) -- and applies it to the later
// pivot columns and to the target, all in
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
template <int N, class F>
__device__ __forceinline__ void qr_static_for(F &&f) {
    if constexpr (N > 0) {
        qr_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
//
// TSQR (tall and thin operands, n <= K): every WAVEFRONT is on its own -- its slab of 64*RPT rows of ALL n columns and of b
// sits in registers, the K rounds factor the slab locally (pivot rows = the slab's first rows, no exchange: the reductions are
// wave reductions, the row-c elements come from their owner lane by v_readlane -- no LDS, no barrier anywhere in the rounds),
// and the only thing written is the slab's n x n triangle and the first n entries of its Q'b, stacked for the next level
// (tsq_S: (slabs*n) x n, tsq_r): ONE pass over the matrix.
template <int NT, int RPT, int K, int S, bool TSQR = false>
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
    constexpr int QS = TSQR ? 64 : NT;           // rows between two elements of a thread
    int g = (int)blockIdx.x, sidx = 0;
    if (TSQR) {                      // one wavefront per slab, a single "group" whose target is b
        g = 0;
        sidx = (int)blockIdx.x * NW + wv;
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
    const int t = i + sidx * (RPT * QS) + (TSQR ? lane : tid);    // row of element 0
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
        if constexpr (TSQR) {
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
        if constexpr (TSQR) {
        } else if (S == 1) {
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
                        if (++spins > QR1_SPIN_LIMIT) { atomicOr(err, 1); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); break; }
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
            if (lane == r) mybeta = beta;   // R(r, r) of this slab, kept by the owner of the slab's row r
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
        const int own = lane;
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
          const double *__restrict__ rhs, int ncolsB, int kslices, double *__restrict__ Wp, int tile0 /* first tile formed: 1
          skips V'V, which only the Householder-step panels' T factor needs */) {
    __shared__ double sA[Q2_NB * Q2_KS];
    __shared__ double sB[Q2_NB * Q2_KS];
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
    const int nt = ntile - tile0;
    const int tile = tile0 + blockIdx.x % nt, slice = blockIdx.x / nt;
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

// The same partial products with WAVE-PRIVATE tiles (round 5): no LDS, no barriers.  k_qr1_vtb parks its four waves at two
// barriers per 32-row slab (32 MFMAs per wave between them); here a wavefront owns 32 columns of B and all 64 of V over its
// k slice and feeds the MFMAs straight from global memory:
//   D'[B col j][V col i] = sum_rows B[row][j] V[row][i]:   a operand B[row][16 jt + ij], b operand V[row][16 it + ij],
//   lane (ij = lane & 15, kq = lane >> 4), step s of a 16-row chunk: row = chunk + 4 kq + s  (32 contiguous bytes per lane and
//   column: whole 128-byte lines per 4 lanes), result lane layout: B col 16 jt + kq + 4 r, V col 16 it + ij -- the partials
//   leave in 128-byte pieces.  Chunks of the next V3_D steps are in flight (register ring).  The four waves of a workgroup
//   take neighbouring 32-column groups over the SAME rows, so that three of their four reads of V hit the CU's L1; workgroups
//   of one k slice sit behind one L2 (slice = blockIdx % kslices).
// SWZ (round 6, late): the full chunks of V come from its fragment-order copy Vs (lsq_cqr_vs_index, written by pass 1 of the Q1
// form): eight loads of 1 KB contiguous per chunk instead of eight in which every lane reads 32 bytes of its own column -- a load
// costs the MFMA stream about two clocks per (4-lane group, cache line) pair it touches, 64 against 16 here
// (profiles/r06/ab_c3_vtb_overlap.txt, section 4).  The ragged last chunk still reads Vb.
constexpr int V3_D = 3;
template <int DBG, bool SWZ = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_qr1_vtb_w(const double *__restrict__ Vb, int ldv, const double *__restrict__ A, int M, int c0, int cend, int n,
            const double *__restrict__ rhs, int ncolsB, int kslices, double *__restrict__ Wp, int tile0,
            const double *__restrict__ Vs = nullptr) {
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
    const int rows = M - c0;
    const int slice = blockIdx.x % kslices, cgrp = blockIdx.x / kslices;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, ij = lane & 15, kq = lane >> 4;
    const int cb0 = tile0 * Q2_NB + cgrp * 128 + w * 32;          // first B column of this wave
    if (cb0 >= ncolsB) return;
    const int kper = ((rows + kslices - 1) / kslices + 15) / 16 * 16;
    const int kb = slice * kper, ke = min(rows, kb + kper);
    // column pointers of the lane: two of B, four of V (rows from c0)
    const int cbA = min(cb0 + ij, ncolsB - 1), cbB = min(cb0 + 16 + ij, ncolsB - 1);
    const double *pa0 = q2_bcol(Vb, ldv, A, M, c0, cend, n, rhs, cbA) + 4 * kq;
    const double *pa1 = q2_bcol(Vb, ldv, A, M, c0, cend, n, rhs, cbB) + 4 * kq;
    const double *pv = Vb + (size_t)ij * ldv + 4 * kq;
    const size_t vstep = (size_t)16 * ldv;
    v4d_qr acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d_qr){0.0, 0.0, 0.0, 0.0};
    auto fetch = [&](int R, double (*af)[4], double (*bf)[4]) {           // a full chunk
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            af[0][s] = pa0[R + s];
            af[1][s] = pa1[R + s];
        }
        if (SWZ) {
            const double *ps = Vs + lsq_cqr_vs_index(R >> 4, 0, 0, lane);        // (R is a multiple of 16: a chunk of the panel)
#pragma unroll
            for (int ih = 0; ih < 8; ++ih) {                                     // ih = 2 it + h
                const double2 v = *reinterpret_cast<const double2 *>(ps + ih * 128);
                bf[ih >> 1][2 * (ih & 1)] = v.x;
                bf[ih >> 1][2 * (ih & 1) + 1] = v.y;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[it][s] = pv[it * vstep + R + s];
    };
    auto chunk = [&](double (*af)[4], double (*bf)[4]) {
        if (DBG == 1) { acc[0][0][0] += af[0][0] * bf[0][0] + af[1][3] * bf[3][3]; return; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[jt][s], bf[it][s], acc[jt][it], 0, 0, 0);
    };
    double a0[2][4], b0[4][4], a1[2][4], b1[4][4], a2[2][4], b2[4][4];
    static_assert(V3_D == 3, "the ring below is written out for three stages");
    const int kfull = kb + ((max(ke - kb, 0)) & ~15);           // end of the full 16-row chunks
    if (kb < kfull) fetch(kb, a0, b0);
    if (kb + 16 < kfull) fetch(kb + 16, a1, b1);
    if (kb + 32 < kfull) fetch(kb + 32, a2, b2);
#define V3_STEP(R, AF, BF)                                     \
    {                                                          \
        if ((R) >= kfull) break;                               \
        chunk(AF, BF);                                         \
        if ((R) + 16 * V3_D < kfull) fetch((R) + 16 * V3_D, AF, BF); \
    }
    for (int R = kb; R < kfull; R += 16 * V3_D) {
        V3_STEP(R, a0, b0)
        V3_STEP(R + 16, a1, b1)
        V3_STEP(R + 32, a2, b2)
    }
#undef V3_STEP
    if (kfull < ke) {                              // the ragged last chunk: clamped addresses, zeroed by selection
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = kfull + 4 * kq + s, rc = min(r, ke - 1) - 4 * kq;
            const double x0 = pa0[rc], x1 = pa1[rc];
            a0[0][s] = r < ke ? x0 : 0.0;
            a0[1][s] = r < ke ? x1 : 0.0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const double y = pv[it * vstep + rc];
                b0[it][s] = r < ke ? y : 0.0;
            }
        }
        chunk(a0, b0);
    }
    // partials: Wp[slice][tile][B col in tile][V col]
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cb = cb0 + 16 * jt + kq + 4 * r;
            if (cb < ntile * Q2_NB) {
                double *out = Wp + ((size_t)slice * ntile + cb / Q2_NB) * (Q2_NB * Q2_NB) + (size_t)(cb % Q2_NB) * Q2_NB + ij;
#pragma unroll
                for (int it = 0; it < 4; ++it) out[16 * it] = cb < ncolsB ? acc[jt][it][r] : 0.0;
            }
        }
}

// W[cb][0:64] = sum over slices (fixed order)
__global__ void __launch_bounds__(256)
k_qr1_wreduce(const double *__restrict__ Wp, int ncolsB, int kslices, double *__restrict__ W, int tile0 /* first tile formed by k_qr1_vtb */) {
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
    const long long tot = (long long)ntile * Q2_NB * Q2_NB;
    for (long long e = (long long)tile0 * Q2_NB * Q2_NB + blockIdx.x * 256LL + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        double s = 0.0;
        int sl = 0;
        for (; sl + 8 <= kslices; sl += 8) {           // eight loads in flight, added in slice order
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = Wp[(size_t)(sl + u) * tot + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; sl < kslices; ++sl) s += Wp[(size_t)sl * tot + e];
        W[e] = s;   // layout [tile][col][row] == [cb][row]
    }
}

// W2 = T' W for the columns of A2 and b without forming T: for H_0 ... H_{nb-1} = I - V T V' the inverse
// of T is the upper triangle of the Gram block G = V'V with 1/tau on the diagonal (tau_j = 2 / v_j'v_j),
// so T' W = W2 solves the unit-structured lower-triangular system
//      W2[j] = tau_j * (W[j] - sum_{k<j} G[j][k] W2[k])
// (tau_j = 0, i.e. H_j = I, gives W2[j] = 0 as dlarft's zero column does).
// The 64-step recurrence is taken off the critical path: it reads
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

// The same update with WAVE-PRIVATE tiles (round 5): no LDS, no barriers.  At NB = 64 the block update is 8 flops per byte of
// A2 -- the fp64 MFMA issue ceiling (48 TFLOP/s) and the streaming rate (6 TB/s) meet.  Measured on this kernel (C3, average
// over the 32 panels): A2 traffic alone 39.4 us (6.2 TB/s), MFMAs alone 44.5 us, k_qr1_update (LDS-staged, 4 barriers per
// 64-column tile) 68.6 us.  Forms tried: one tile of prefetch 67 us (one tile's MFMAs, 2 us, do not cover a loaded memory
// round trip); V in LDS 80 us (a ds_read in front of every MFMA: MFMAs alone 68 us); this one 58 us:
//  * a wavefront owns 32 rows x 16 columns per tile and computes the TRANSPOSED product (V W2)' = W2' V': the MFMA result
//    layout then has a lane's 16 neighbours on 16 consecutive rows of one column, and the read-modify-write of A2 goes
//    straight from / to registers in 128-byte pieces -- no product image in LDS;
//  * its V fragment (32 x 64) stays in registers for all its tiles; the A2 tiles of the next U3_DA tiles are in flight
//    (register ring), W2 fragments (L2-resident, 32-byte pieces per lane, k index permuted alike in both operands) one
//    tile ahead; 242 VGPRs, two waves per SIMD (held to three the ring spills);
//  * panels 0-5 of C3 stay at 4 TB/s: their trailing matrix (> 230 MB) does not fit the 256 MB MALL between k_qr1_vtb's
//    read and this kernel's; from there on 5 TB/s.
//   lane (ij = lane & 15, kq = lane >> 4), step s: k = 16 (s >> 2) + 4 kq + (s & 3)
//   a operand  W2[col jj + ij][k]      b operand  V[row r0 + 16 b + ij][k]      D'[col jj + kq + 4 r][row r0 + 16 b + ij]
constexpr int U3_DA = 4;                    // depth of the A2 ring
typedef double u3_d2 __attribute__((ext_vector_type(2), aligned(8)));     // two adjacent rows of a column (8-byte aligned: M may be odd)
// GRAM (round 6): the launch carries nrg extra workgroups AT ITS FRONT that update the NEXT panel's 64 columns (trailing columns
// 0..63, whatever jbeg says) and, with the updated 64 x 64 block still at hand, form the Gram partial of that slab for the next
// panel's CholeskyQR pass (what k_cqr_pass<0> would read back from memory one launch later: same values, same cq_slab_gram,
// same bits) -- row group rg of this panel is slab rg - 1 of the next (its first 64 rows become rows of R).  The other
// workgroups take columns jbeg..jend as before (the caller passes jbeg >= 64).
// SWZ: W2 is read from its fragment-order copy (lsq_cqr_w2s_index: 1 KB contiguous per load instruction instead of 64 pieces of
// 32 bytes -- the pieces cost the MFMA stream four times as many clocks, lsq_qr_cholqr.h); W2 then points at that copy.
template <int DBG, bool GRAM = false, bool SWZ = false>     // DBG 0: the product; 1: no MFMAs (memory side alone); 2: no A2 traffic (MFMA side alone) -- timing experiments only
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_qr1_update_w(const double *__restrict__ Vb, int ldv, double *__restrict__ A, int M, int c0, int cend, int n,
               double *__restrict__ rhs, int ncols /* n - cend + 1 */, const double *__restrict__ W2, int jbeg, int jend, int tpw,
               double *gram_out = nullptr, double *gram_q = nullptr /* group sums too (cq_group_reduce) */, unsigned *gram_cnt = nullptr,
               int flat_g = 0 /* > 0: that many workgroups share the (row group, tile pair) units evenly, see qr1_update_wave */) {
    extern __shared__ __attribute__((aligned(16))) double u3_qs[];      // GRAM: the slab image [col][row], CQ_QST apart
    const int rows = M - c0;
    const int nrg = (rows + Q2_NB - 1) / Q2_NB;         // row groups of 64 rows: waves 0,1 the upper half, 2,3 the lower
    int bid = blockIdx.x;
    bool narrow = false;
    if (GRAM) {
        narrow = bid < nrg;
        if (!narrow) bid -= nrg;
        else { jbeg = 0; jend = Q2_NB; tpw = 2; flat_g = 0; }
    }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncolsA = n - cend;                        // columns of A proper; column ncolsA is the right-hand side
    if (GRAM && narrow) {
        // rows past the end of the matrix (the ragged last slab) count as zeros; every wavefront stays for the Gram
        for (int e = tid; e < Q2_NB * CQ_QST; e += 256) u3_qs[e] = 0.0;
        __syncthreads();
    }
    // SEGMENTS: a workgroup takes tile pairs [p0, pe) of one row group at a time.  Legacy mapping (flat_g == 0): one segment,
    // row group bid % nrg, pairs of column group bid / nrg.  Flat mapping: the nrg x npair units in row-group-major order are
    // cut into flat_g equal runs; a run that crosses into the next row group reloads the V fragment there.
    const int npair = (max(jend - jbeg, 0) + 31) >> 5;
    long long u = 0, uend = 0;
    if (flat_g > 0) {
        const long long U = (long long)nrg * npair;
        u = U * bid / flat_g;
        uend = U * (bid + 1) / flat_g;
    }
    for (bool first = true;; first = false) {
    // (the lane's coordinates are re-derived per segment behind an opaque move: addresses hoisted out of this loop would cost
    //  the 14 registers the kernel has to spare -- 23-37 VGPRs spilled without it)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int ij = lane_o & 15, kq = lane_o >> 4;
    int rg, p0, pe;
    if (flat_g > 0) {
        if (u >= uend) break;
        rg = (int)(u / npair);
        p0 = (int)(u - (long long)rg * npair);
        pe = (int)min((long long)npair, p0 + (uend - u));
        u += pe - p0;
    } else {
        if (!first) break;
        rg = bid % nrg;
        p0 = (bid / nrg) * tpw;
        pe = min(npair, p0 + tpw);
    }
    const int rbase = rg * Q2_NB;
    const int r0 = rbase + 32 * (w >> 1);
    const bool rfull = r0 + 32 <= rows;
    // (round 6, late) the wave's two 16-row MFMA tiles are the EVEN and the ODD rows of its 32: lane ij holds rows r0 + 2 ij and
    // r0 + 2 ij + 1, adjacent in memory, so V and A2 travel as 16-byte pieces -- half the load / store instructions, each still
    // one cache line per 4-lane group (a load or store costs the MFMA stream ~2 clocks per (lane group, line) pair).  Which
    // physical row an MFMA row is changes nothing about an element's dot product: same bits as rows r0 + ij / r0 + 16 + ij.
    const int row0 = r0 + 2 * ij, row1 = row0 + 1;
    const bool in0 = row0 < rows, in1 = row1 < rows;
    // tiles of this wave: jfirst + 32 t, t = 0 .. nt - 1 (the two waves of a row half take alternate 16-column tiles)
    const int jfirst = jbeg + 32 * p0 + 16 * (w & 1);
    const int jstop = min(jend, jbeg + 32 * pe);
    const bool idle = r0 >= rows || jfirst >= jstop;
    if (idle) continue;
    const int nt = (jstop - jfirst + 31) >> 5;
    double v0[16], v1[16];                              // the wave's V fragment: 32 rows x 64, for all its tiles
    {
        if (rfull) {
            const double *pv2 = Vb + row0;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const size_t k = 16 * (s >> 2) + 4 * kq + (s & 3);
                const u3_d2 x = *reinterpret_cast<const u3_d2 *>(pv2 + k * ldv);
                v0[s] = x[0];
                v1[s] = x[1];
            }
        } else {
        const double *p0v = Vb + (in0 ? row0 : 0), *p1v = Vb + (in1 ? row1 : 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const size_t k = 16 * (s >> 2) + 4 * kq + (s & 3);
            const double x0 = p0v[k * ldv], x1 = p1v[k * ldv];
            v0[s] = in0 ? x0 : 0.0;
            v1[s] = in1 ? x1 : 0.0;
        }
        }
    }
    auto colptr = [&](int col) -> double * { return (col < ncolsA ? A + (size_t)(cend + col) * M : rhs) + c0; };
    auto fetch_w = [&](int t, double *wf) {
        if (SWZ) {
            const double *pw = W2 + (size_t)((jfirst + 32 * t) >> 4) * 1024 + (size_t)lane_o * 2;
#pragma unroll
            for (int gh = 0; gh < 8; ++gh) {          // gh = 2 g + h: k = 16 g + 4 kq + 2 h + (0, 1) = wf index 4 g + 2 h + (0, 1)
                const double2 v = *reinterpret_cast<const double2 *>(pw + gh * 128);
                wf[2 * gh] = v.x;
                wf[2 * gh + 1] = v.y;
            }
            return;
        }
        const int col = min(jfirst + 32 * t + ij, ncols - 1);
        const double *pw = W2 + (size_t)col * Q2_NB + 4 * kq;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) wf[4 * g + u] = pw[16 * g + u];
    };
    auto fetch_a = [&](int t, double *at) {
        const int jj = jfirst + 32 * t;
        if (DBG == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) at[r] = 0.0;
        } else if (rfull && jj + 16 <= ncolsA) {
            const double *pa = A + (size_t)(cend + jj + kq) * M + c0 + row0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const u3_d2 x = *reinterpret_cast<const u3_d2 *>(pa + (size_t)4 * r * M);
                at[r] = x[0];
                at[4 + r] = x[1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = jj + kq + 4 * r;
                const double *pc = colptr(min(col, ncols - 1));
                const double x0 = pc[in0 ? row0 : 0], x1 = pc[in1 ? row1 : 0];
                at[r] = x0;
                at[4 + r] = x1;
            }
        }
    };
    auto tile = [&](int t, const double *wf, const double *at) {
        const int jj = jfirst + 32 * t;
        v4d_qr c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;       // [row tile][k half]: four independent chains
        if (DBG == 1) { c00[0] = wf[0] * v0[0]; c10[0] = wf[1] * v1[1]; }
        else
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s], v0[s], c00, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s], v1[s], c10, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s + 1], v0[s + 1], c01, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s + 1], v1[s + 1], c11, 0, 0, 0);
        }
        if (DBG == 2) {
            double sacc = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc += (c00[r] + c01[r]) + (c10[r] + c11[r]);
            if (sacc == 1.2345e300) rhs[0] = sacc;      // (keeps the products alive)
        } else if (rfull && jj + 16 <= ncolsA) {
            double *pa = A + (size_t)(cend + jj + kq) * M + c0 + row0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double n0 = at[r] - (c00[r] + c01[r]), n1 = at[4 + r] - (c10[r] + c11[r]);
                u3_d2 y;
                y[0] = n0;
                y[1] = n1;
                *reinterpret_cast<u3_d2 *>(pa + (size_t)4 * r * M) = y;
                if (GRAM && narrow) {       // (column jj + kq + 4 r of the next panel, rows of this slab)
                    double *qs = u3_qs + (jj + kq + 4 * r) * CQ_QST + 32 * (w >> 1) + 2 * ij;
                    qs[0] = n0;
                    qs[1] = n1;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = jj + kq + 4 * r;
                if (col < ncols) {
                    double *pc = colptr(col);
                    const double n0 = at[r] - (c00[r] + c01[r]), n1 = at[4 + r] - (c10[r] + c11[r]);
                    if (in0) pc[row0] = n0;
                    if (in1) pc[row1] = n1;
                    if (GRAM && narrow && col < Q2_NB) {
                        double *qs = u3_qs + col * CQ_QST + 32 * (w >> 1) + 2 * ij;
                        if (in0) qs[0] = n0;
                        if (in1) qs[1] = n1;
                    }
                }
            }
        }
    };
    double wa[16], wb[16], a0[8], a1[8], a2[8], a3[8];
    static_assert(U3_DA == 4, "the ring below is written out for four stages");
    if (0 < nt) {
        fetch_w(0, wa);
        fetch_a(0, a0);
    }
    if (1 < nt) fetch_a(1, a1);
    if (2 < nt) fetch_a(2, a2);
    if (3 < nt) fetch_a(3, a3);
    // one step: W2 of tile t + 1 requested, tile t multiplied and written back, its ring slot refilled with tile t + U3_DA
#define U3_STEP(T, WCUR, WNXT, ACUR)                          \
    {                                                         \
        if ((T) >= nt) break;                                 \
        if ((T) + 1 < nt) fetch_w((T) + 1, WNXT);             \
        tile((T), WCUR, ACUR);                                \
        if ((T) + U3_DA < nt) fetch_a((T) + U3_DA, ACUR);     \
    }
    for (int t = 0; t < nt; t += 4) {
        U3_STEP(t, wa, wb, a0)
        U3_STEP(t + 1, wb, wa, a1)
        U3_STEP(t + 2, wa, wb, a2)
        U3_STEP(t + 3, wb, wa, a3)
    }
#undef U3_STEP
    }   // segments
    if (GRAM && narrow) {
        const int rg = bid;
        __syncthreads();                               // the slab's 64 x 64 block of the next panel is in the image
        if (rg >= 1 && gram_out) {                     // (uniform over the workgroup)
            if (gram_q) {
                cq_slab_gram<true>(u3_qs, gram_out + (size_t)(rg - 1) * 4096, tid);
                cq_group_reduce(gram_out, gram_q, gram_cnt, rg - 1, nrg - 1, tid);
            } else
                cq_slab_gram(u3_qs, gram_out + (size_t)(rg - 1) * 4096, tid);
        }
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

// launches the wave-private update (default since round 5; LSQ_QR_UPDATE_W=0: false, the caller launches k_qr1_update; 2 / 3: the
// timing experiments DBG 1 / 2 -- wrong results).  All read per call.
// gram_out (or null): the launch also updates trailing columns 0..63 -- the next panel -- in nrg workgroups of their own and leaves
// the Gram partials of the next panel's slabs there (k_qr1_update_w<0, true>); the caller then passes jbeg >= 64 (jbeg == jend:
// those workgroups alone).
static int qr1_update_wave(lsq_ctx *c, const double *Vb, int ldv, double *A, int M, int c0, int cend, int n, double *rhs,
                           int ncols, const double *W2, bool *taken, int jbeg = 0, int jend = -1, bool beside_passes = false,
                           double *gram_out = nullptr, double *gram_q = nullptr, unsigned *gram_cnt = nullptr,
                           const double *W2s = nullptr /* W2 in fragment order too (lsq_cqr_w2s_index): the kernel reads that copy */) {
    const char *e = getenv("LSQ_QR_UPDATE_W");
    const int mode = e ? atoi(e) : 1;
    *taken = mode != 0;
    if (mode == 0) return LSQ_OK;
    if (jend < 0) jend = ncols;
    if (jbeg >= jend && !gram_out) return LSQ_OK;
    const int rows = M - c0, nrg = (rows + Q2_NB - 1) / Q2_NB;
    // 32-column tile pairs per workgroup: as many as leave one workgroup per CU, at most 32 -- a wave's V fragment then serves
    // many tiles (measured at C3, all panels: 32 pairs 57.8 us, 16 58.9, 8 61.0, 4 62.3; a rule that keeps two workgroups per
    // CU 60.1)
    const char *t = getenv("LSQ_QR_UPDATE_TPW");
    const int npair = (std::max(jend - jbeg, 0) + 31) / 32, want = std::max(1, c->num_cus / nrg);
    int tpw = t ? std::max(1, atoi(t)) : std::max(2, std::min(beside_passes ? 64 : 32, (npair + want - 1) / want));
    int ncg = (npair + tpw - 1) / tpw;
    // beside the next panel's passes (look-ahead): ONE workgroup per CU, by an LDS reservation it does not use -- two of them
    // fill a CU's registers (242 VGPRs per wave) and the pass workgroups (75 KB of LDS, 4 waves) would wait for them to end
    const char *le = getenv("LSQ_QR_UPDATE_LDS");
    size_t lds = beside_passes ? (le ? (size_t)atoi(le) : (size_t)84 * 1024) : 0;
    if (gram_out) lds = std::max(lds, (size_t)Q2_NB * CQ_QST * sizeof(double));
    // FLAT MAPPING (round 6, later): with one workgroup per (row group, column group) the grid is a multiple of nrg, and once
    // nrg does not divide the CUs some CUs run a workgroup more than the others -- at 18432 rows (288 row groups, 576
    // workgroups of 31 pairs on 256 CUs: 3 against 2) the launch took 94 us where 16384 rows take 62, 1.52 x for 1.125 x the
    // work; 200 row groups x 40 pairs became 200 workgroups of 32 pairs and 200 of 8.  In the flat mapping num_cus (or
    // 2 num_cus, beyond 32 pairs each) workgroups take equal runs of the nrg x npair units in row-group-major order; a run that
    // crosses a row group starts its pipeline again (about 1.5 pairs' time), so the rectangular grid stays where it is even:
    // both are priced in pair times -- a CU's rate does not depend on how many workgroups it holds (the 1.52 above), a
    // workgroup costs half a pair on top of its tiles (32 / 16 / 8 / 4 pairs: 57.8 / 58.9 / 61.0 / 62.3 us) -- and flat must
    // win by 3 %.  Measured (ms, rectangular rule of round 5 -> this): 12000 x 2048 6.20 -> 5.97, 18432 x 2048 7.81 -> 7.42,
    // 24000 x 2048 8.86 -> 8.56.  LSQ_QR_UPDATE_FLAT=0: round 5's grid; 1 / 2: flat with that many workgroups per CU.
    const char *fe = getenv("LSQ_QR_UPDATE_FLAT");
    const long long units = (long long)nrg * npair;
    int flat_g = 0;
    if (!(fe && atoi(fe) == 0) && !t && units > 0) {
        const int cap = beside_passes ? 64 : 32;
        double t_rect = 1e30;
        int tpw_rect = tpw;
        for (int g = 1; g <= npair; ++g) {
            const int tp = (npair + g - 1) / g;
            if (tp > cap) continue;
            if (tp < 2 && g > 1) break;
            const double tt = (double)(((long long)nrg * ((npair + tp - 1) / tp) + c->num_cus - 1) / c->num_cus) * (tp + 0.5);
            if (tt < t_rect) { t_rect = tt; tpw_rect = tp; }
        }
        const int per_cu = fe ? std::max(1, atoi(fe)) : (beside_passes || units <= (long long)cap * c->num_cus ? 1 : 2);
        const double t_flat = (double)units / c->num_cus + 0.5 * per_cu + 1.5;
        // (beside the next panel's passes: LSQ_QR_UPDATE_SPARE CUs, default 2, are left without a workgroup of this launch --
        //  the look-ahead's one-workgroup kernels find a quieter CU: 7.41 -> 7.34 ms on LM's stacked C3 operand, 24000 x 2048
        //  8.59 -> 8.46; forcing the factor kernel onto a FREE CU by an LDS reservation, LSQ_QR_FACTOR_LDS=102400, loses 3 %;
        //  profiles/r06/ab_c3_lookahead_spare_cus.txt)
        static const int spare = [] { const char *e = getenv("LSQ_QR_UPDATE_SPARE"); return e ? std::max(0, atoi(e)) : 2; }();
        const int sp = beside_passes ? std::min(spare, c->num_cus / 2) : 0;
        if (fe || sp > 0 || t_flat < 0.97 * t_rect)
            flat_g = (int)std::max(1LL, std::min((long long)per_cu * c->num_cus - sp, units / 2));
        else { tpw = tpw_rect; ncg = (npair + tpw - 1) / tpw; }
    }
    const int grid = (flat_g > 0 ? flat_g : units > 0 ? nrg * ncg : 0) + (gram_out ? nrg : 0);
    auto go = [&](auto kern) -> int {
        if (lds > 48 * 1024) LSQ_TRY(lsq_set_lds(c, (const void *)kern, lds));
        LSQ_LAUNCH(kern, dim3(grid), dim3(256), lds, c->stream, Vb, ldv, A, M, c0, cend, n, rhs, ncols, W2, jbeg, jend, tpw, gram_out,
                   gram_q, gram_cnt, flat_g);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    };
    if (W2s && mode == 1) {
        W2 = W2s;
        if (gram_out) return go(k_qr1_update_w<0, true, true>);
        return go(k_qr1_update_w<0, false, true>);
    }
    if (gram_out) return go(k_qr1_update_w<0, true>);
    if (mode == 2) return go(k_qr1_update_w<1>);
    if (mode == 3) return go(k_qr1_update_w<2>);
    return go(k_qr1_update_w<0>);
}

static void qr2_free(void *p) {
    Qr2Work *q = (Qr2Work *)p;
    if (!q) return;
    hipFree(q->Vb); hipFree(q->Vb2); hipFree(q->Vs); hipFree(q->Vs2); hipFree(q->Wp); hipFree(q->W); hipFree(q->W2); hipFree(q->W2s); hipFree(q->R); hipFree(q->rhs2); hipFree(q->tau1);
    hipFree(q->vn); hipFree(q->colat); hipFree(q->ice); hipFree(q->lazy);
    hipFree(q->Xinv); hipFree(q->T2); hipFree(q->fro); hipFree(q->xslot); hipFree(q->bslot); hipFree(q->d_err); hipFree(q->Pn); hipFree(q->tsS[0]); hipFree(q->tsS[1]); hipFree(q->tsr[0]); hipFree(q->tsr[1]);
    if (q->h_fro) hipHostFree(q->h_fro);
    lsq_cqr_free(&q->cq);
    delete q;
}

// k slices of k_qr1_vtb for a panel with `nt` tiles to form out of `ntile` (the partials are indexed slice * ntile + tile):
// about four workgroups per CU whatever the width of the trailing matrix.  (Until round 5 the count was fixed by the WIDEST
// panel -- 31 at C3 -- so that the last panels ran 62 workgroups of 15 dependent slabs each: 33 us for 1 % of the flops.)
static int qr1_vtb_slices(const lsq_ctx *c, const Qr2Work *q, int rows, int nt, int ntile) {
    const char *e = getenv("LSQ_QR_VTB_KSMAX");
    const int ksmax = e ? atoi(e) : 64;
    // (rounded DOWN, and four slots kept free: k_cqr_top holds a CU's LDS on the side stream while this grid runs -- a
    //  workgroup over the resident 4 per CU waits for a whole second round: 127 against 102 us at 30 tiles x 35 slices)
    int ks = ksmax > 0 ? std::min(ksmax, (4 * c->num_cus - 4) / std::max(1, nt)) : q->kslices;
    ks = std::min(ks, q->wp_slots / std::max(1, ntile));
    ks = std::min(ks, (rows + 4 * Q2_KC - 1) / (4 * Q2_KC));
    return std::max(1, ks);
}

// V'[A2 | b] partials by the wave-private kernel (default; LSQ_QR_VTB_W=0: k_qr1_vtb; 2: timing experiment without MFMAs).
// Returns the number of k slices written (for k_qr1_wreduce), or -1 if the launch failed.
static int qr1_vtb_launch(lsq_ctx *c, Qr2Work *q, const double *Vb, int ldv, const double *A, int M, int c0, int cend, int n,
                          const double *rhs, int ncolsB, int tile0, const double *Vs = nullptr /* V in fragment order too */) {
    const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB, rows = M - c0;
    const char *e = getenv("LSQ_QR_VTB_W");
    const int mode = e ? atoi(e) : 1;
    if (mode == 0) {
        const int ks = qr1_vtb_slices(c, q, rows, ntile - tile0, ntile);
        LSQ_LAUNCH(k_qr1_vtb, dim3((ntile - tile0) * ks), dim3(256), 0, c->stream, Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks,
                           q->Wp, tile0);
        return hipGetLastError() == hipSuccess ? ks : -1;
    }
    // workgroups of 4 waves x 32 columns; ONE per CU (measured at C3, average over the panels: 56.2 us; two per CU -- what
    // the registers allow -- 59.0; four queued 71.1; k_qr1_vtb 64.0)
    const int ncgrp = (ncolsB - tile0 * Q2_NB + 127) / 128;
    const char *k = getenv("LSQ_QR_VTB_WGS");
    const int per_cu = k ? std::max(1, atoi(k)) : 1;
    int ks = std::max(1, (per_cu * c->num_cus - 2) / std::max(1, ncgrp));
    ks = std::min(ks, 128);
    ks = std::min(ks, q->wp_slots / std::max(1, ntile));
    ks = std::max(1, std::min(ks, (rows + 63) / 64));
    // An LDS reservation the kernel does not use (round 6): with more than half a CU's LDS per workgroup there is at most ONE of
    // these workgroups on a CU, so the grid (<= CUs - 2 workgroups) leaves whole CUs free -- and k_cqr_top (141 KB of LDS, on
    // the high-priority side stream beside this product) can only be placed on one of THOSE: it runs alone on its CU instead of
    // sharing issue slots with a product workgroup (measured: 47 us alone, 78 us beside one).
    const char *le = getenv("LSQ_QR_VTB_LDS");
    const size_t lds = per_cu == 1 ? (le ? (size_t)atoi(le) : (size_t)84 * 1024) : 0;
    if (mode == 2) {
        if (lds > 48 * 1024 && lsq_set_lds(c, (const void *)k_qr1_vtb_w<1>, lds) != LSQ_OK) return -1;
        LSQ_LAUNCH(k_qr1_vtb_w<1>, dim3(ncgrp * ks), dim3(256), lds, c->stream, Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks, q->Wp, tile0);
    } else if (Vs) {
        if (lds > 48 * 1024 && lsq_set_lds(c, (const void *)k_qr1_vtb_w<0, true>, lds) != LSQ_OK) return -1;
        LSQ_LAUNCH((k_qr1_vtb_w<0, true>), dim3(ncgrp * ks), dim3(256), lds, c->stream, Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks, q->Wp,
                   tile0, Vs);
    } else {
        if (lds > 48 * 1024 && lsq_set_lds(c, (const void *)k_qr1_vtb_w<0>, lds) != LSQ_OK) return -1;
        LSQ_LAUNCH(k_qr1_vtb_w<0>, dim3(ncgrp * ks), dim3(256), lds, c->stream, Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, ks, q->Wp, tile0);
    }
    return hipGetLastError() == hipSuccess ? ks : -1;
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
    return M >= 32768;
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
                LSQ_LAUNCH(kern, dim3(lsq_div_up(S, slabs_per_block)), dim3(256), 0, c->stream, Acur, Mcur, n, 0, n, q->tau1,
                                   q->lazy, q->lazy + n, S, q->xslot, ++q->epoch, q->d_err, q->Pn, 0, bcur, So, Ms, ro);
            };
            if (n <= 8) go(k_qr1_step_multi<256, 8, 8, 1, true>, 4);
            else if (n <= 12) go(k_qr1_step_multi<256, 6, 12, 1, true>, 4);
            else if (n <= 16) go(k_qr1_step_multi<256, 4, 16, 1, true>, 4);
            else if (n <= 20) go(k_qr1_step_multi<256, 3, 20, 1, true>, 4);
            else if (n <= 24) go(k_qr1_step_multi<256, 3, 24, 1, true>, 4);
            else if (n <= 28) go(k_qr1_step_multi<256, 2, 28, 1, true>, 4);
            else go(k_qr1_step_multi<256, 2, 32, 1, true>, 4);
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
        LSQ_HIP(hipMalloc(&q->Vs, ((size_t)(M + 16) * Q2_NB + 64) * sizeof(double)));
        // split-K partials of V'[A2 | b]: room for the widest panel at q->kslices slices AND for the narrow last panels at many
        // more slices each (qr1_vtb_slices)
        q->wp_slots = std::max(q->kslices * ntile, 4 * c->num_cus + 2 * ntile + 256);
        LSQ_HIP(hipMalloc(&q->Wp, (size_t)q->wp_slots * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->W, (size_t)ntile * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->W2, (size_t)ntile * Q2_NB * Q2_NB * sizeof(double)));
        LSQ_HIP(hipMalloc(&q->W2s, ((size_t)ntile * Q2_NB * Q2_NB + 1024) * sizeof(double)));
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
        q->no_exchange = s->fb_qrx.off();
        q->no_cholqr = s->fb_cholqr.off();
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
    // LOOK-AHEAD (round 5): once panel k's W2 is known, the NEXT panel's 64 columns are updated first and its passes run on
    // their own stream beside the update of the other trailing columns; its Q goes to the second V buffer.  (Round 3 built
    // this on LDS-staged update kernels: pass and update workgroups fought for the CUs' LDS, both stretched, 8.0 against
    // 7.55 ms.  The wave-private update holds no LDS.)  Only with the wave-private update (it takes a column range).
    const char *lae = getenv("LSQ_QR_LOOKAHEAD");
    const char *uwe = getenv("LSQ_QR_UPDATE_W");
    // Round 6: in the Q1 form of the panel (one pass, the rest on the side stream) the chain that the look-ahead would hide is
    // already short and runs beside the V'[A2 | b] product; measured with it 6.82 / 7.91 ms (C3 / LM's stacked operand), without
    // 6.77 / 7.92 -- so it is taken only where asked for (LSQ_QR_LOOKAHEAD=1) or in the three-pass form (LSQ_QR_CQR_PASS2=1).
    // ... EXCEPT for panels with more 64-row slabs than the device has CUs (round 6, later): their passes no longer fit one round
    // of workgroups, the chain grows by a half and there it pays to hide it.  With every panel factored ahead (no minimum of
    // other columns) 20000 x 1000 takes 3.04 instead of 3.32 ms, 40000 x 512 1.76 / 1.94, 24000 x 2048 8.81 / 9.35, LM's stacked
    // 18432 x 2048 operand 7.80 / 8.32 -- while C3's 16384 rows (256 slabs) are a wash (6.76 / 6.88) and 4096 x 512, 3000 x 700,
    // 8192 x 1024 lose 5-10 % (profiles/r06/ab_c3_lookahead_tall.txt).  So in the Q1 form, unless LSQ_QR_LOOKAHEAD says otherwise,
    // a panel is factored ahead iff it has more than num_cus * CQ_RS rows.
    const int la_q1_min_rows = c->num_cus * CQ_RS + 1;
    const bool la_auto = !lae && lsq_cqr_q1form();
    const bool la_on = cq_ok && (lae ? atoi(lae) != 0 : true) && (uwe ? atoi(uwe) == 1 : true);
    // worth it while the update of the other columns outlasts most of the passes (which run 1.5-2x slower beside it):
    // measured at C3 7.20 -> 7.06 ms, 18432 x 2048 8.94 -> 8.22; 4096 x 512 and 3000 x 700 lose 3-5 % with it
    const char *lmc = getenv("LSQ_QR_LOOKAHEAD_MINCOLS");
    const int la_min_cols = lmc ? atoi(lmc) : (la_auto ? 0 : 1024);
    const bool swz_on = !getenv("LSQ_QR_NO_SWIZZLE");
    bool pre = false;                        // this panel was factored ahead (its Q is in vcur, ev_panel says when)
    // FUSED GRAM (round 6): the update of panel k forms the Gram partials of panel k + 1 (its first pass, k_cqr_pass<0>, is one
    // launch less on every panel's chain); only the wave-private update does it (mode 1, no timing experiment)
    const bool gram_on = cq_ok && !getenv("LSQ_QR_NO_FUSED_GRAM") && (uwe ? atoi(uwe) == 1 : true);
    bool gram_ready = false;                 // ... and did so for the panel at hand
    // GROUP-LEVEL GRAM SUMS (round 6, cq_group_reduce: no reduce launches; Q1 form without look-ahead) -- measured slower,
    // LSQ_QR_HIER=1 only (lsq_cqr_hier)
    const bool hier_on = cq_ok && (!la_on || la_auto) && getenv("LSQ_QR_HIER") != nullptr;   // (panels not factored ahead)
    bool hier_ready = false;                 // the partials at hand came with their group sums
    double *vcur = q->Vb;
    double *vscur = nullptr;                 // vcur's fragment-order copy (Q1 form), or null
    for (int c0 = 0; c0 < n; c0 += Q2_NB) {
        const int nb = std::min(Q2_NB, n - c0), cend = c0 + nb;
        if (cq_ok && nb == Q2_NB && M - c0 >= 256) {
            // panel: two Gram / Cholesky passes, no column-by-column chain; block reflector in basis-kernel form
            if (!q->cq.ready) LSQ_TRY(lsq_cqr_alloc(c, &q->cq, q->M));
            q->cholqr_used = true;
            const int rows = M - c0, ldv = rows;
            if (pre) LSQ_HIP(hipStreamWaitEvent(c->stream, q->cq.ev_panel, 0));
            else {
                vcur = q->Vb;
                vscur = swz_on && lsq_cqr_q1form() ? q->Vs : nullptr;
                const bool hier = hier_on && lsq_cqr_hier((rows + CQ_RS - 1) / CQ_RS) && (!gram_ready || hier_ready);
                LSQ_TRY(lsq_cqr_panel(c, &q->cq, A, M, c0, vcur, ldv, q->d_err, c->stream, gram_ready, hier, vscur));
            }
            pre = false;
            gram_ready = false;
            hier_ready = false;
            const int ncols = n - cend + 1, ncolsB = Q2_NB + ncols;
            const int ntile = (ncolsB + Q2_NB - 1) / Q2_NB;
            // (V'V, tile 0, is not formed: the basis-kernel form needs V'[A2 | b] only.  Measured and dropped in round 4: the sum
            //  over the k slices taken by k_cqr_tw itself instead of the k_qr1_wreduce launch -- 32 workgroups reading 1 MB of
            //  partials each take longer than the 10 us launch over 256: C3 7.83 against 7.60 ms, profiles/r04/ab_c3_tw.txt)
            const int ks = qr1_vtb_launch(c, q, vcur, ldv, A, M, c0, cend, n, rhs, ncolsB, 1, vscur);
            if (ks < 0) { lsq_set_error("qr: the V'[A2 | b] launch failed"); return LSQ_EHIP; }
            {
                long long tot = (long long)(ntile - 1) * Q2_NB * Q2_NB;
                int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 4);
                LSQ_LAUNCH(k_qr1_wreduce, dim3(g), dim3(256), 0, c->stream, q->Wp, ncolsB, ks, q->W, 1);
            }
            // (Q1 form: W2 also in the update's fragment order -- LSQ_QR_NO_SWIZZLE=1: round 5's loads, A/B)
            const double *w2s = q->cq.q1form && swz_on ? q->W2s : nullptr;
            LSQ_TRY(lsq_cqr_tw(c, &q->cq, q->W, ncolsB, A, M, c0, cend, n, rhs, vcur, ldv, q->W2, const_cast<double *>(w2s)));
            // is the next panel a CholeskyQR2 panel too, with enough other columns beside it?
            const int c1 = cend;
            const bool next_cq = n - c1 >= Q2_NB && M - c1 >= 256;       // the next panel is a CholeskyQR panel too
            const bool ahead = la_on && next_cq && ncols - 1 >= Q2_NB + la_min_cols && (!la_auto || M - c1 >= la_q1_min_rows);
            double *const gram_out = gram_on && next_cq ? q->cq.Gp : nullptr;
            if (ahead) {
                if (!q->Vb2) LSQ_HIP(hipMalloc(&q->Vb2, ((size_t)q->M * Q2_NB + 64) * sizeof(double)));
                double *vnext = vcur == q->Vb ? q->Vb2 : q->Vb;
                double *vsnext = nullptr;
                if (swz_on && lsq_cqr_q1form()) {
                    if (!q->Vs2) LSQ_HIP(hipMalloc(&q->Vs2, ((size_t)(q->M + 16) * Q2_NB + 64) * sizeof(double)));
                    vsnext = vnext == q->Vb ? q->Vs : q->Vs2;
                }
                bool tk = false;
                if (gram_out) LSQ_TRY(qr1_update_wave(c, vcur, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk, Q2_NB, Q2_NB, false, gram_out, nullptr, nullptr, w2s));
                else LSQ_TRY(qr1_update_wave(c, vcur, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk, 0, Q2_NB, false, nullptr, nullptr, nullptr, w2s));
                LSQ_HIP(hipEventRecord(q->cq.ev_first, c->stream));
                LSQ_HIP(hipStreamWaitEvent(q->cq.ahead, q->cq.ev_first, 0));
                LSQ_TRY(lsq_cqr_panel(c, &q->cq, A, M, c1, vnext, M - c1, q->d_err, q->cq.ahead, gram_out != nullptr, false, vsnext));
                LSQ_HIP(hipEventRecord(q->cq.ev_panel, q->cq.ahead));
                LSQ_TRY(qr1_update_wave(c, vcur, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk, Q2_NB, ncols, true, nullptr, nullptr, nullptr, w2s));
                vcur = vnext;
                vscur = vsnext;
                pre = true;
            } else {
                const int nrt = (rows + Q2_NB - 1) / Q2_NB, nct = (ncols + Q2_NB - 1) / Q2_NB;
                bool tk = false;
                if (gram_out) {
                    const bool hn = hier_on && lsq_cqr_hier((M - c1 + CQ_RS - 1) / CQ_RS);      // (the NEXT panel's slabs)
                    LSQ_TRY(qr1_update_wave(c, vcur, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk, Q2_NB, ncols, false, gram_out,
                                            hn ? q->cq.Gq1 : nullptr, hn ? q->cq.gcnt : nullptr, w2s));
                    gram_ready = true;
                    hier_ready = hn;
                } else
                LSQ_TRY(qr1_update_wave(c, vcur, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk, 0, -1, false, nullptr, nullptr, nullptr, w2s));
                if (!tk)
                LSQ_LAUNCH(k_qr1_update, dim3(nrt * ((nct + Q2_UCT - 1) / Q2_UCT)), dim3(256), 0, c->stream, vcur, ldv, A, M,
                                   c0, cend, n, rhs, ncols, q->W2);
            }
            continue;
        }
        bool lazy = false;
        int side_k = 0;   // > 0: later pivot columns of a launch sit in the side panel
        // last panel: b rides through the steps as one more target column, so no block update is left to do
        const bool ride = cend == n;
        bool rode = false;
        auto steps = [&](auto kern) {
            LSQ_LAUNCH(kern, dim3(1), dim3(QR_NT), 0, c->stream, A, M, cend, c0, 1, q->tau1);
            for (int i = c0; i + 1 < cend; ++i)
                LSQ_LAUNCH(kern, dim3(cend - i - 1), dim3(QR_NT), 0, c->stream, A, M, cend, i, 0, q->tau1);
        };
        auto steps_multi = [&](auto kern, int nt, int K, int S) {
            for (int i = c0; i < cend; i += K) {
                const int kk = std::min(K, cend - i), G = std::max(1, cend - i - kk + (ride ? 1 : 0));
                const int grid = S >= 64 ? G * S : S > 1 ? 8 * S * ((G + 7) / 8) : G;
                LSQ_LAUNCH(kern, dim3(grid), dim3(nt), 0, c->stream, A, M, cend, i, kk, q->tau1, q->lazy, q->lazy + n, G,
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
        const int coop = c->num_cus < 256 || q->no_exchange ? 0 : cv ? atoi(cv) : 1;
        if (coop && prow > 2 * 8 * 256 && prow <= 4 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 4>, 256, 4, 4);
        else if (coop && prow > 4 * 8 * 256 && prow <= 8 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 8>, 256, 4, 8);
        else if (coop && prow > 8 * 8 * 256 && prow <= 8 * 10 * 256) steps_multi(k_qr1_step_multi<256, 10, 4, 8>, 256, 4, 8);
        // tall operands: more slabs (the exchange of a round grows with S; there are few columns to pay it)
        else if (coop && prow > 8 * 10 * 256 && prow <= 8 * 16 * 256) steps_multi(k_qr1_step_multi<256, 16, 4, 8>, 256, 4, 8);
        else if (coop && prow > 8 * 16 * 256 && prow <= 32 * 16 * 256) steps_multi(k_qr1_step_multi<256, 16, 4, 32>, 256, 4, 32);
        else if (coop && prow > 32 * 16 * 256 && prow <= 64 * 32 * 256) steps_multi(k_qr1_step_multi<256, 32, 2, 64>, 256, 2, 64);
        else if (coop && prow > 64 * 32 * 256 && prow <= 256 * 32 * 256) steps_multi(k_qr1_step_multi<256, 32, 2, 256>, 256, 2, 256);
        else if (coop && prow > 8 * 256 && prow <= 2 * 8 * 256) steps_multi(k_qr1_step_multi<256, 8, 4, 2>, 256, 4, 2);
        else if (prow <= 8 * 512) steps_multi(k_qr1_step_multi<512, 8, 4, 1>, 512, 4, 1);
        else if (prow <= 8 * 1024) steps_multi(k_qr1_step_multi<512, 16, 4, 1>, 512, 4, 1);
        else if (prow <= 32 * 512) steps_multi(k_qr1_step_multi<512, 32, 2, 1>, 512, 2, 1);
        else if (prow <= 40 * 512) steps_multi(k_qr1_step_multi<512, 40, 2, 1>, 512, 2, 1);
        else steps(k_qr1_step);     // (taller than 20480 rows with the slab exchange off: the plain per-column loop)
        if (rode) {
            LSQ_LAUNCH(k_qr1_fin, dim3(1), dim3(256), 0, c->stream, A, M, c0, nb, (const double *)q->lazy,
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
            LSQ_LAUNCH(k_qr1_vbuf, dim3(g), dim3(256), 0, c->stream, A, M, c0, nb, q->Vb, ldv,
                               lazy ? (const double *)q->lazy : (const double *)nullptr,
                               lazy ? (const double *)(q->lazy + n) : (const double *)nullptr,
                               side_k ? (const double *)q->Pn : (const double *)nullptr, std::max(1, side_k));
        }
        const int ks = qr1_vtb_launch(c, q, q->Vb, ldv, A, M, c0, cend, n, rhs, ncolsB, 0);
        if (ks < 0) { lsq_set_error("qr: the V'[A2 | b] launch failed"); return LSQ_EHIP; }
        {
            long long tot = (long long)ntile * Q2_NB * Q2_NB;
            int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 4);
            LSQ_LAUNCH(k_qr1_wreduce, dim3(g), dim3(256), 0, c->stream, q->Wp, ncolsB, ks, q->W, 0);
        }
        LSQ_LAUNCH(k_qr1_tw_mfma, dim3(std::max(1, lsq_div_up(ncols, Q2_NB))), dim3(256), 0, c->stream, q->W, ncolsB,
                               q->tau1, c0, nb, q->W2);
        {
            const int nrt = (rows + Q2_NB - 1) / Q2_NB, nct = (ncols + Q2_NB - 1) / Q2_NB;
            bool tk = false;
            LSQ_TRY(qr1_update_wave(c, q->Vb, ldv, A, M, c0, cend, n, rhs, ncols, q->W2, &tk));
            if (!tk)
            LSQ_LAUNCH(k_qr1_update, dim3(nrt * ((nct + Q2_UCT - 1) / Q2_UCT)), dim3(256), 0, c->stream, q->Vb, ldv, A, M, c0, cend, n, rhs, ncols,
                               q->W2);
        }
    }
    {
        long long tot = (long long)n * n;
        int g = (int)std::min<long long>((tot + 255) / 256, (long long)c->num_cus * 8);
        LSQ_LAUNCH(k_qr1_extract, dim3(g), dim3(256), 0, c->stream, A, M, n, rhs, q->R, q->rhs2);
    }
    LSQ_HIP(hipGetLastError());
    s->last_qr_panel = q->cholqr_used ? 2 : 1;
    *R_out = q->R;
    *rhs_out = q->rhs2;
    return LSQ_OK;
}

bool lsq_qr2_applies(int M, int n) { return qr2_applies(M, n); }
int lsq_qr2_factor(lsq_solver *s, int M, int n, double **R_out, double **rhs_out) { return qr2_factor(s, M, n, R_out, rhs_out); }
