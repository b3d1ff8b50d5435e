// 64 x 64 fp64 matrix routines for ONE 256-thread workgroup (4 wavefronts), operands in LDS.
//
// Why: every 64-column step of the dense factorisations (panel of the blocked QR, dense_qr.jl:30-88; diagonal block of
// the blocked Cholesky, dense_cholesky.jl:43-59) ends in a small dependent chain -- a 64 x 64 Cholesky / LU /
// triangular inverse -- that one CU has to run alone.  Written column-by-column with workgroup barriers such a chain
// costs 20-60 us; here the only serial pieces are 16 x 16 diagonal blocks factored by ONE wavefront in registers
// (lane = column, pivots broadcast through SGPRs with v_readlane, reciprocal square roots by v_rsq_f64 + Newton),
// and everything off the diagonal is 16 x 16 tile products on the fp64 MFMA unit (v_mfma_f64_16x16x4_f64).
//
// Storage: row-major in LDS, element (r, c) at [r * S64_LS + c].  Triangular results are written with explicit zeros in
// the other triangle, so tile products need no masking.
//
// v_mfma_f64_16x16x4_f64:  D(16x16) += A(16x4) B(4x16);  lane l holds A[i = l & 15][k = l >> 4],  B[k = l >> 4][j = l & 15]
// and D[i = (l >> 4) + 4 r][j = l & 15], r = 0..3   (cdna_hip_programming.md, MFMA operand layouts)
#pragma once
#include <hip/hip_runtime.h>

typedef double s64_v4d __attribute__((ext_vector_type(4)));
constexpr int S64_N = 64;
constexpr int S64_LS = 65;                 // row stride (doubles)
constexpr int S64_MAT = S64_N * S64_LS;    // doubles per matrix buffer
constexpr int S64_TMP = 32 * 33;           // doubles of the level-product scratch

__device__ __forceinline__ double s64_readlane(double x, int l) {   // l: wave-uniform (a constant after unrolling)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// 1/sqrt(x) and 1/x to full double precision: hardware estimate (relative error 2^-24, measured) + ONE third-order
// correction step (error -> e^3 = 2^-72), explicit FMAs: four dependent operations instead of a v_div_* / v_sqrt sequence
__device__ __forceinline__ double s64_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);          // 1 - x y^2
    const double q = e * __builtin_fma(0.375, e, 0.5);         // e/2 + 3 e^2/8
    return __builtin_fma(y, q, y);
}
__device__ __forceinline__ double s64_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);                // 1 - x r
    return __builtin_fma(r, __builtin_fma(e, e, e), r);        // r (1 + e + e^2)
}

// acc += A_tile * B_tile over `ktiles` 16-wide k tiles.
//   A tile: rows ar..ar+15, k along the columns from ac  (TA: the transpose -- rows are k from ar, columns ac..ac+15)
//   B tile: k along the rows from br, columns bc..bc+15  (TB: the transpose -- rows br..br+15 are the columns, k from bc)
template <bool TA, bool TB>
__device__ __forceinline__ void s64_tile_mma(s64_v4d &acc, const double *__restrict__ A, int ar, int ac,
                                             const double *__restrict__ B, int br, int bc, int ktiles, int lane) {
    const int ij = lane & 15, kq = lane >> 4;
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kt * 16 + kk * 4 + kq;
            const double a = TA ? A[(ar + k) * S64_LS + ac + ij] : A[(ar + ij) * S64_LS + ac + k];
            const double b = TB ? B[(br + ij) * S64_LS + bc + k] : B[(br + k) * S64_LS + bc + ij];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
}
// tile (rows r0.., columns c0..) of an LDS matrix <- sgn * acc (+ the old tile when ADD)
template <bool ADD>
__device__ __forceinline__ void s64_tile_store(double *__restrict__ M, int r0, int c0, const s64_v4d &acc, double sgn, int lane) {
    const int j = lane & 15, i0 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double *p = M + (r0 + i0 + 4 * r) * S64_LS + c0 + j;
        *p = ADD ? *p + sgn * acc[r] : sgn * acc[r];
    }
}

// ---- 16 x 16 diagonal block of a Cholesky factorisation G = U'U, ONE wavefront, registers ----------------------
// lanes 0..15: column c of the block (rows r <= c); lanes 16..31: column c of the identity, which the same row
// operations turn into inv(U_kk)' (lower triangular).  Writes U_kk (zeros below the diagonal) into M in place and
// inv(U_kk)' into W at the same position.  Returns 0, or 1 + the index (inside the block) of the first pivot that is not
// positive (wave-uniform).
__device__ __forceinline__ int s64_chol16(double *__restrict__ M, double *__restrict__ W, int o, int lane) {
    const int c = lane & 15;
    const bool mat = lane < 16, idn = lane >= 16 && lane < 32;
    double u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = mat ? (r <= c ? M[(o + r) * S64_LS + o + c] : 0.0) : ((idn && r == c) ? 1.0 : 0.0);
    int bad = 0;                                 // 1 + index of the first pivot that is not positive
    // software-pipelined: the next pivot's reciprocal square root (a long dependent chain) is issued as soon as row j+1
    // has its update, and the updates of the other rows fill its latency (one wavefront per SIMD: nothing else would)
    double ajj = s64_readlane(u[0], 0);
    double sj = s64_rsqrt(ajj);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        bad = (bad == 0 && !(ajj > 0.0)) ? j + 1 : bad;   // (such a pivot turns the rest into NaN / Inf: reported, not used)
        const double rowj = u[j] * sj;           // U[j][c] (c >= j); identity lanes: row j of inv(U)' so far
        u[j] = rowj;
        if (j + 1 < 16) {
            u[j + 1] = __builtin_fma(-s64_readlane(rowj, j + 1), rowj, u[j + 1]);
            ajj = s64_readlane(u[j + 1], j + 1);
            sj = s64_rsqrt(ajj);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = j + 2; i < 16; ++i) u[i] = __builtin_fma(-s64_readlane(rowj, i), rowj, u[i]);   // U[j][i] * U[j][c]
        __builtin_amdgcn_sched_barrier(0);
    }
    if (mat) {
#pragma unroll
        for (int r = 0; r < 16; ++r) M[(o + r) * S64_LS + o + c] = r <= c ? u[r] : 0.0;
    } else if (idn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) W[(o + r) * S64_LS + o + c] = c <= r ? u[r] : 0.0;
    }
    return bad;
}
// (An alternative measured and NOT used: tools/micro/chol16_bench.hip keeps a form of this routine in which every update
// is one v_fmac_f64_dpp row_newbcast instead of two v_readlane_b32 + an fma -- 9.5 clocks instead of 22 per update
// (tools/micro/dpp_probe.hip) -- with the matrix row copied into the identity lanes' row by v_permlane16_swap_b32.  Same
// bits, but 4800 clocks per block against 4430 for this one: a single wavefront issues in order, the copy + broadcast
// lengthen the dependent chain of every pivot by more than the updates save.)

// ---- Cholesky G = U'U of a 64 x 64 matrix (upper triangle of M is read), in place: M <- U (zeros below) --------------
// W receives the four inv(U_kk)' diagonal blocks (the other entries of W are not touched).  Returns 0, or (uniform)
// 1 + the index of the first pivot that is not positive -- M then holds garbage.  `fail` is an LDS int.
// idle(kb, wv, lane): what wavefronts 1..3 do while wavefront 0 factors diagonal block kb (no barriers in there; tiles of M
// other than (kb, kb) and buffers other than W may be touched).
struct S64NoIdle { __device__ __forceinline__ void operator()(int, int, int) const {} };
template <bool ZERO_LOWER = true, class F = S64NoIdle>     // ZERO_LOWER = false: the blocks below the diagonal blocks keep G / garbage
__device__ __forceinline__ int s64_chol(double *M, double *W, int *fail, int tid, long long *tr = nullptr,   // tr: debug stamps (10 ns)
                                        F idle = F()) {
    const int lane = tid & 63, wv = tid >> 6;
    if (tid == 0) *fail = 0;
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
        const int o = kb * 16;
        if (wv == 0) {
            const int bad = s64_chol16(M, W, o, lane);
            if (bad && lane == 0 && *fail == 0) *fail = o + bad;
        } else {
            idle(kb, wv, lane);
        }
        if (tr && tid == 0) tr[3 * kb] = wall_clock64();
        __syncthreads();
        const int nt = 3 - kb;                   // tiles to the right
        if (wv < nt) {                           // row panel: U[o.., t] = inv(U_kk)' * G[o.., t]
            const int t = kb + 1 + wv;
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<false, false>(acc, W, o, o, M, o, 16 * t, 1, lane);
            s64_tile_store<false>(M, o, 16 * t, acc, 1.0, lane);
        }
        __syncthreads();
        if (tr && tid == 0) tr[3 * kb + 1] = wall_clock64();
        // trailing tiles (t <= t') -= U[o.., t]' U[o.., t']
        for (int q = wv; q < nt * (nt + 1) / 2; q += 4) {
            int t = 0, r = q;
            while (r >= nt - t) { r -= nt - t; ++t; }
            const int ta = kb + 1 + t, tb = ta + r;
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<true, false>(acc, M, o, 16 * ta, M, o, 16 * tb, 1, lane);
            s64_tile_store<true>(M, 16 * ta, 16 * tb, acc, -1.0, lane);
        }
        __syncthreads();
        if (tr && tid == 0) tr[3 * kb + 2] = wall_clock64();
    }
    // zeros below the diagonal blocks (the strictly lower tiles still hold G / garbage)
    if (ZERO_LOWER) {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            if ((r >> 4) > (c >> 4)) M[r * S64_LS + c] = 0.0;
        }
        __syncthreads();
    }
    return *fail;
}

// ---- X = inv(U) for an upper triangular U whose diagonal-block inverses are already in X (upper, zeros elsewhere in
// the block); off-diagonal blocks by two levels of  X_AB = -X_AA (U_AB X_BB)  on the MFMA unit.  T: S64_TMP scratch.
// All of X outside the diagonal blocks is overwritten (zeros below).
__device__ __forceinline__ void s64_triinv_levels(const double *__restrict__ U, double *__restrict__ X, double *__restrict__ T,
                                                  int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        if ((r >> 4) != (c >> 4)) X[r * S64_LS + c] = 0.0;
    }
    __syncthreads();
    // level 1: pairs (0,1) and (2,3), 16 x 16 blocks
    if (wv < 2) {
        const int oa = wv * 32, ob = oa + 16;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
        s64_tile_mma<false, false>(acc, U, oa, ob, X, ob, ob, 1, lane);        // U_AB X_BB
        const int j = lane & 15, i0 = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(wv * 16 + i0 + 4 * r) * 33 + j] = acc[r];
    }
    __syncthreads();
    if (wv < 2) {
        const int oa = wv * 32, ob = oa + 16;
        const int ij = lane & 15, kq = lane >> 4;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kk * 4 + kq;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(oa + ij) * S64_LS + oa + k], T[(wv * 16 + k) * 33 + ij], acc, 0, 0, 0);
        }
        s64_tile_store<false>(X, oa, ob, acc, -1.0, lane);
    }
    __syncthreads();
    // level 2: A = rows/cols 0..31, B = 32..63; one 16 x 16 tile of the 32 x 32 block per wavefront
    {
        const int ti = wv >> 1, tj = wv & 1;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
        s64_tile_mma<false, false>(acc, U, 16 * ti, 32, X, 32, 32 + 16 * tj, 2, lane);   // U_AB X_BB (K = 32)
        const int j = lane & 15, i0 = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * ti + i0 + 4 * r) * 33 + 16 * tj + j] = acc[r];
    }
    __syncthreads();
    {
        const int ti = wv >> 1, tj = wv & 1;
        const int ij = lane & 15, kq = lane >> 4;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 4 + kq;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(16 * ti + ij) * S64_LS + k], T[k * 33 + 16 * tj + ij], acc, 0, 0, 0);
        }
        s64_tile_store<false>(X, 16 * ti, 32 + 16 * tj, acc, -1.0, lane);
    }
    __syncthreads();
}

// inverse of the four 16 x 16 diagonal blocks of an upper triangular U by back substitution (one lane per column,
// four wavefronts = four blocks): X_kk <- inv(U_kk) (upper; zeros below inside the block)
__device__ __forceinline__ void s64_diaginv_upper(const double *__restrict__ U, double *__restrict__ X, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < 16) {
        const int o = wv * 16, cc = lane;
        double x[16];
#pragma unroll
        for (int r = 15; r >= 0; --r) {
            double acc = r == cc ? 1.0 : 0.0;
#pragma unroll
            for (int k = r + 1; k < 16; ++k) acc -= U[(o + r) * S64_LS + o + k] * x[k];
            x[r] = r <= cc ? acc * s64_rcp(U[(o + r) * S64_LS + o + r]) : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(o + r) * S64_LS + o + cc] = x[r];
    }
    __syncthreads();
}

// after s64_chol(M = U, W): W <- inv(U) IN PLACE (its diagonal blocks hold inv(U_kk)', i.e. the inverse blocks transposed)
__device__ __forceinline__ void s64_chol_inverse(const double *__restrict__ U, double *__restrict__ W, double *__restrict__ T,
                                                 int tid) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, kb = e >> 8, i = (e >> 4) & 15, j = e & 15, o = kb * 16;
        t[q] = W[(o + j) * S64_LS + o + i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, kb = e >> 8, i = (e >> 4) & 15, j = e & 15, o = kb * 16;
        W[(o + i) * S64_LS + o + j] = t[q];
    }
    __syncthreads();
    s64_triinv_levels(U, W, T, tid);
}

// ---- 16 x 16 diagonal block of the "modified LU" of the Householder reconstruction, ONE wavefront, registers --------
// (Ballard, Demmel, Grigori, Jacquelin, Knight, Nguyen: "Reconstructing Householder vectors from Tall-Skinny QR")
//   for j: S_j = -sign(w_jj); w_jj -= S_j; column j below the diagonal /= w_jj; Schur update.
// lanes 0..15: column c of the block; lanes 16..31: identity columns -> inv(L_kk) (unit lower).  In place in M (L strictly
// below, U on/above); inv(L_kk) into Li; S and 1/U_jj into sS / sR (64-entry LDS arrays).
__device__ __forceinline__ void s64_lu16(double *__restrict__ M, double *__restrict__ Li, double *__restrict__ sS,
                                         double *__restrict__ sR, int o, int lane) {
    const int c = lane & 15;
    const bool mat = lane < 16, idn = lane >= 16 && lane < 32;
    double w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = mat ? M[(o + r) * S64_LS + o + c] : ((idn && r == c) ? 1.0 : 0.0);
    double myrp = 0.0;                                   // lane j keeps 1/U_jj: its column is scaled into L at the end
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double wjj = s64_readlane(w[j], j);
        const double sj = wjj >= 0.0 ? -1.0 : 1.0;      // S = -sign: the pivot |w_jj - S_j| >= 1 never cancels
        const double p = wjj - sj;
        const double rp = s64_rcp(p);
        if (lane == j) { w[j] = p; myrp = rp; }
        if (lane == 0) { sS[o + j] = sj; sR[o + j] = rp; }
        // row j scaled by 1/pivot, ZERO in the columns that are already final (c <= j): those lanes (the multipliers
        // L[i][c] they hold, and column j itself, which is scaled at the end) pass through the updates unchanged
        const double rs = (mat && c <= j) ? 0.0 : w[j] * rp;
#pragma unroll
        for (int i = j + 1; i < 16; ++i) w[i] = __builtin_fma(-s64_readlane(w[i], j), rs, w[i]);   // a_ij * (u_jc / u_jj)
    }
    if (mat) {
#pragma unroll
        for (int r = 0; r < 16; ++r) M[(o + r) * S64_LS + o + c] = r > c ? w[r] * myrp : w[r];
    } else if (idn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Li[(o + r) * S64_LS + o + c] = c <= r ? w[r] : 0.0;
    }
}

// ---- modified LU of a 64 x 64 matrix M = Q_top, in place: strictly lower = Y1 (unit lower L), upper = U; S (signs)
// and the reciprocals of U's diagonal in sS / sR; Li receives the four inv(L_kk) blocks.
__device__ __forceinline__ void s64_lu_modified(double *__restrict__ M, double *__restrict__ Li, double *__restrict__ sS,
                                                double *__restrict__ sR, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    for (int kb = 0; kb < 4; ++kb) {
        const int o = kb * 16;
        if (wv == 0) s64_lu16(M, Li, sS, sR, o, lane);
        __syncthreads();
        const int nt = 3 - kb, nrows = 16 * nt;
        // row panel (wavefronts 1..3, one tile each): U[o.., t] = inv(L_kk) * A[o.., t]
        if (wv >= 1 && wv - 1 < nt) {
            const int t = kb + wv;
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<false, false>(acc, Li, o, o, M, o, 16 * t, 1, lane);
            s64_tile_store<false>(M, o, 16 * t, acc, 1.0, lane);
        }
        // column panel (wavefront 0, one thread per row below): L[r][o..] = A[r][o..] inv(U_kk) by substitution
        if (wv == 0 && lane < nrows) {
            const int r = o + 16 + lane;
            double x[16];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                double acc = M[r * S64_LS + o + cc];
#pragma unroll
                for (int k = 0; k < cc; ++k) acc -= x[k] * M[(o + k) * S64_LS + o + cc];
                x[cc] = acc * sR[o + cc];
            }
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) M[r * S64_LS + o + cc] = x[cc];
        }
        __syncthreads();
        // trailing tiles -= L[ti][o..] U[o..][tj]
        for (int q = wv; q < nt * nt; q += 4) {
            const int ti = kb + 1 + q / nt, tj = kb + 1 + q % nt;
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<false, false>(acc, M, 16 * ti, o, M, o, 16 * tj, 1, lane);
            s64_tile_store<true>(M, 16 * ti, 16 * tj, acc, -1.0, lane);
        }
        __syncthreads();
    }
}

// C (64 x 64) = sgn * op(A) * op(B) for LDS matrices; TA/TB as in s64_tile_mma.  The fp64 MFMA unit of one CU delivers
// 128 flop/clock, so a full 64^3 product costs ~1.7 us: SHAPE names the zero structure of the operands so that only the
// tile products that can be non-zero are formed.
//   S64_FULL   all 16 tiles, k over 0..63
//   S64_UU     op(A), op(B) upper triangular: tiles ti <= tj, k-tiles ti..tj; the tiles below are set to zero  (20 of 64)
//   S64_LtU    op(A) lower (e.g. X' for an upper X), op(B) upper: C = A'A-like, only tiles ti <= tj are formed and nothing
//              else is written (the caller reads the upper triangle); k-tiles 0..ti                               (20 of 64)
//   S64_UL     op(A) upper, op(B) lower: full result, k-tiles max(ti, tj)..3                                       (30 of 64)
//   S64_LF     op(A) lower, op(B) full: k-tiles 0..ti                                                                (40 of 64)
//   S64_UF     op(A) upper, op(B) full: k-tiles ti..3                                                                (40 of 64)
// ldc: row stride of C (S64_LS for an LDS matrix, 64 for a row-major matrix in global memory).
enum { S64_FULL = 0, S64_UU = 1, S64_LtU = 2, S64_UL = 3, S64_LF = 4, S64_UF = 5 };
template <bool TA, bool TB, int SHAPE = S64_FULL>
__device__ __forceinline__ void s64_gemm(double *__restrict__ C, const double *__restrict__ A, const double *__restrict__ B,
                                         double sgn, int tid, int ldc = S64_LS) {
    const int lane = tid & 63, wv = tid >> 6;
    const int ij = lane & 15, kq = lane >> 4;
    for (int q = wv; q < 16; q += 4) {
        // tiles are dealt so that the four wavefronts get equal work for the triangular shapes (ti + tj pairs)
        const int ti = (SHAPE == S64_LF) ? ((q >> 2) + (q & 3)) & 3 : q >> 2;      // (S64_LF: rows dealt round-robin: equal work)
        const int tj = (SHAPE == S64_FULL || SHAPE == S64_UL || SHAPE == S64_LF || SHAPE == S64_UF) ? (q & 3) : ((q & 3) + ti) & 3;
        int k0 = 0, k1 = 4;
        bool live = true;
        if (SHAPE == S64_UU) { live = ti <= tj; k0 = ti; k1 = tj + 1; }
        if (SHAPE == S64_LtU) { live = ti <= tj; k1 = ti + 1; }
        if (SHAPE == S64_UL) { k0 = ti > tj ? ti : tj; }
        if (SHAPE == S64_LF) { k1 = ti + 1; }
        if (SHAPE == S64_UF) { k0 = ti; }       // (ti = q >> 2: every wavefront gets one tile row of each length)
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
        if (live) {
            for (int kt = k0; kt < k1; ++kt) {
                double a[4], b[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = kt * 16 + kk * 4 + kq;
                    a[kk] = TA ? A[k * S64_LS + 16 * ti + ij] : A[(16 * ti + ij) * S64_LS + k];
                    b[kk] = TB ? B[(16 * tj + ij) * S64_LS + k] : B[k * S64_LS + 16 * tj + ij];
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
            }
        }
        if (live || SHAPE == S64_UU) {
            const int i0 = lane >> 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(16 * ti + i0 + 4 * r) * ldc + 16 * tj + ij] = sgn * acc[r];
        }
    }
    __syncthreads();
}
