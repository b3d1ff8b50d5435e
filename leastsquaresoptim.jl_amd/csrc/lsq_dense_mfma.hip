// Dense normal-equation path (dense_cholesky.jl:29-59): fp64-MFMA SYRK for J'J (k_syrk_mfma; the pair kernel
// k_syrk_small when there are few columns and many rows), blocked right-looking Cholesky (64-wide panels, diagonal
// block and row panel in one launch: k_chol_panel_mfma; trailing update on the MFMA kernel again) and the two
// triangular solves pipelined over the 64-blocks (lsq_tri_chol_solve in lsq_dense.hip; k_chol_trsv as fallback).
//
// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16), one f64 of A and of B per lane
//   A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
//   D: 4 f64 per lane, col = lane&15, row = (lane>>4) + 4*reg            (cdna_hip_programming.md §3)
// Here A = (tile of J)' and B = tile of J, both read from an LDS image [column][k] of J's columns
// (J is column-major, so a column's k-run is contiguous in global memory and in LDS).
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "lsq_small64.h"
#include "lsq_solver.h"

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int MT = 64;        // output tile (MT x MT) per 256-thread workgroup; each wave 32 x 32
constexpr int KC = 32;        // k-rows staged per step
constexpr int KS = KC + 2;    // LDS row stride in doubles: 68 dwords = 4 banks apart => conflict-free b64 reads

// C(tile bi,bj) (+)= A(:, i-range)' * A(:, j-range) over rows [k_begin, k_end) of A (lda).
// MODE 0: write the partial tile to W[slice][tile] (split-K, reduced by k_syrk_reduce)
// MODE 1: C_tile -= product (trailing update of the blocked Cholesky; upper tiles only)
template <int MODE>
__global__ void __launch_bounds__(256)
k_syrk_mfma(const double *__restrict__ A, int lda, int krows, int ncols, int col0, int kslices, double *__restrict__ W,
            double *__restrict__ C, int ldc, const double *__restrict__ jy = nullptr, double *__restrict__ jx = nullptr) {
    __shared__ double sA[MT * KS];
    __shared__ double sB[MT * KS];
    const int nt = (ncols + MT - 1) / MT;
    const int ntiles = nt * (nt + 1) / 2;
    if constexpr (MODE == 0) {
        // mul!(x, J', y) of the normal-equations solve rides in this launch: workgroups past the tiles take one column each
        // (k_dense_t's arithmetic, sum for sum: the result does not depend on which launch formed it)
        if ((int)blockIdx.x >= ntiles * kslices) {
            const int b = (int)blockIdx.x - ntiles * kslices;
            const double *col = A + (size_t)(col0 + b) * lda;
            double a0 = 0.0, a1 = 0.0;
            int i = threadIdx.x;
            for (; i + LSQ_NT < krows; i += 2 * LSQ_NT) {
                const double c0 = col[i], c1 = col[i + LSQ_NT];
                a0 += c0 * jy[i];
                a1 += c1 * jy[i + LSQ_NT];
            }
            if (i < krows) a0 += col[i] * jy[i];
            const double sum = block_sum<LSQ_NT>(a0 + a1, sA);
            if (threadIdx.x == 0) jx[b] = 1.0 * sum;
            return;
        }
    }
    const int tile = blockIdx.x % ntiles, slice = blockIdx.x / ntiles;
    int t = tile, bi = 0;
    while (t >= nt - bi) { t -= nt - bi; ++bi; }
    const int bj = bi + t;
    const int i0 = bi * MT, j0 = bj * MT;
    const int kper = ((krows + kslices - 1) / kslices + KC - 1) / KC * KC;
    const int kb = slice * kper, ke = min(krows, kb + kper);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;   // this wave's 32x32 quadrant of the tile
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lc = tid >> 2, lk = (tid & 3) * 8;        // loader: column lc (0..63), 8 consecutive k
    const int ci = i0 + lc, cj = j0 + lc;
    const double *pa0 = A + (size_t)(col0 + min(ci, ncols - 1)) * lda + lk;
    const double *pb0 = A + (size_t)(col0 + min(cj, ncols - 1)) * lda + lk;
    // the next k-slab is fetched into registers while the MFMAs of the current one run
    double ra[8], rb[8];
    auto fetch = [&](int k0) {
        if (k0 + KC <= ke) {                   // full slab: unconditional fetches (columns past the end are clamped, then zeroed)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double x = pa0[k0 + q], y = pb0[k0 + q];
                ra[q] = ci < ncols ? x : 0.0;
                rb[q] = cj < ncols ? y : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool kin = (k0 + lk + q) < ke;
                ra[q] = (kin && ci < ncols) ? pa0[k0 + q] : 0.0;
                rb[q] = (kin && cj < ncols) ? pb0[k0 + q] : 0.0;
            }
        }
    };
    if (kb < ke) fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += KC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sA[lc * KS + lk + q] = ra[q];
            sB[lc * KS + lk + q] = rb[q];
        }
        __syncthreads();
        if (k0 + KC < ke) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const int ko = kk + (lane >> 4);
            const double a0 = sA[(wr + (lane & 15)) * KS + ko];
            const double a1 = sA[(wr + 16 + (lane & 15)) * KS + ko];
            const double b0 = sB[(wc + (lane & 15)) * KS + ko];
            const double b1 = sB[(wc + 16 + (lane & 15)) * KS + ko];
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
                const int row = wr + a * 16 + (lane >> 4) + 4 * r;   // f64 C/D map
                const int col = wc + b * 16 + (lane & 15);
                if (MODE == 0) {
                    W[((size_t)slice * ntiles + tile) * (MT * MT) + (size_t)col * MT + row] = acc[a][b][r];
                } else {
                    const int gi = i0 + row, gj = j0 + col;
                    if (gi < ncols && gj < ncols && gi <= gj)
                        C[(size_t)(col0 + gj) * ldc + col0 + gi] -= acc[a][b][r];
                }
            }
}

// C(upper) = sum over slices of W, + damp on the diagonal (dense_cholesky.jl:51-53)
__global__ void __launch_bounds__(256)
k_syrk_reduce(const double *__restrict__ W, int n, int kslices, const double *__restrict__ damp, double *__restrict__ C,
              int *__restrict__ info) {   // info: the factorisation's status word, cleared here (no memset launch of its own)
    if (blockIdx.x == 0 && threadIdx.x == 0) *info = 0;
    const int nt = (n + MT - 1) / MT;
    const int ntiles = nt * (nt + 1) / 2;
    const int tile = blockIdx.x / 16, part = blockIdx.x % 16;   // 16 workgroups per tile, 256 entries each
    int t = tile, bi = 0;
    while (t >= nt - bi) { t -= nt - bi; ++bi; }
    const int bj = bi + t;
    {
        const int e = part * 256 + threadIdx.x;
        const int row = e % MT, col = e / MT;
        const int gi = bi * MT + row, gj = bj * MT + col;
        if (gi < n && gj < n && gi <= gj) {
            double s = 0.0;
            int sl = 0;
            for (; sl + 8 <= kslices; sl += 8) {     // eight loads in flight, added in slice order (fixed order)
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t8[u] = W[((size_t)(sl + u) * ntiles + tile) * (MT * MT) + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t8[u];
            }
            for (; sl < kslices; ++sl) s += W[((size_t)sl * ntiles + tile) * (MT * MT) + e];
            if (gi == gj && damp) s += damp[gi];
            C[(size_t)gj * n + gi] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// blocked Cholesky, step j0: every workgroup factors the NB x NB diagonal block in LDS (cheap and saves a launch) while
// the loads of its own 64 columns of the row panel A(j0:j0+NB, j0+NB:n) are in flight, then turns them into
// U12 = U11^{-T} A12.  The factored diagonal block is NOT written in place (a workgroup that starts late would read it
// as input): workgroup 0 parks it in Ds[j0 / 64] and k_chol_diag_restore moves all blocks back after the last panel.
// (Round 1 did this with register / substitution kernels -- k_chol_diag16, k_chol_trsm16, k_chol_panel16: 58 us per step.)
// ---------------------------------------------------------------------------------------------
constexpr int NB = 64;
// On the fp64 MFMA unit (lsq_small64.h): the only dependent chains left are the four 16 x 16 diagonal
// sub-blocks (one wavefront, registers); row panels and trailing tiles of the 64 x 64 block, its explicit inverse
// W = inv(U11) and the row panel of the big matrix, U12 = W' A12 (one 64 x 64 x 64 product per workgroup), are tile
// products.  58 us -> ~20 us per step at n = 512.  The inverse also goes to Xd[j0 / 64] (column-major 64 x 64): the
// pipelined triangular solves need exactly these blocks (k_tri_diaginv is then skipped).
constexpr size_t CHP_LDS = (size_t)(3 * S64_MAT + S64_TMP) * sizeof(double);
__global__ void __launch_bounds__(256)
k_chol_panel_mfma(double *__restrict__ C, int n, int j0, int *__restrict__ info, double *__restrict__ Ds, double *__restrict__ Xd) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *M1 = sm, *M2 = sm + S64_MAT, *M3 = sm + 2 * S64_MAT, *T = sm + 3 * S64_MAT;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int nb = min(NB, n - j0);
    if (*info != 0) return;  // an earlier panel failed
    const int c0 = j0 + nb + blockIdx.x * NB;
    {   // diagonal block (upper triangle; identity padding past nb) and this workgroup's 64 columns of the row panel
        double g[16], x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e & 63, cidx = e >> 6;
            const bool in = r < nb && cidx < nb && r <= cidx;
            g[q] = in ? C[(size_t)(j0 + cidx) * n + j0 + r] : (r == cidx ? 1.0 : 0.0);
            x[q] = (r < nb && c0 + cidx < n) ? C[(size_t)(c0 + cidx) * n + j0 + r] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e & 63, cidx = e >> 6;
            M1[r * S64_LS + cidx] = g[q];
            M3[r * S64_LS + cidx] = x[q];       // A12 chunk, [k = row of the panel][j = column]
        }
    }
    __syncthreads();
    const int bad = s64_chol(M1, M2, &s_fail, tid);
    if (bad) {
        if (tid == 0 && blockIdx.x == 0) *info = j0 + bad;   // PosDefException position (1-based)
        return;
    }
    s64_chol_inverse(M1, M2, T, tid);           // M2 = inv(U11)
    if (blockIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e & 63, cidx = e >> 6;
            if (r <= cidx && r < nb && cidx < nb) Ds[(size_t)(j0 / NB) * NB * NB + (size_t)cidx * NB + r] = M1[r * S64_LS + cidx];
            if (Xd) Xd[(size_t)(j0 / NB) * NB * NB + (size_t)cidx * NB + r] = (r < nb && cidx < nb) ? M2[r * S64_LS + cidx] : 0.0;
        }
    }
    if (c0 >= n) return;   // last panel: nothing to the right
    __syncthreads();       // (workgroup 0 has parked U11: M1 is free now)
    s64_gemm<true, false, S64_LF>(M1, M2, M3, 1.0, tid);          // U12 chunk = inv(U11)' A12 chunk
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, r = e & 63, cidx = e >> 6;
        if (r < nb && c0 + cidx < n) C[(size_t)(c0 + cidx) * n + j0 + r] = M1[r * S64_LS + cidx];
    }
}

// ---- the whole blocked factorisation as ONE launch for n <= 64 * 22 (at most one 64 x 64 upper tile per CU) ------------
// Workgroup (i, j), i <= j, owns tile (i, j) of J'J + D and keeps it in LDS from the first to the last instruction:
//     for k < i:   wait for U(k, i) and U(k, j);   tile -= U(k, i)' U(k, j)                 (right-looking, MFMA)
//     i == j:      U(i, i) = chol(tile), inv(U(i, i)) -> Xd[i]; publish
//     i <  j:      wait for (i, i);   U(i, j) = inv(U(i, i))' tile; publish
// "publish" = the tile goes to global memory with agent-scope stores, every wave drains them, then one epoch-tagged flag
// is released (flags are never reset: the epoch changes with every factorisation).  Workgroup (i, j) only ever waits for
// tiles of rows < i or for (i, i), so the waits cannot form a cycle as long as every workgroup is resident -- one per CU
// here (the caller checks tiles <= CUs).  Waits are bounded (CHT_SPIN_LIMIT polls): a workgroup that gives up writes
// info = -1, releases its own flag so that nobody waits for IT, and the host repeats the factorisation with the
// launch-per-panel path.  Against that path (8 x (24 us panel + 10 us update + launch gaps) at n = 512) the chain per
// 64 columns is: factor + inverse 14 us, one flag, 5 us row-panel tile, one flag, 4 us update of the next diagonal tile.
constexpr size_t CHT_LDS = (size_t)(3 * S64_MAT + S64_TMP) * sizeof(double);
constexpr size_t CHC_LDS = (size_t)(4 * S64_MAT + S64_TMP) * sizeof(double);   // k_chol_chain: + the next diagonal tile
constexpr int CHT_SPIN_LIMIT = 1 << 22;
__device__ __forceinline__ bool cht_wait(const unsigned *flag, unsigned epoch, int *info, int spin_limit,
                                         const unsigned *flag2 = nullptr) {   // flag2: a second flag to wait for (same bound)
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 1, spins = 0;
        const unsigned *f2 = flag2 ? flag2 : flag;       // (both loads are in flight together)
        for (;;) {
            const unsigned a = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned b = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == epoch && b == epoch) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) {
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) atomicExch(info, -1);
        s_ok = ok;
    }
    __syncthreads();
    const bool r = s_ok != 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave, not only the polling one
    __syncthreads();
    return r;
}
// tile (rows 64 ti.., columns 64 tj..) of the column-major n x n matrix C -> LDS image [r][c]; entries outside the matrix:
// identity on a diagonal tile, zero elsewhere
// PUBLISHED = the tile was written by ANOTHER workgroup of this launch (cht_publish): agent-scope atomic loads, so that
// neither the compiler (no invariance / no-alias assumption) nor a non-coherent cache level can serve a stale value
template <bool PUBLISHED>
__device__ __forceinline__ void cht_fetch(double (&g)[16], const double *C, int n, int ti, int tj, int tid) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, r = e & 63, c = e >> 6;
        const int gr = 64 * ti + r, gc = 64 * tj + c;
        if (gr < n && gc < n)
            g[q] = PUBLISHED ? __hip_atomic_load(C + (size_t)gc * n + gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : C[(size_t)gc * n + gr];
        else
            g[q] = (ti == tj && r == c) ? 1.0 : 0.0;
    }
}
__device__ __forceinline__ void cht_stash(double *M, const double (&g)[16], int tid) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, r = e & 63, c = e >> 6;
        M[r * S64_LS + c] = g[q];
    }
}
template <bool PUBLISHED>
__device__ __forceinline__ void cht_load(double *M, const double *C, int n, int ti, int tj, int tid) {
    double g[16];
    cht_fetch<PUBLISHED>(g, C, n, ti, tj, tid);
    cht_stash(M, g, tid);
}
// the stores of cht_publish without the drain and the flag (cht_release later)
__device__ __forceinline__ void cht_store(double *C, int n, int ti, int tj, const double *M, bool upper_only, int tid) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, r = e & 63, c = e >> 6;
        const int gr = 64 * ti + r, gc = 64 * tj + c;
        if (gr < n && gc < n && (!upper_only || r <= c))
            __hip_atomic_store(C + (size_t)gc * n + gr, M[r * S64_LS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void cht_release(unsigned *flag, unsigned epoch, int tid) {   // every wave drains, then one flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cht_publish(double *C, int n, int ti, int tj, const double *M, bool upper_only,
                                            unsigned *flag, unsigned epoch, int tid) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, r = e & 63, c = e >> 6;
        const int gr = 64 * ti + r, gc = 64 * tj + c;
        if (gr < n && gc < n && (!upper_only || r <= c))
            __hip_atomic_store(C + (size_t)gc * n + gr, M[r * S64_LS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256)
k_chol_tiles(double *C, int n, int nt, int *info, double *Xd, unsigned *flags,   // (no __restrict__: workgroups exchange tiles through C and Xd)
             unsigned epoch, unsigned wait_epoch, int spin_limit) {   // (wait_epoch != epoch: the tests' fault injector)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *M0 = sm, *M1 = sm + S64_MAT, *M2 = sm + 2 * S64_MAT, *T = sm + 3 * S64_MAT;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int t = blockIdx.x, ti = 0;
    while (t >= nt - ti) { t -= nt - ti; ++ti; }
    const int tj = ti + t;
    unsigned *myflag = flags + ti * nt + tj;
    cht_load<false>(M0, C, n, ti, tj, tid);
    __syncthreads();
    bool ok = true;
    for (int k = 0; k < ti && ok; ++k) {
        ok = cht_wait(flags + k * nt + ti, wait_epoch, info, spin_limit);
        if (ok && tj != ti) ok = cht_wait(flags + k * nt + tj, wait_epoch, info, spin_limit);
        if (!ok) break;
        cht_load<true>(M1, C, n, k, ti, tid);               // U(k, i): rows = the k index
        if (tj != ti) cht_load<true>(M2, C, n, k, tj, tid);
        __syncthreads();
        const double *B = tj != ti ? M2 : M1;
        for (int q = wv; q < 16; q += 4) {                  // tile -= U(k, i)' U(k, j)
            const int a = q >> 2, b = q & 3;
            if (ti == tj && a > b) continue;                // (only the upper triangle of a diagonal tile is read later)
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<true, false>(acc, M1, 0, 16 * a, B, 0, 16 * b, 4, lane);
            s64_tile_store<true>(M0, 16 * a, 16 * b, acc, -1.0, lane);
        }
        __syncthreads();
    }
    if (!ok) {   // a tile this one needs never arrived: let the tiles waiting for THIS one go (the host sees info = -1)
        if (tid == 0) __hip_atomic_store(myflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (ti == tj) {
        const int bad = s64_chol(M0, M1, &s_fail, tid);
        if (bad) {
            if (tid == 0) {
                atomicCAS(info, 0, 64 * ti + bad);          // PosDefException position (1-based); the first failure wins
                __hip_atomic_store(myflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        s64_chol_inverse(M0, M1, T, tid);                   // M1 = inv(U(i, i))
#pragma unroll
        for (int q = 0; q < 16; ++q) {                      // the pipelined triangular solves and the row panel read Xd[i]
            const int e = tid + 256 * q, r = e & 63, c = e >> 6;
            const bool in = 64 * ti + r < n && 64 * ti + c < n;
            __hip_atomic_store(Xd + (size_t)ti * 4096 + (size_t)c * 64 + r, in ? M1[r * S64_LS + c] : 0.0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        cht_publish(C, n, ti, tj, M0, true, myflag, epoch, tid);
    } else {
        if (!cht_wait(flags + ti * nt + ti, wait_epoch, info, spin_limit)) {
            if (tid == 0) __hip_atomic_store(myflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        {   // inv(U(i, i)) (column-major 64 x 64 in Xd) -> M1 [r][c]
            double g[16];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                g[q] = __hip_atomic_load(Xd + (size_t)ti * 4096 + tid + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + 256 * q, r = e & 63, c = e >> 6;
                M1[r * S64_LS + c] = g[q];
            }
        }
        __syncthreads();
        s64_gemm<true, false, S64_LF>(M2, M1, M0, 1.0, tid);   // U(i, j) = inv(U(i, i))' tile
        cht_publish(C, n, ti, tj, M2, false, myflag, epoch, tid);
    }
}

// ---- the same factorisation with the DIAGONAL CHAIN inside one workgroup ----------------------------------------------
// In k_chol_tiles every 64 columns of the critical path cross workgroups twice: (i, i) -> flag -> (i, i+1) -> flag ->
// (i+1, i+1), each hop a drain of the stores, a release, a poll and a reload of a 32 KB tile (~26 us per 64 columns for
// ~14 us of arithmetic).  Here workgroup 0 (the chain) keeps the whole path local:
//     chain, step i:   M0 = T2'(i) - U(i-1, i)' U(i-1, i)   (T2' = the diagonal tile with the terms k <= i-2 applied, published
//                                                             by the helper of (i, i); U(i-1, i) is still in this LDS)
//                      U(i, i) = chol(M0);  publish it with the four inv(U_kk)' blocks (Wd)              -> flag D(i)
//                      T1'(i) (tile (i, i+1) with every term k < i applied, published by ITS helper)  -> U(i, i+1) by
//                      forward block substitution in registers;  publish                                 -> flag F(i, i+1)
//     helper (i, i):   accumulates k <= i-2, publishes T2' (flag P); later, off the path, turns D(i) into the full inverse
//                      Xd[i] the pipelined triangular solves read
//     helper (i, i+1): accumulates k < i, publishes T1' (flag P)
//     tile (i, j>i+1): as before, but it waits for D(i) only and substitutes with the 16 x 16 inverse blocks (no full inverse
//                      on anybody's path)
// The helpers' inputs are ready a whole chain step early (the chain is the slowest producer), so the chain's own waits are
// normally satisfied at once.  Deadlock: the chain waits for helpers of rows <= i, they wait for tiles of rows < i and for
// D(<= i); no cycle, all workgroups resident (tiles + 1 <= CUs).  Giving up / a non-positive pivot: as in k_chol_tiles,
// plus the chain releases every flag it owns.
// X (64 x 64, LDS) <- inv(U)' X for an upper triangular U whose inv(U_kk)' blocks are the diagonal blocks of W: forward
// substitution over the four block rows; wavefront = 16 columns, which it carries through all four stages in registers --
// the accumulator layout of v_mfma_f64_16x16x4 (lane (j, q) holds rows q + 4 r) IS the B-operand layout of the next product
// when its k index is taken as 4 kk + q, so nothing goes back through LDS between the stages.
__device__ __forceinline__ void s64_trsm_blocks(double *X, const double *U, const double *W, int tid) {
    const int lane = tid & 63, c = tid >> 6, ij = lane & 15, kq = lane >> 4;
    s64_v4d x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s64_v4d t;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) t[rr] = X[(16 * r + kq + 4 * rr) * S64_LS + 16 * c + ij];
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
    __syncthreads();                                  // (every wavefront has read its columns of X)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) X[(16 * r + kq + 4 * rr) * S64_LS + 16 * c + ij] = x[r][rr];
    __syncthreads();
}
// 16 x 16 tile (a, b) of M -= B'B (B: 64 x 64 in LDS), one wavefront; the operands are loaded eight k-steps ahead of the products
__device__ __forceinline__ void s64_syrk_tile_sub(double *M, const double *B, int a, int b, int lane) {
    const int ij = lane & 15, kq = lane >> 4;
    s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double x[8], y[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 32 * h + 4 * kk + kq;
            x[kk] = B[k * S64_LS + 16 * a + ij];
            y[kk] = B[k * S64_LS + 16 * b + ij];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[kk], y[kk], acc, 0, 0, 0);
    }
    s64_tile_store<true>(M, 16 * a, 16 * b, acc, -1.0, lane);
}
// the four inv(U_kk)' blocks of a diagonal tile: LDS image (diagonal block positions of W) <-> Wd[tile][kb][16 x 16]
__device__ __forceinline__ void chc_store_w(double *Wd, int i, const double *W, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, o = 16 * (e >> 8), r = (e >> 4) & 15, c = e & 15;
        __hip_atomic_store(Wd + (size_t)i * 1024 + e, W[(o + r) * S64_LS + o + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// a published tile fetched / put into LDS by 192 threads (wavefronts 1..3 while wavefront 0 factors a diagonal block)
__device__ __forceinline__ void chc_fetch192(double (&g)[22], const double *C, int n, int ti, int tj, int t) {
    // element e = t + 192 q: row t & 63 (192 = 3 x 64), column (t >> 6) + 3 q -- one base address, one stride
    const int r = t & 63, c0 = t >> 6, gr = 64 * ti + r;
    const double *p = C + (size_t)(64 * tj + c0) * n + gr;
    const size_t step = (size_t)3 * n;
    const int cleft = n - (64 * tj + c0);                        // columns c0 + 3 q with 3 q < cleft are inside the matrix
#pragma unroll
    for (int q = 0; q < 22; ++q) {
        const int c = c0 + 3 * q;
        g[q] = (ti == tj && r == c) ? 1.0 : 0.0;
        if (c < 64 && gr < n && 3 * q < cleft) g[q] = __hip_atomic_load(p + q * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void chc_stash192(double *M, const double (&g)[22], int t) {
    const int r = t & 63, c0 = t >> 6;
#pragma unroll
    for (int q = 0; q < 22; ++q)
        if (c0 + 3 * q < 64) M[r * S64_LS + c0 + 3 * q] = g[q];
}
// sixteen rows (from R0) of the diagonal tile (i, i) and inverse block kb, stored by `nthreads` threads (t = 0..nthreads-1)
__device__ __forceinline__ void chc_store_rows(double *C, int n, int i, const double *M, int R0, int t, int nthreads) {
    for (int e = t; e < 1024; e += nthreads) {
        const int r = R0 + (e & 15), c = e >> 4;
        const int gr = 64 * i + r, gc = 64 * i + c;
        if (gr < n && gc < n && r <= c)
            __hip_atomic_store(C + (size_t)gc * n + gr, M[r * S64_LS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void chc_store_wblock(double *Wd, int i, const double *W, int kb, int t, int nthreads) {
    for (int e = t; e < 256; e += nthreads)
        __hip_atomic_store(Wd + (size_t)i * 1024 + kb * 256 + e, W[(16 * kb + (e >> 4)) * S64_LS + 16 * kb + (e & 15)], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chc_load_w(double *W, const double *Wd, int i, int tid) {
    double g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = __hip_atomic_load(Wd + (size_t)i * 1024 + tid + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, o = 16 * (e >> 8), r = (e >> 4) & 15, c = e & 15;
        W[(o + r) * S64_LS + o + c] = g[q];
    }
}

__global__ void __launch_bounds__(256)
k_chol_chain(double *C, int n, int nt, int *info, double *Xd, double *Wd, unsigned *flags, unsigned *pflags,
             unsigned epoch, unsigned wait_epoch, int spin_limit, long long *trace,   // trace: LSQ_CHOL_TRACE (10 ns ticks)
             const double *bvec, double *zvec, unsigned long long *zslot, unsigned zep, int *zerr) {   // bvec: + U'z = b (see below)
#define CHC_STAMP(p) do { if (trace && tid == 0) trace[i * 16 + (p)] = wall_clock64(); } while (0)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *M0 = sm, *M1 = sm + S64_MAT, *M2 = sm + 2 * S64_MAT, *T = sm + 3 * S64_MAT;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x == 0) {                                   // ---- the chain
        // step i:  chol(i); in the shadow of its diagonal blocks (wavefronts 1..3 are idle while wavefront 0 factors one):
        //            block 0: the six tiles of M0 -= U(i-1, i)'U(i-1, i) that block 0 does not touch
        //            block k: the rows of U(i, i) and the inverse block that block k-1 finished go to memory
        //            block 2: wait for T2'(i+1) (published a chain step ago, normally) and bring it into the other diagonal buffer
        //            block 3: the same for T1'(i)
        //          last rows of D(i), D(i) released | U(i, i+1) by substitution | its stores | block row 0 of
        //          T2'(i+1) -= U(i, i+1)'U(i, i+1) | F(i, i+1) released
        // The diagonal tile being factored (Mc) and the next one (Mn, filled in a shadow) swap roles every step.
        __shared__ int s_pfail;
        double *Mc = M0, *Mn = sm + 3 * S64_MAT + S64_TMP;
        bool ok = true;
        cht_load<false>(Mc, C, n, 0, 0, tid);
        if (tid == 0) s_pfail = 0;
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            CHC_STAMP(0);
            const bool pending = i > 0, more = i + 1 < nt;
            const int bad = s64_chol<false>(Mc, M1, &s_fail, tid, trace ? trace + 512 + i * 16 : nullptr, [&](int kb, int w, int ln) {
                const int t = 64 * (w - 1) + ln;
                if (kb == 0) {
                    if (!pending) return;
                    if (w == 1) { s64_syrk_tile_sub(Mc, M2, 1, 1, ln); s64_syrk_tile_sub(Mc, M2, 1, 2, ln); }
                    if (w == 2) { s64_syrk_tile_sub(Mc, M2, 1, 3, ln); s64_syrk_tile_sub(Mc, M2, 2, 2, ln); }
                    if (w == 3) { s64_syrk_tile_sub(Mc, M2, 2, 3, ln); s64_syrk_tile_sub(Mc, M2, 3, 3, ln); }
                    return;
                }
                chc_store_rows(C, n, i, Mc, 16 * (kb - 1), t, 192);
                chc_store_wblock(Wd, i, M1, kb - 1, t, 192);
                if (kb == 1 || !more) return;
                // block 2: T2'(i+1) -> Mn;  block 3: T1'(i) -> M2 (U(i-1, i) in there was last read in the shadow of block 0)
                const unsigned *f = kb == 2 ? pflags + (i + 1) * nt + i + 1 : pflags + i * nt + i + 1;
                int got = 1;
                if (ln == 0) {                                   // (every wavefront polls for itself: no barrier in here)
                    int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > spin_limit) {
                            got = 0;
                            break;
                        }
                    }
                    if (!got) {
                        atomicExch(info, -1);
                        atomicOr(&s_pfail, 1);
                    }
                }
                got = __builtin_amdgcn_readfirstlane(got);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (!got) return;
                double g[22];
                chc_fetch192(g, C, n, kb == 2 ? i + 1 : i, i + 1, t);
                chc_stash192(kb == 2 ? Mn : M2, g, t);
            });
            if (bad) {
                if (tid == 0) atomicCAS(info, 0, 64 * i + bad);   // PosDefException position (1-based)
                ok = false;
                break;
            }
            if (s_pfail) {                                       // (uniform: read after the barriers that end s64_chol)
                ok = false;
                break;
            }
            CHC_STAMP(1);
            chc_store_rows(C, n, i, Mc, 48, tid, 256);
            chc_store_wblock(Wd, i, M1, 3, tid, 256);
            cht_release(flags + i * nt + i, epoch, tid);          // D(i)
            if (!more) break;
            CHC_STAMP(3);
            s64_trsm_blocks(M2, Mc, M1, tid);
            CHC_STAMP(4);
            cht_store(C, n, i, i + 1, M2, false, tid);
            CHC_STAMP(6);
            s64_syrk_tile_sub(Mn, M2, 0, wv, lane);               // block row 0 now; the rest inside the next chol
            CHC_STAMP(5);
            cht_release(flags + i * nt + i + 1, epoch, tid);      // F(i, i+1)
            __syncthreads();
            double *sw = Mc;
            Mc = Mn;
            Mn = sw;
        }
#undef CHC_STAMP
        if (!ok && tid < nt) {     // nobody may be left waiting for the chain
            __hip_atomic_store(flags + tid * nt + tid, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (tid + 1 < nt) __hip_atomic_store(flags + tid * nt + tid + 1, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if ((int)blockIdx.x > nt * (nt + 1) / 2) {               // ---- U'z = b behind the factorisation (k_tri_fsolve_t's scheme)
        // workgroup t owns unknowns 64 t ..: for e < t it takes tile U(e, t) as soon as its flag is up, z_e as soon as the
        // workgroup of block e has published it (flag-in-data slots), subtracts U(e, t)'z_e; X(t) -- the inverse of the
        // diagonal tile, from the helper of (t, t) -- finishes z_t.  It trails the chain by the helper's inversion + one
        // product, instead of a launch + 8 x 3.7 us after it.
        const int t = (int)blockIdx.x - nt * (nt + 1) / 2 - 1;
        double *sc = sm, *sz = sm + 64, *sp = sm + 128;      // sp[4][64]
        const int col = tid & 63, part = tid >> 6, c0 = 64 * t;
        const bool cin = c0 + col < n;
        if (tid < 64) sc[tid] = cin ? bvec[c0 + tid] : 0.0;
        double tile[16];
        auto apply = [&](double sign) {                      // sc += sign * tile' * sz
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += tile[q] * sz[part * 16 + q];
            sp[part * 64 + col] = acc;
            __syncthreads();
            if (tid < 64) sc[tid] += sign * (((sp[tid] + sp[64 + tid]) + sp[128 + tid]) + sp[192 + tid]);
            __syncthreads();
        };
        for (int e = 0; e < t; ++e) {
            if (!cht_wait(flags + e * nt + t, wait_epoch, info, spin_limit)) return;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                tile[q] = cin ? __hip_atomic_load(C + (size_t)(c0 + col) * n + e * 64 + part * 16 + q, RLX_AGENT) : 0.0;
            if (tid < 64) {
                const unsigned long long *f = zslot + ((size_t)e * 64 + tid) * 2;
                unsigned long long w0 = __hip_atomic_load(f, RLX_AGENT), w1 = __hip_atomic_load(f + 1, RLX_AGENT);
                int spins = 0;
                while ((unsigned)(w0 >> 32) != zep || (unsigned)(w1 >> 32) != zep) {
                    if (++spins > spin_limit) { atomicOr(zerr, 1); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); break; }
                    __builtin_amdgcn_s_sleep(1);
                    w0 = __hip_atomic_load(f, RLX_AGENT);
                    w1 = __hip_atomic_load(f + 1, RLX_AGENT);
                }
                sz[tid] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
            }
            __syncthreads();
            apply(-1.0);
        }
        if (!cht_wait(pflags + t * nt, wait_epoch, info, spin_limit)) return;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = part * 16 + q;                     // X_tt(r, col)
            tile[q] = (cin && c0 + r < n) ? __hip_atomic_load(Xd + (size_t)t * 4096 + (size_t)col * 64 + r, RLX_AGENT) : 0.0;
        }
        if (tid < 64) { sz[tid] = sc[tid]; sc[tid] = 0.0; }
        __syncthreads();
        apply(1.0);
        if (tid < 64) {
            const double v = sc[tid];
            unsigned long long *mine = zslot + ((size_t)t * 64 + tid) * 2;
            const unsigned long long hi = (unsigned long long)zep << 32;
            __hip_atomic_store(mine, hi | (unsigned)__double2loint(v), RLX_AGENT);
            __hip_atomic_store(mine + 1, hi | (unsigned)__double2hiint(v), RLX_AGENT);
            if (cin) zvec[c0 + tid] = v;
        }
        return;
    }
    int t = blockIdx.x - 1, ti = 0;
    while (t >= nt - ti) { t -= nt - ti; ++ti; }
    const int tj = ti + t;
    const bool diag = tj == ti, sup = tj == ti + 1;
    unsigned *myflag = (diag || sup) ? pflags + ti * nt + tj : flags + ti * nt + tj;
    cht_load<false>(M0, C, n, ti, tj, tid);
    __syncthreads();
    bool ok = true;
    const int kend = diag ? ti - 1 : ti;                     // (the chain applies k = i-1 to its diagonal tile itself)
    for (int k = 0; k < kend && ok; ++k) {
        ok = cht_wait(flags + k * nt + ti, wait_epoch, info, spin_limit);
        if (ok && !diag) ok = cht_wait(flags + k * nt + tj, wait_epoch, info, spin_limit);
        if (!ok) break;
        cht_load<true>(M1, C, n, k, ti, tid);
        if (!diag) cht_load<true>(M2, C, n, k, tj, tid);
        __syncthreads();
        const double *B = diag ? M1 : M2;
        for (int q = wv; q < 16; q += 4) {
            const int a = q >> 2, b = q & 3;
            if (diag && a > b) continue;
            s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
            s64_tile_mma<true, false>(acc, M1, 0, 16 * a, B, 0, 16 * b, 4, lane);
            s64_tile_store<true>(M0, 16 * a, 16 * b, acc, -1.0, lane);
        }
        __syncthreads();
    }
    if (!ok) {
        if (tid == 0) __hip_atomic_store(myflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (diag) {
        if (ti > 0) cht_publish(C, n, ti, ti, M0, true, myflag, epoch, tid);              // T2'(i)
        unsigned *xflag = pflags + ti * nt;                  // X(i): Xd[i] is complete (slot (i, 0) of the P flags is nobody's)
        if (!cht_wait(flags + ti * nt + ti, wait_epoch, info, spin_limit)) {
            if (tid == 0) __hip_atomic_store(xflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        cht_load<true>(M0, C, n, ti, ti, tid);               // U(i, i) (the blocks below the diagonal blocks are not read)
        chc_load_w(M1, Wd, ti, tid);
        __syncthreads();
        s64_chol_inverse(M0, M1, T, tid);                    // M1 = inv(U(i, i)): what the pipelined triangular solves read
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e & 63, c = e >> 6;
            const bool in = 64 * ti + r < n && 64 * ti + c < n;
            __hip_atomic_store(Xd + (size_t)ti * 4096 + (size_t)c * 64 + r, in ? M1[r * S64_LS + c] : 0.0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        cht_release(xflag, epoch, tid);
    } else if (sup) {
        cht_publish(C, n, ti, tj, M0, false, myflag, epoch, tid);                           // T1'(i): the chain finishes it
    } else {
        if (!cht_wait(flags + ti * nt + ti, wait_epoch, info, spin_limit)) {
            if (tid == 0) __hip_atomic_store(myflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        cht_load<true>(M1, C, n, ti, ti, tid);
        chc_load_w(M2, Wd, ti, tid);
        __syncthreads();
        s64_trsm_blocks(M0, M1, M2, tid);
        cht_publish(C, n, ti, tj, M0, false, myflag, epoch, tid);
    }
}

__global__ void __launch_bounds__(256)
k_chol_diag_restore(double *__restrict__ C, int n, const double *__restrict__ Ds, const int *__restrict__ info) {
    const int j0 = blockIdx.x * NB, nb = min(NB, n - j0);
    const int failed = *info;               // panels from the failing one on were never parked
    if (failed != 0 && j0 + NB >= failed) return;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e % NB, cidx = e / NB;
        if (r <= cidx && cidx < nb) C[(size_t)(j0 + cidx) * n + j0 + r] = Ds[(size_t)blockIdx.x * NB * NB + (size_t)cidx * NB + r];
    }
}

// blocked solves U'z = b then U x = z in ONE workgroup: per 64-block, one wavefront does the
// triangular part (lane r holds unknown r; shuffles broadcast each solved value), then all
// threads subtract the block's contribution from the rest of the right-hand side.
__global__ void __launch_bounds__(1024)
k_chol_trsv(const double *__restrict__ U, int n, double *__restrict__ b) {
    __shared__ double zb[NB];
    __shared__ double D[NB][NB + 1];   // the current diagonal block, D[r][c] = U[j0+r, j0+c]
    __shared__ double Dinv[NB];        // reciprocals of its diagonal (one division per row, off the serial chain)
    const int tid = threadIdx.x, lane = tid & 63;
    auto load_diag = [&](int j0, int nb) {
        for (int e = tid; e < NB * NB; e += 1024) {
            const int r = e % NB, cidx = e / NB;
            D[r][cidx] = (r < nb && cidx < nb && r <= cidx) ? U[(size_t)(j0 + cidx) * n + j0 + r] : 0.0;
        }
        if (tid < NB) Dinv[tid] = tid < nb ? 1.0 / U[(size_t)(j0 + tid) * n + j0 + tid] : 0.0;
        __syncthreads();
    };
    // forward: U' z = b   (row r of U' = column r of U: contiguous)
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int nb = min(NB, n - j0);
        load_diag(j0, nb);
        if (tid < 64) {
            // the lane's column of the block sits in registers: the 64-step dependent chain is then
            // shuffle + multiply + FMA per step, with no LDS read on it
            double dcol[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) dcol[r] = D[r][lane];
            double v = lane < nb ? b[j0 + lane] : 0.0;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double zr = __shfl(v, r, 64) * Dinv[r];
                if (r < nb) {
                    if (lane == r) v = zr;
                    else if (lane > r && lane < nb) v -= dcol[r] * zr;
                }
            }
            if (lane < nb) { b[j0 + lane] = v; zb[lane] = v; }
        }
        __syncthreads();
        for (int i = j0 + nb + tid; i < n; i += 1024) {   // b_i -= sum_r U[j0+r, i] z_r  (column i, rows j0..)
            const double *ci = U + (size_t)i * n + j0;
            double s = 0.0;
            for (int r = 0; r < nb; ++r) s += ci[r] * zb[r];
            b[i] -= s;
        }
        __syncthreads();
    }
    // backward: U x = z
    for (int jb = (n - 1) / NB; jb >= 0; --jb) {
        const int j0 = jb * NB, nb = min(NB, n - j0);
        load_diag(j0, nb);
        if (tid < 64) {
            double drow[NB];   // D[lane][r]: the lane's row of the block
#pragma unroll
            for (int r = 0; r < NB; ++r) drow[r] = D[lane][r];
            double v = lane < nb ? b[j0 + lane] : 0.0;
#pragma unroll
            for (int r = NB - 1; r >= 0; --r) {
                const double xr = __shfl(v, r, 64) * Dinv[r];
                if (r < nb) {
                    if (lane == r) v = xr;
                    else if (lane < r) v -= drow[r] * xr;
                }
            }
            if (lane < nb) { b[j0 + lane] = v; zb[lane] = v; }
        }
        __syncthreads();
        for (int i = tid; i < j0; i += 1024) {            // b_i -= sum_r U[i, j0+r] x_r
            double s = 0.0;
            for (int r = 0; r < nb; ++r) s += U[(size_t)(j0 + r) * n + i] * zb[r];
            b[i] -= s;
        }
        __syncthreads();
    }
}

// J'J for FEW columns (n <= 32) and many rows: the 64 x 64 MFMA tile would compute 4096 products per row for
// n(n+1)/2 useful ones, so here a workgroup streams a window of rows through LDS (32 rows at a time) and every
// thread owns up to three (i, j) pairs of the upper triangle; per-window partial sums land in W[window][pair]
// and k_syrk_small_reduce adds the windows in index order (+ damp on the diagonal).  Memory-bound: one pass over J.
constexpr int SS_ROWS = 64;
constexpr int SS_MAXN = 32;
constexpr int SS_PPT = (SS_MAXN * (SS_MAXN + 1) / 2 + 255) / 256;   // pairs per thread (3)
__global__ void __launch_bounds__(256)
k_syrk_small(const double *__restrict__ A, int m, int n, int wrows, double *__restrict__ W) {
    __shared__ double sA[SS_ROWS][SS_MAXN + 1];
    const int tid = threadIdx.x;
    const int npairs = n * (n + 1) / 2;
    int pi[SS_PPT], pj[SS_PPT];
#pragma unroll
    for (int q = 0; q < SS_PPT; ++q) {       // pair index -> (i <= j), column-major upper triangle: p = j(j+1)/2 + i
        const int p = tid + q * 256;
        int j = 0;
        while ((j + 1) * (j + 2) / 2 <= p && j + 1 < n) ++j;
        pj[q] = j;
        pi[q] = min(p - j * (j + 1) / 2, j);
    }
    double acc[SS_PPT] = {0.0, 0.0, 0.0};
    const int r0 = blockIdx.x * wrows, r1 = min(m, r0 + wrows);
    for (int rb = r0; rb < r1; rb += SS_ROWS) {
        for (int e = tid; e < SS_ROWS * n; e += 256) {
            const int r = e % SS_ROWS, cidx = e / SS_ROWS;          // (rows contiguous in memory: coalesced per column)
            sA[r][cidx] = rb + r < r1 ? A[(size_t)cidx * m + rb + r] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SS_PPT; ++q) {
            if (tid + q * 256 < npairs) {
                double a = acc[q];
#pragma unroll 8
                for (int r = 0; r < SS_ROWS; ++r) a += sA[r][pi[q]] * sA[r][pj[q]];
                acc[q] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < SS_PPT; ++q)
        if (tid + q * 256 < npairs) W[(size_t)blockIdx.x * npairs + tid + q * 256] = acc[q];
}
__global__ void __launch_bounds__(256)
k_syrk_small_reduce(const double *__restrict__ W, int n, int nwin, const double *__restrict__ damp, double *__restrict__ C,
                    int *__restrict__ info) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *info = 0;
    const int npairs = n * (n + 1) / 2;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npairs) return;
    int j = 0;
    while ((j + 1) * (j + 2) / 2 <= p) ++j;
    const int i = p - j * (j + 1) / 2;
    double s0 = 0.0;
    int w = 0;
    for (; w + 8 <= nwin; w += 8) {           // eight loads in flight, added in window order
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = W[(size_t)(w + u) * npairs + p];
#pragma unroll
        for (int u = 0; u < 8; ++u) s0 += t[u];
    }
    for (; w < nwin; ++w) s0 += W[(size_t)w * npairs + p];
    if (i == j && damp) s0 += damp[i];
    C[(size_t)j * n + i] = s0;
}

// largest diagonal entry of the n x n matrix C (one workgroup): the reference point of the full-rank certificate
__global__ void __launch_bounds__(256)
k_diag_max(const double *__restrict__ C, int n, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) v = fmax(v, C[(size_t)j * n + j]);
    v = block_max<256>(v, sh);
    if (threadIdx.x == 0) *out = v;
}

// dense_cholesky.jl:43-59: returns LSQ_ENOTPD through *info like the small kernel.  d_x == nullptr: factor only
// (the caller decides about the solves); d_dmax != nullptr: also max_j (J'J + damp)_jj before the factorisation.
// d_y: the right-hand side is J'y and d_x does not hold it yet (it is formed here, inside the SYRK launch where that applies)
int lsq_cholesky_blocked(lsq_solver *s, lsq_mat *J, const double *d_damp, double *d_x, double *d_dmax, bool allow_tiles,
                         const double *d_y) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n;
    const int nt = (n + MT - 1) / MT, ntiles = nt * (nt + 1) / 2;
    // split-K slices: enough workgroups for ~6 per CU when there are many tiles (their barriers and LDS phases interleave:
    // 42 TFLOP/s at 16384 x 2048), ~2 per CU when the slice reduction would otherwise dominate (4096 x 512)
    const int occ = ntiles >= 128 ? 6 : 2;
    int kslices = std::max(1, std::min(512, (occ * c->num_cus + ntiles - 1) / ntiles));
    kslices = std::min(kslices, std::max(1, m / (4 * KC)));
    const size_t need = (size_t)kslices * ntiles * MT * MT;
    if (s->work_elems < need) {
        hipFree(s->d_T);
        LSQ_HIP(hipMalloc(&s->d_T, need * sizeof(double)));
        s->work_elems = need;
    }
    const bool small_syrk = n <= SS_MAXN && m >= 16384;
    const bool jty_rides = d_y && d_x && !small_syrk && lsq_dense_t_windows(c, m, n) == 0 && !getenv("LSQ_CHOL_SEPARATE_JTY");
    if (d_y && d_x && !jty_rides) LSQ_TRY(lsq_dense_mul(J, 1, 1.0, d_y, 0.0, d_x));  // mul!(x, J', y)
    if (small_syrk) {
        // few columns, many rows: the pair kernel (one pass over J, no 64 x 64 tile of mostly padding)
        const int nwin = std::max(1, std::min(2 * c->num_cus, m / 1024));
        const int wrows = ((m + nwin - 1) / nwin + SS_ROWS - 1) / SS_ROWS * SS_ROWS;
        const int npairs = n * (n + 1) / 2;
        const size_t need2 = (size_t)nwin * npairs;
        if (s->work_elems < need2) {
            hipFree(s->d_T);
            LSQ_HIP(hipMalloc(&s->d_T, need2 * sizeof(double)));
            s->work_elems = need2;
        }
        const int nw = (m + wrows - 1) / wrows;
        LSQ_LAUNCH(k_syrk_small, dim3(nw), dim3(256), 0, c->stream, J->d_dense, m, n, wrows, s->d_T);
        LSQ_LAUNCH(k_syrk_small_reduce, dim3((npairs + 255) / 256), dim3(256), 0, c->stream, s->d_T, n, nw, d_damp, s->d_chol,
                           s->d_info);
    } else {
        LSQ_LAUNCH((k_syrk_mfma<0>), dim3(ntiles * kslices + (jty_rides ? n : 0)), dim3(256), 0, c->stream, J->d_dense, m, m,
                           n, 0, kslices, s->d_T, (double *)nullptr, 0, jty_rides ? d_y : nullptr, jty_rides ? d_x : nullptr);
        LSQ_LAUNCH(k_syrk_reduce, dim3(ntiles * 16), dim3(256), 0, c->stream, s->d_T, n, kslices, d_damp, s->d_chol, s->d_info);
    }
    if (d_dmax) LSQ_LAUNCH(k_diag_max, dim3(1), dim3(256), 0, c->stream, s->d_chol, n, d_dmax);
    // one launch for the whole factorisation when every 64 x 64 upper tile gets a CU of its own (k_chol_tiles)
    if (allow_tiles && !s->fb_tiles.off() && ntiles <= c->num_cus && n >= 2 * NB && !getenv("LSQ_CHOL_PANELS")) {
        double *Xt = lsq_tri_chol_diagbuf(s, n);
        if (Xt) {
            // flags: [0, 1024) 'tile final' (D on the diagonal), [1024, 2048) 'partial tile for the chain'; then the
            // 16 x 16 inverse blocks of the diagonal tiles (k_chol_chain)
            constexpr size_t FLAG_BYTES = 2 * 32 * 32 * sizeof(unsigned);
            if (!s->d_chol_flags) {
                LSQ_HIP(hipMalloc(&s->d_chol_flags, FLAG_BYTES + 32 * 1024 * sizeof(double)));
                LSQ_ZERO(s->d_chol_flags, 0, FLAG_BYTES);   // (waits for the memset: see LSQ_ZERO)
            }
            if (++s->chol_epoch == 0) ++s->chol_epoch;
            const bool inject = getenv("LSQ_TEST_EXCHANGE_TIMEOUT") != nullptr;   // the waits never see their flag
            const unsigned wait_epoch = inject ? s->chol_epoch ^ 0x40000000u : s->chol_epoch;
            static const bool v1 = getenv("LSQ_CHOL_TILES_V1") != nullptr;      // (A/B: the chain across workgroups)
            if (!v1 && ntiles + 1 <= c->num_cus) {
                LSQ_TRY(lsq_set_lds(c, (const void *)k_chol_chain, CHC_LDS));
                static const bool tracing = getenv("LSQ_CHOL_TRACE") != nullptr;   // (debug: the chain's phase stamps)
                static long long *d_trace = nullptr;
                if (tracing && !d_trace) { LSQ_HIP(hipMalloc(&d_trace, 64 * 16 * sizeof(long long))); }
                // with a right-hand side at hand the forward half of the solve rides along: nt more workgroups
                double *zv = nullptr;
                unsigned long long *zslot = nullptr, zep = 0;
                int *zerr = nullptr;
                static const bool nofuse = getenv("LSQ_CHOL_NO_FUSED_FSOLVE") != nullptr;     // (A/B)
                const bool fuse = d_x && !nofuse && ntiles + 1 + nt <= c->num_cus &&
                                  lsq_tri_chol_fwd_operands(s, n, &zv, &zslot, &zep, &zerr) == LSQ_OK;
                LSQ_LAUNCH(k_chol_chain, dim3(ntiles + 1 + (fuse ? nt : 0)), dim3(256), CHC_LDS, c->stream, s->d_chol, n, nt,
                                   s->d_info, Xt, (double *)((char *)s->d_chol_flags + FLAG_BYTES), s->d_chol_flags,
                                   s->d_chol_flags + 1024, s->chol_epoch, wait_epoch, inject ? 64 : CHT_SPIN_LIMIT, d_trace,
                                   (const double *)(fuse ? d_x : nullptr), zv, zslot, (unsigned)zep, zerr);
                if (tracing) {
                    long long h[64 * 16];
                    LSQ_HIP(hipStreamSynchronize(c->stream));
                    LSQ_HIP(hipMemcpy(h, d_trace, sizeof h, hipMemcpyDeviceToHost));
                    static int shown = 0;
                    if (shown++ % 50 == 10)
                        for (int i = 0; i < nt; ++i) {
                            fprintf(stderr, "chain step %2d: chol %.2f", i, (h[i * 16 + 1] - h[i * 16]) * 0.01);
                            if (i + 1 < nt)
                                fprintf(stderr, " tail+release %.2f trsm %.2f store %.2f syrk %.2f   total %.2f us", (h[i * 16 + 3] - h[i * 16 + 1]) * 0.01,
                                        (h[i * 16 + 4] - h[i * 16 + 3]) * 0.01, (h[i * 16 + 6] - h[i * 16 + 4]) * 0.01,
                                        (h[i * 16 + 5] - h[i * 16 + 6]) * 0.01, (h[(i + 1) * 16] - h[i * 16]) * 0.01);
                            fprintf(stderr, "\n    chol:");
                            for (int p = 0; p < 12; ++p) fprintf(stderr, " %.2f", (h[512 + i * 16 + p] - (p ? h[512 + i * 16 + p - 1] : h[i * 16])) * 0.01);
                            fprintf(stderr, "\n");
                        }
                }
            } else {
                LSQ_TRY(lsq_set_lds(c, (const void *)k_chol_tiles, CHT_LDS));
                LSQ_LAUNCH(k_chol_tiles, dim3(ntiles), dim3(256), CHT_LDS, c->stream, s->d_chol, n, nt, s->d_info, Xt,
                                   s->d_chol_flags, s->chol_epoch, wait_epoch, inject ? 64 : CHT_SPIN_LIMIT);
            }
            s->chol_have_diaginv = true;
            s->last_chol_tiles = true;
            if (!d_x) { LSQ_HIP(hipGetLastError()); return LSQ_OK; }
            if (lsq_tri_chol_solve(s, s->d_chol, n, d_x) != LSQ_OK)
                LSQ_LAUNCH(k_chol_trsv, dim3(1), dim3(1024), 0, c->stream, s->d_chol, n, d_x);
            LSQ_HIP(hipGetLastError());
            return LSQ_OK;
        }
    }
    s->last_chol_tiles = false;
    // parking space for the factored diagonal blocks (k_chol_panel_mfma)
    bool merged = false;
    double *Ds = s->d_Ds;
    // the MFMA panel kernel leaves the inverted diagonal blocks where the pipelined triangular solves look for them
    double *Xd = nullptr;
    s->chol_have_diaginv = false;
    LSQ_TRY(lsq_set_lds(c, (const void *)k_chol_panel_mfma, CHP_LDS));
    Xd = lsq_tri_chol_diagbuf(s, n);
    s->chol_have_diaginv = Xd != nullptr;
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int nb = std::min(NB, n - j0), rest = n - j0 - nb;
        LSQ_LAUNCH(k_chol_panel_mfma, dim3(std::max(1, (rest + NB - 1) / NB)), dim3(256), CHP_LDS, c->stream, s->d_chol,
                           n, j0, s->d_info, Ds, Xd);
        merged = true;
        if (rest > 0) {
            const int nt2 = (rest + MT - 1) / MT;
            // A22 -= U12' U12 : "A" = rows j0..j0+nb of chol (lda n), columns from j0+nb
            LSQ_LAUNCH((k_syrk_mfma<1>), dim3(nt2 * (nt2 + 1) / 2), dim3(256), 0, c->stream, s->d_chol + j0, n, nb,
                               rest, j0 + nb, 1, (double *)nullptr, s->d_chol, n);
        }
    }
    if (merged)
        LSQ_LAUNCH(k_chol_diag_restore, dim3((n + NB - 1) / NB), dim3(256), 0, c->stream, s->d_chol, n, Ds,
                           (const int *)s->d_info);
    if (!d_x) { LSQ_HIP(hipGetLastError()); return LSQ_OK; }
    // U'z = b, U x = z: pipelined over the 64-blocks on several CUs; the single-workgroup kernel otherwise
    if (lsq_tri_chol_solve(s, s->d_chol, n, d_x) != LSQ_OK)
        LSQ_LAUNCH(k_chol_trsv, dim3(1), dim3(1024), 0, c->stream, s->d_chol, n, d_x);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// the two triangular solves on an already factored s->d_chol (b overwritten by x)
int lsq_cholesky_blocked_solve(lsq_solver *s, int n, double *d_x) {
    lsq_ctx *c = s->ctx;
    if (lsq_tri_chol_solve(s, s->d_chol, n, d_x) != LSQ_OK)
        LSQ_LAUNCH(k_chol_trsv, dim3(1), dim3(1024), 0, c->stream, s->d_chol, n, d_x);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
