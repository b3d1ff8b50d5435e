// CholeskyQR2 panel + basis-kernel block reflector for the blocked QR (lsq_qr_cholqr.hip); used by lsq_dense.hip
#pragma once
#include "lsq_common.h"
#include "lsq_small64.h"

constexpr int CQ_RS = 64;                  // rows of the panel per workgroup (256 workgroups at 16384 rows: every CU)
constexpr int CQ_QST = CQ_RS + 2;          // slab image [col][row], row stride (doubles)

// Gram of the slab image, UPPER 16 x 16 tiles only (10 of 16; the consumers read the upper triangle), 3 / 3 / 2 / 2 tiles
// per wavefront, K = CQ_RS; partial -> Gp (row-major 64 x 64; the strictly lower tiles stay as allocated: zero)
// AGENT: the partial is summed by ANOTHER workgroup of the same launch (cq_group_reduce): agent-scope stores.
template <bool AGENT = false>
__device__ __forceinline__ void cq_slab_gram(const double *__restrict__ Qs, double *__restrict__ Gp, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int ij = lane & 15, kq = lane >> 4;
    const int first = w < 2 ? 3 * w : 6 + 2 * (w - 2), count = w < 2 ? 3 : 2;
    for (int t = 0; t < count; ++t) {
        const int id = first + t;                       // 0..9 -> (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
        const int ti = id < 4 ? 0 : id < 7 ? 1 : id < 9 ? 2 : 3;
        const int tj = id < 4 ? id : id < 7 ? id - 3 : id < 9 ? id - 5 : 3;
        const double *pa = Qs + (16 * ti + ij) * CQ_QST + kq, *pb = Qs + (16 * tj + ij) * CQ_QST + kq;
        s64_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < CQ_RS; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = pa[k0 + 4 * u]; b[u] = pb[k0 + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *dst = Gp + (16 * ti + kq + 4 * r) * 64 + 16 * tj + ij;
            if (AGENT) __hip_atomic_store(dst, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = acc[r];
        }
    }
}

// GROUP-LEVEL SUMS OF THE GRAM PARTIALS INSIDE THE PRODUCING LAUNCH (round 6).  The reduce launch between a Gram pass and its
// consumer (k_cqr_reduce: 9.5 us alone, ~19 us beside the V'[A2 | b] product, on every panel's chain twice) goes away for
// panels of up to CQ_GS * CQ_HIER_MAX_GROUPS slabs: the slabs form groups of CQ_GS; a producing workgroup that has stored its
// partial (agent scope, drained) takes a ticket of its group, and the group's LAST arriver adds the group's partials in slab
// order into Gq[group] -- whoever it is, the association is fixed.  The consumers (every pass-1 workgroup; k_cqr_top) add
// the <= 32 group sums in group order themselves (cq_factor), one batch of loads in front of work they do anyway.
// Counters return to zero by themselves.  Only the upper tiles exist (as in Gp; Gq's lower tiles stay as allocated: zero).
constexpr int CQ_GS = 16;                  // slabs per group
constexpr int CQ_HIER_MAX_GROUPS = 32;     // beyond (more than 32768 rows) the reduce kernel stays
__device__ __forceinline__ void cq_group_reduce(const double *Gp, double *Gq, unsigned *cnt, int slab, int nslab, int tid) {
    __shared__ int s_glast;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = slab / CQ_GS, first = g * CQ_GS, gsize = min(CQ_GS, nslab - first);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(gsize - 1);
        if (last) __hip_atomic_store(cnt + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_glast = last;
    }
    __syncthreads();
    if (!s_glast) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // thread (r = tid >> 4, c = tid & 15) of upper tile id: three tiles' worth of loads (48) in flight per round
    const int r = tid >> 4, c = tid & 15;
    for (int id0 = 0; id0 < 10; id0 += 3) {
        double v[3][CQ_GS];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int id = min(id0 + u, 9);
            const int ti = id < 4 ? 0 : id < 7 ? 1 : id < 9 ? 2 : 3;
            const int tj = id < 4 ? id : id < 7 ? id - 3 : id < 9 ? id - 5 : 3;
            const int e = (16 * ti + r) * 64 + 16 * tj + c;
#pragma unroll
            for (int s = 0; s < CQ_GS; ++s)
                v[u][s] = __hip_atomic_load(Gp + (size_t)(first + min(s, gsize - 1)) * 4096 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int id = id0 + u;
            if (id > 9) break;
            const int ti = id < 4 ? 0 : id < 7 ? 1 : id < 9 ? 2 : 3;
            const int tj = id < 4 ? id : id < 7 ? id - 3 : id < 9 ? id - 5 : 3;
            const int e = (16 * ti + r) * 64 + 16 * tj + c;
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < CQ_GS; ++s) acc += s < gsize ? v[u][s] : 0.0;
            Gq[(size_t)g * 4096 + e] = acc;
        }
    }
}
// entry e (row e >> 6, column e & 63) of the reduced Gram matrix from `ngroups` group sums (ngroups == 0: G is the matrix)
__device__ __forceinline__ void cq_gram_entries(const double *__restrict__ G, int ngroups, double (&g)[16], int tid) {
    if (ngroups <= 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) g[q] = G[tid + 256 * q];
        return;
    }
    // (entry e = tid + 256 q: row 4 q + (tid >> 6), column tid & 63; a strictly lower TILE was never formed: zero)
    // four groups x all 16 entries per round of loads (64 in flight); every entry adds its groups in index order
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = 0.0;
    for (int gr = 0; gr < ngroups; gr += 4) {
        double v[16][4];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q;
            const bool up = ((e >> 6) >> 4) <= ((e & 63) >> 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gi = min(gr + k, ngroups - 1);
                v[q][k] = up ? G[(size_t)gi * 4096 + e] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) g[q] += gr + k < ngroups ? v[q][k] : 0.0;
    }
}


struct CqrWork {
    double *Gp = nullptr;    // [max_slabs][64 x 64] Gram partials of the 128-row slabs
    double *G = nullptr, *G2 = nullptr;   // reduced Gram matrices of the raw panel / of Q1 (row-major, upper triangle)
    double *R1 = nullptr;    // R1 (row-major 64 x 64)
    double *Binv = nullptr;  // inv(Q_top - S)
    double *Minv = nullptr;  // look-ahead panel: inv(R1) | inv(R2) from k_cqr_factor (2 x 64 x 64, row-major)
    double *Gq1 = nullptr, *Gq2 = nullptr;   // group sums of the Gram partials of the raw panel / of Q1 ([groups][64 x 64]; cq_group_reduce)
    unsigned *gcnt = nullptr;                //   ... and the groups' ticket counters (zero between launches)
    double *R2inv = nullptr; // Q1 form (round 6): inv(R2) from k_cqr_top, applied to the 64 x N matrices by k_cqr_tw_q1
    bool q1form = false;     //   ... the form the panel in flight was launched in (lsq_cqr_panel sets it, lsq_cqr_tw reads it)
    double *S = nullptr;     // 64 signs
    double *SR = nullptr;    // S R2 R1, the panel's part of the factor ([col][row]); k_cqr_tw moves it into A
    int max_slabs = 0;
    hipStream_t side = nullptr;           // the LU of Q_top runs here, beside the V'[A2 | b] product
    hipStream_t ahead = nullptr;          // look-ahead: panel k + 1's passes run here beside the update of panel k's other columns
    hipEvent_t ev_q = nullptr, ev_lu = nullptr;
    hipEvent_t ev_first = nullptr, ev_panel = nullptr;   // look-ahead: next panel's columns updated / its passes done
    bool ready = false;
};
int lsq_cqr_alloc(lsq_ctx *c, CqrWork *w, int M);
void lsq_cqr_free(CqrWork *w);
// Panel c0..c0+63 of A (column-major, leading dimension M, rows c0..M-1).  In stream order afterwards: Vb (ldv = M - c0)
// holds Q1 (Q1 form, round 6: the panel has had ONE orthogonalisation pass; inv(R2) is applied to the small matrices by
// lsq_cqr_tw) or Q (LSQ_QR_CQR_PASS2=1: both passes), and the panel's part of R (w->SR) and the kernel of the block reflector
// are on their way on the side stream; lsq_cqr_tw puts the former into A's 64 x 64 triangle.  A breakdown (cond(panel) beyond
// ~1e7) sets bit 1 of *d_err.
// ps: the stream the passes run on (the context's, or w->ahead for a look-ahead panel).
// gram_ready: w->Gp already holds the Gram partials of this panel's 64-row slabs (left by the previous panel's update).
// hier: the Gram partials are summed per group inside the producing launches (cq_group_reduce) -- then, with gram_ready, the
// producer of this panel's partials must have left w->Gq1 as well.
int lsq_cqr_panel(lsq_ctx *c, CqrWork *w, double *A, int M, int c0, double *Vb, int ldv, int *d_err, hipStream_t ps,
                  bool gram_ready = false, bool hier = false, double *Vs = nullptr /* Q1 form: Q1 also in fragment order */);
// ... and Q1 (the panel's V) for k_qr1_vtb_w: per 16-row chunk c of the panel, 16-column tile it and half h, 64 lanes x 16 bytes,
//   lane = ij + 16 kq  holds  Q1[row 16 c + 4 kq + 2 h + (0, 1)][column 16 it + ij]      (full chunks only)
__host__ __device__ inline size_t lsq_cqr_vs_index(int chunk, int it, int h, int lane) {
    return ((size_t)((chunk * 4 + it) * 2 + h) * 64 + (size_t)lane) * 2;
}
bool lsq_cqr_q1form();
// the group-level Gram sums apply to a panel of nslab slabs (Q1 form, <= CQ_GS * CQ_HIER_MAX_GROUPS slabs, LSQ_QR_HIER=1: an
// experiment that measured slower than the reduce launches -- see lsq_cqr_hier)
bool lsq_cqr_hier(int nslab);
// after W = Vb'[Vb | A2 | b] (k_qr1_vtb + k_qr1_wreduce):  the update kernel's 64 x N operand for the trailing columns and b.
// Q1 form: A2 -= Vb W2 over ALL rows finishes the block step (the [S W2; 0] part has been added to A2's top rows here);
// three-pass form: Vb becomes V = Q - [S; 0] here.
int lsq_cqr_tw(lsq_ctx *c, CqrWork *w, const double *W, int ncolsB, double *A, int M, int c0, int cend, int n,
               double *rhs, double *Vb, int ldv, double *W2, double *W2s = nullptr);
// FRAGMENT ORDER of W2 (round 6, late).  A load instruction costs the MFMA stream of its SIMD about two clocks per (4-lane group,
// cache line) pair it touches (profiles/r06/ab_c3_vtb_overlap.txt, section 4): a wave in which every lane reads 32 bytes of its OWN
// column -- the natural way to fetch an MFMA operand from a column-major matrix -- touches 64 of them per instruction, a wave
// that reads 1 KB contiguous 16.  So the producer of W2 (k_cqr_tw_q1, Q1 form) also stores it in the order in which the trailing
// update's lanes consume it: per 16-column tile T and k group g (16 k values) and half h, 64 lanes x 16 bytes contiguous,
//   lane = ij + 16 kq  holds  W2[column 16 T + ij][k = 16 g + 4 kq + 2 h + (0, 1)].
__host__ __device__ inline size_t lsq_cqr_w2s_index(int T, int ij, int k) {
    const int g = k >> 4, kq = (k >> 2) & 3, h = (k >> 1) & 1, u = k & 1;
    return ((size_t)((T * 4 + g) * 2 + h) * 64 + (size_t)(ij + 16 * kq)) * 2 + u;
}
