// Workspace of the two-stage QR (allocated on first use, owned by the solver): shared by lsq_qr_stage1.hip (stage 1: the
// unpivoted blocked factorisation) and lsq_qr.hip (certificate, pivoted sweep on R, solve).
#pragma once
#include "lsq_qr_cholqr.h"
#include "lsq_solver.h"

constexpr int Q2_NB = 64;     // panel width
constexpr int Q2_KC = 32;     // k-rows staged per MFMA step
constexpr int Q2_KS = Q2_KC + 2;
constexpr int QR_NT = 1024;   // threads of the one-workgroup / per-column kernels
constexpr int QR1_SPIN_LIMIT = 1 << 22;   // bound of the in-kernel exchange waits (slab exchange, pipelined solves)
typedef double v4d_qr __attribute__((ext_vector_type(4)));

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

struct Qr2Work {
    double *Vb = nullptr, *Vb2 = nullptr /* look-ahead: the next panel's Q */, *Vs = nullptr, *Vs2 = nullptr /* Vb / Vb2 in V'B's fragment order (lsq_cqr_vs_index) */, *Wp = nullptr, *W = nullptr, *W2 = nullptr, *W2s = nullptr /* W2 in the update's fragment order (lsq_cqr_w2s_index) */, *R = nullptr, *rhs2 = nullptr, *tau1 = nullptr;
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
    int kslices = 0, wp_slots = 0, M = 0, n = 0;
    CqrWork cq;               // stage 1: CholeskyQR2 panel (lsq_qr_cholqr.hip)
    bool no_cholqr = false;   //   ... off for this solver after a breakdown (ill-conditioned / rank-deficient panels)
    bool cholqr_used = false; //   the current factorisation took it at least once
};

// stage 1 (lsq_qr_stage1.hip): factors [A | b] (s->d_qr, s->d_qu), returns the n x n triangle and the first n entries of Q'b
bool lsq_qr2_applies(int M, int n);
int lsq_qr2_factor(lsq_solver *s, int M, int n, double **R_out, double **rhs_out);
