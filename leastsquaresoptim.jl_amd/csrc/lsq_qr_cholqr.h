// CholeskyQR2 panel + basis-kernel block reflector for the blocked QR (lsq_qr_cholqr.hip); used by lsq_dense.hip
#pragma once
#include "lsq_common.h"

struct CqrWork {
    double *Gp = nullptr;    // [max_slabs][64 x 64] Gram partials of the 128-row slabs
    double *G = nullptr, *G2 = nullptr;   // reduced Gram matrices of the raw panel / of Q1 (row-major, upper triangle)
    double *R1 = nullptr;    // R1 (row-major 64 x 64)
    double *Binv = nullptr;  // inv(Q_top - S)
    double *Minv = nullptr;  // look-ahead panel: inv(R1) | inv(R2) from k_cqr_factor (2 x 64 x 64, row-major)
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
// holds Q, and the panel's part of R (w->SR) and the kernel of the block reflector are on their way on the side stream;
// lsq_cqr_tw puts the former into A's 64 x 64 triangle.  A breakdown (cond(panel) beyond ~1e7) sets bit 1 of *d_err.
// ps: the stream the passes run on (the context's, or w->ahead for a look-ahead panel).
int lsq_cqr_panel(lsq_ctx *c, CqrWork *w, double *A, int M, int c0, double *Vb, int ldv, int *d_err, hipStream_t ps);
// after W = Vb'[Vb | A2 | b] (k_qr1_vtb + k_qr1_wreduce):  W2 = T'W for the trailing columns and b; turns Vb into V = Q - [S; 0]
int lsq_cqr_tw(lsq_ctx *c, CqrWork *w, const double *W, int ncolsB, double *A, int M, int c0, int cend, int n,
               const double *rhs, double *Vb, int ldv, double *W2);
