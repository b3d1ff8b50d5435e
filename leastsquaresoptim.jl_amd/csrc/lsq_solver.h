// AbstractAllocatedSolver equivalents (types.jl:138-139): buffers are allocated once per problem
// (README "minimal allocation"), never inside an iteration.
#pragma once
#include "lsq_common.h"

// Device-resident LSMR recurrence (lsmr.jl:82-113 initial values, :127-196 per-iteration update).
struct LsmrState {
    double alpha, beta, rho, rhobar, cbar, sbar, zeta, zetabar, alphabar;
    double betadd, betad, rhodold, tautildeold, thetatilde, d;
    double normA2, maxrbar, minrbar, normb, normr, normAr, normA, condA, normx;
    // coefficients consumed by the vector kernels
    double cu;        // u~ <- A v - cu * u~        (cu = alpha / beta: u is kept unnormalised)
    double inv_beta;  // v~ <- P (J'u~ + d ux~) * inv_beta - beta * v
    double vscale;    // v  <- v~ * vscale          (1/alpha, or 1 when the v update was skipped)
    double c1, c2, c3;
    double atol, btol, ctol;
    int iter, istop, done, maxiter;
    int beta_zero;    // lsmr.jl:120: beta == 0 skips the v update
    int first;        // setup pass (lsmr.jl:73-78)
    unsigned epoch;
};

struct lsq_solver {
    lsq_ctx *ctx;
    int kind;
    int for_lm;
    int m, n;
    // --- LSMR (iterative_lsmr.jl:161-171, 216-231) ---
    LsmrState *d_state = nullptr;
    double *d_u = nullptr;     // m
    double *d_ux = nullptr;    // n   (damped rows of u; the reference's `zerosvector`)
    double *d_v = nullptr, *d_h = nullptr, *d_hbar = nullptr, *d_t = nullptr;  // n
    double *d_P = nullptr;     // n   InverseDiagonal._
    double *d_dg = nullptr;    // n   sqrt(damp)
    double *d_red = nullptr;   // deferred-reduction partials: pu[4096] | pv[4096] | int counts[2]
    unsigned epoch = 0;
    int last_iter = 0, last_istop = 0;
    // --- dense Cholesky (dense_cholesky.jl:7-21) ---
    double *d_chol = nullptr;  // n*n
    double *d_rhs = nullptr;   // n
    int *d_info = nullptr;
    // --- dense QR (dense_qr.jl:6-28, 50-54) ---
    double *d_qr = nullptr;    // (m [+n]) * n
    double *d_qu = nullptr;    // max(m,n) or m+n
    double *d_tau = nullptr;
    double *d_T = nullptr;     // block reflector factors
    double *d_work = nullptr;
    size_t work_elems = 0;
    std::vector<double> h_R;   // n*n host copy for pivoting / rank decisions
    int last_rank = -1;
};

// implemented in lsq_lsmr.hip
int lsq_lsmr_alloc(lsq_solver *s);
void lsq_lsmr_free(lsq_solver *s);
// d_Jty (optional): J'*y already formed by the caller (the LM gradient) -- skips the setup product
int lsq_lsmr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul,
                   const double *d_Jty = nullptr);
// implemented in lsq_dense.hip
int lsq_dense_solver_alloc(lsq_solver *s);
void lsq_dense_solver_free(lsq_solver *s);
int lsq_cholesky_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul);
int lsq_qr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul);
