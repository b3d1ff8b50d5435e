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
    int notdone;      // !done: the skip flag of launches that were queued to run AFTER the solve (the LM loop's step /
                      // predicted residual / trial residual chain, enqueued behind the predicted last iteration)
};

// A fast path that relies on what the hardware does but HIP does not promise (all workgroups of a launch co-resident and
// dispatched in index order: in-kernel tile / slab exchanges) has bounded waits; a wait that gives up is COUNTED and the path
// is switched off for a while -- not for ever: after `cooldown` further solves it is armed again (a neighbour on the device,
// e.g. an RCCL kernel of a sharded run, may be gone by then); if it gives up again right away the next pause is four times
// as long (16, 64, ... 4096 solves), a clean armed solve resets the pause to 16.
struct LsqFallback {
    int giveups = 0;        // how often a bounded wait of this path gave up (lsq_solver_stats)
    int cooldown = 0;       // solves left before the path is armed again (0: armed)
    int next_pause = 16;
    bool exchange = true;   // false: the path has no in-kernel exchange (fb_cholqr: a numerical fallback) -- LSQ_DEBUG_SERIAL=2 leaves it on
    bool off() const { return cooldown > 0 || (exchange && lsq_dbg_serial >= 2); }
    void gave_up(lsq_ctx *c, int which) {
        giveups++;
        c->fallback_giveups[which]++;
        cooldown = next_pause;
        next_pause = next_pause < 4096 ? next_pause * 4 : 4096;
    }
    void solve_done(bool used_and_clean) {      // once per solve of the owning solver
        if (cooldown > 0) --cooldown;
        else if (used_and_clean) next_pause = 16;
    }
};
enum { LSQ_FB_CHOL_TILES = 0, LSQ_FB_TRI_PIPE = 1, LSQ_FB_QR_EXCHANGE = 2, LSQ_FB_CHOLQR = 3 };

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
    double *d_f3 = nullptr;    // three-launch iteration (lsq_lsmr3.h): x2 | hbar2 | h2 (n each) | pu[2][4096] | pn[2][3][UB_MAX] | int counts
    size_t f3_elems = 0;
    unsigned f3_tag = 0;       // tag of the newest in-launch record of k_lsmr_fused
    unsigned epoch = 0;
    int last_iter = 0, last_istop = 0;
    // row-sharded single problem (lsq_options.row_allreduce): J is this rank's row block; J'u and sum(u^2) are summed over
    // the ranks in d_xbuf (n + 1 doubles) between the adjoint product and its epilogue
    int (*row_cb)(double *, int, void *, void *) = nullptr;
    void *row_user = nullptr;
    long long global_rows = 0;
    double *d_xbuf = nullptr;  // n + 2
    // colsumabs2 of the WHOLE Jacobian (sum over the ranks of the row blocks' column sums) for the default Jacobi
    // preconditioner and LM's damping: solver-owned, keyed on the handle and its version -- the handle's own cache keeps
    // the LOCAL block's sums, which is what lsq_colsumabs2(J) and any unsharded solve on the same handle must see
    double lsmr_ratio2 = 0.1;     // test2(k) / test2(k-1) as last observed: the starting value of the next solve's stop prediction
    double *d_colsum_g = nullptr;
    unsigned long long colsum_g_uid = 0;     // lsq_mat::uid of the handle the buffer belongs to (0: none)
    unsigned long long colsum_g_version = ~0ull;
    int *d_one = nullptr;      // the constant 1 (partial count of a sum that is already complete)
    // --- dense Cholesky (dense_cholesky.jl:7-21) ---
    double *d_chol = nullptr;  // n*n
    double *d_Ds = nullptr;    // ceil(n/64) factored 64 x 64 diagonal blocks in flight (blocked Cholesky)
    double *d_rhs = nullptr;   // n
    int *d_info = nullptr;
    // --- dense QR (dense_qr.jl:6-28, 50-54) ---
    int last_qr_path = 0;      // lsq_solver_qr_path
    int last_qr_panel = 0;     // lsq_solver_qr_panel
    double *d_qr = nullptr;    // (m [+n]) * n
    double *d_qu = nullptr;    // max(m,n) or m+n
    double *d_tau = nullptr;
    double *d_T = nullptr;     // block reflector factors
    double *d_work = nullptr;
    size_t work_elems = 0;
    std::vector<double> h_R;   // n*n host copy for pivoting / rank decisions
    int last_rank = -1;
    int (*precond_cb)(double *, lsq_mat *, const double *, void *) = nullptr;   // LSMR(preconditioner!, P), diagonal P
    void *precond_user = nullptr;
    // LSMR(preconditioner!, P) with ANY P supporting ldiv! (lsq_lsmr_general.hip): update = preconditioner!(P, x, J, damp),
    // ldiv = ldiv!(out, P, in).  Set => the operator-level LSMR runs instead of the fused one.
    int (*gen_update)(lsq_mat *, const double *, void *) = nullptr;
    int (*gen_ldiv)(double *, const double *, void *) = nullptr;
    void *gen_user = nullptr;
    void *qr2 = nullptr;            // two-stage QR workspace (lsq_dense.hip), allocated on first use
    void (*qr2_free)(void *) = nullptr;
    void *tripipe = nullptr;        // pipelined triangular solves of the blocked Cholesky (lsq_dense.hip)
    void (*tripipe_free)(void *) = nullptr;
    double *tri_X = nullptr, *tri_T = nullptr, *tri_fro = nullptr, *tri_hfro = nullptr;   // explicit inverse of the Cholesky factor (Dogleg certificate)
    int last_chol_path = 0;         // lsq_solver_chol_path
    LsqFallback fb_pipe;            // pipelined triangular solves (else single-workgroup solves)
    bool chol_have_diaginv = false; // the last blocked factorisation left inv(U_kk) in the solve pipeline's buffer
    unsigned *d_chol_flags = nullptr; // k_chol_tiles: epoch-tagged 'tile published' flags
    unsigned chol_epoch = 0;
    LsqFallback fb_tiles;           // one-launch factorisation k_chol_tiles (else launch-per-panel)
    LsqFallback fb_qrx;             // QR: slab exchange of the panel steps + pipelined certified solve (mirrors Qr2Work::no_exchange)
    LsqFallback fb_cholqr{0, 0, 16, false};   // QR: CholeskyQR2 panels (numerical breakdowns; mirrors Qr2Work::no_cholqr)
    bool last_chol_tiles = false;   // the last blocked factorisation was the one-launch one
    bool pub_want = false;          // lsq_tri_chol_solve: let the backward solve's last block publish {info, pipeline flag}
    unsigned long long pub_seq = 0; // ... sequence number of that hand-over (0: it was not launched, use lsq_read_ints)
};
int lsq_tri_chol_solve(lsq_solver *s, const double *U, int n, double *d_bx);
int lsq_tri_chol_fwd_operands(lsq_solver *s, int n, double **z, unsigned long long **slot, unsigned long long *epoch, int **err);
// buffer of the inverted 64 x 64 diagonal blocks of the pipelined solves (allocates the pipeline; nullptr when it is off)
double *lsq_tri_chol_diagbuf(lsq_solver *s, int n);

#ifdef __HIPCC__
// the scalar recurrence of one iteration (lsmr.jl:127-196, :205), on a private copy of the state
__device__ inline void lsmr_rotate_inline(LsmrState &s, double alpha, double beta) {
    const double lambda = 0.0;
    // lsmr.jl:127-130
    double alphahat = sqrt(s.alphabar * s.alphabar + lambda * lambda);
    double chat = s.alphabar / alphahat, shat = lambda / alphahat;
    // :132-138
    double rhoold = s.rho;
    double rho = sqrt(alphahat * alphahat + beta * beta);
    double c = alphahat / rho, sn = beta / rho;
    double thetanew = sn * alpha;
    s.alphabar = c * alpha;
    // :140-149
    double rhobarold = s.rhobar, zetaold = s.zeta;
    double thetabar = s.sbar * rho;
    double rhotemp = s.cbar * rho;
    double rhobar = sqrt((s.cbar * rho) * (s.cbar * rho) + thetanew * thetanew);
    s.cbar = s.cbar * rho / rhobar;
    s.sbar = thetanew / rhobar;
    s.zeta = s.cbar * s.zetabar;
    s.zetabar = -s.sbar * s.zetabar;
    s.rho = rho;
    s.rhobar = rhobar;
    // :152-156 coefficients of the vector updates
    s.c1 = -thetabar * rho / (rhoold * rhobarold);
    s.c2 = s.zeta / (rho * rhobar);
    s.c3 = -thetanew / rho;
    // :164-184 estimate of ||r||
    double betaacute = chat * s.betadd, betacheck = -shat * s.betadd;
    double betahat = c * betaacute;
    s.betadd = -sn * betaacute;
    double thetatildeold = s.thetatilde;
    double rhotildeold = sqrt(s.rhodold * s.rhodold + thetabar * thetabar);
    double ctildeold = s.rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
    s.thetatilde = stildeold * rhobar;
    s.rhodold = ctildeold * rhobar;
    s.betad = -stildeold * s.betad + ctildeold * betahat;
    s.tautildeold = (zetaold - thetatildeold * s.tautildeold) / rhotildeold;
    double taud = (s.zeta - s.thetatilde * s.tautildeold) / s.rhodold;
    s.d = s.d + betacheck * betacheck;
    s.normr = sqrt(s.d + (s.betad - taud) * (s.betad - taud) + s.betadd * s.betadd);
    // :187-189 ||A||
    s.normA2 = s.normA2 + beta * beta;
    s.normA = sqrt(s.normA2);
    s.normA2 = s.normA2 + alpha * alpha;
    // :192-196 cond(A)
    s.maxrbar = fmax(s.maxrbar, rhobarold);
    if (s.iter + 1 > 1) s.minrbar = fmin(s.minrbar, rhobarold);
    s.condA = fmax(s.maxrbar, rhotemp) / fmin(s.minrbar, rhotemp);
    s.normAr = fabs(s.zetabar);                      // :205
}

#endif

// implemented in lsq_lsmr.hip
int lsq_lsmr_alloc(lsq_solver *s);
void lsq_lsmr_free(lsq_solver *s);
// d_Jty (optional): J'*y already formed by the caller (the LM gradient) -- skips the setup product;
// y_sumsq (optional, < 0 = unknown): sum(y.^2) when the caller already holds it (the LM loop's ssr)
// lm (optional, with d_Jty and y_sumsq): the LM loop's damping and projected gradient norm (levenberg_marquardt.jl:82-86,
// 102-104) are formed by the solve's own setup launch -- d_damp is then an OUTPUT (dtd/Delta, left as its square root like
// after every damped solve) and *lm->out_grad receives max|g| -- instead of a launch of their own in front of it
struct LsmrLmPrep {
    const double *colsum;     // colsumabs2(J)
    double inv_delta, min_diag, max_diag;
    const double *x, *lo, *hi;
    double *out_grad;
};
constexpr int LSMR_LM_PREP_MAX_N = 16384;   // every workgroup of that launch reduces colsum and g over all n itself
// tail (optional): what the caller will launch once the solve is over and which depends on nothing but d_x.  `predict` > 0:
// the tail is enqueued right behind inner iteration `predict` with skip = &state->notdone (its kernels return at once if the
// solve is NOT over by then) and no further iteration is queued until that iteration has reported -- if the prediction holds
// (the usual case: the inner count of an LM run changes slowly) the device goes from the last inner iteration straight into
// the caller's next kernels, without the early-exit launches of the look-ahead and without waiting for the host to notice.
// Otherwise, and always when predict == 0, the tail is called (again) with skip = nullptr after the solve.
// `dynamic`: from inner iteration 1 on the guess is replaced by a prediction from the solve's own stopping quantities (the
// hints K3 publishes beside the progress word): on LM's damped, Jacobi-preconditioned operators test2 = |A'r|/(|A||r|) falls
// geometrically and the iteration at which it crosses atol (or test1 crosses btol) is known one or two iterations ahead; a
// tail that skipped itself may be queued again, guarded, behind a later iteration.
struct LsmrTail {
    int predict = 0;
    int (*fn)(const int *skip, void *user) = nullptr;
    void *user = nullptr;
    bool dynamic = false;
};
int lsq_rowshard_colsum(lsq_solver *s, lsq_mat *J, const double **out);   // colsumabs2 of the whole J (row-sharded: summed over the ranks)
int lsq_lsmr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul,
                   const double *d_Jty = nullptr, double y_sumsq = -1.0, const LsmrLmPrep *lm = nullptr,
                   const LsmrTail *tail = nullptr);
// whether lsq_lsmr_solve takes the LsmrLmPrep route for this solver / Jacobian (else the caller launches its own damping)
bool lsq_lsmr_takes_lm_prep(const lsq_solver *s, const lsq_mat *J);
// implemented in lsq_lsmr_general.hip (operator-level LSMR around a caller-supplied ldiv!(., P, .))
int lsq_lsmr_general_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul);
// implemented in lsq_exact.hip (reference-order kernels for small problems)
int lsq_lsmr_exact_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul);
// implemented in lsq_dense.hip
int lsq_dense_solver_alloc(lsq_solver *s);
void lsq_dense_solver_free(lsq_solver *s);
int lsq_cholesky_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul);
int lsq_qr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, const double *d_damp, double *d_x, int *nmul);
