// Device-resident LSMR (Fong & Saunders) on the damped, Jacobi-preconditioned operator
//     A = [J; diag(sqrt(damp))] * diag(P)          b = (y, 0)
// replacing lsmr.jl:53-238 driven by iterative_lsmr.jl:179-198 (Dogleg) / :238-259 (LM).
//
// One inner iteration = three launches, zero host synchronisations:
//   K1  u~ <- J t - cu*u~   (+ damped rows, + sum u~^2 -> beta)          t = P.*v, cu = alpha/beta
//   K2  v~ <- P.*(J'u~ + d.*ux~)/beta - beta*v  (+ sum v~^2 -> alpha, + the two Givens chains,
//        ||r||, ||A||, cond(A) estimates: ~60 scalar flops in the last block)
//   K3  v <- v~/alpha; hbar, x, h updates; t <- P.*v; sum x^2 -> ||x||; the 7 stopping rules
// The PreconditionedMatrix / DampenedMatrix / MyAdjoint / InverseDiagonal wrappers
// (iterative_lsmr.jl:12-122) become epilogue arithmetic; u is kept UNNORMALISED (its 1/beta is
// folded into the consumer), which removes the rmul! passes over the m-vector (lsmr.jl:121).
// The recurrence lives in device memory; the host learns about termination through a pinned
// mailbox word and runs at most LSQ_LOOKAHEAD iterations ahead (kernels of a finished solve exit
// on their first instruction).  Stopping rules are evaluated every iteration, so `iter` (and
// mul_calls = 2*iter, lsmr.jl:236) is the count the reference would report.
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "lsq_solver.h"
#include "lsq_spmv.h"

static constexpr int LSQ_LOOKAHEAD_DEFAULT = 2;
static constexpr bool LSQ_TAIL_REACT_DEFAULT = true;       // (lsq_lsmr_solve, three-launch iteration: react to a cautious launch instead of pre-placing the tail)

// mailbox word: [63:41] epoch | [40] done | [39:32] istop | [31:0] iter
__device__ __forceinline__ void publish(LsqMailbox *mail, const LsmrState *st) {
    unsigned long long w = ((unsigned long long)(st->epoch & 0x7fffffu) << 41) |
                           ((unsigned long long)(st->done ? 1 : 0) << 40) |
                           ((unsigned long long)(st->istop & 0xff) << 32) | (unsigned)st->iter;
    __hip_atomic_store((unsigned long long *)mail, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- setup: P, sqrt(damp) (iterative_lsmr.jl:129-141, 251-252) ------------------------------
__global__ void __launch_bounds__(LSQ_NT)
k_lsmr_prep(int n, const double *__restrict__ colsum, double *__restrict__ damp, double *__restrict__ P,
            double *__restrict__ dg, double *__restrict__ ux, int custom_p) {
    for (int j = blockIdx.x * LSQ_NT + threadIdx.x; j < n; j += gridDim.x * LSQ_NT) {
        double s = colsum[j];
        if (damp) {
            double d = damp[j];
            s += d;                  // axpy!(1, damp, out._)  -- damp is still dtd/Delta here
            double r = sqrt(d);      // map!(sqrt, damp, damp)
            dg[j] = r;
            damp[j] = r;             // the reference clobbers the caller's damp (:252)
            ux[j] = 0.0;             // zerosvector (:246)
        }
        if (!custom_p) P[j] = s > 0.0 ? 1.0 / sqrt(s) : 0.0;   // (custom_p: the caller's preconditioner! filled P)
    }
}

// ---- beta_1^2 = sum(b^2) as block partials, and state reset (lsmr.jl:73-75 with x == 0) --------
__global__ void __launch_bounds__(LSQ_NT)
k_lsmr_begin(int m, const double *__restrict__ y, LsmrState *st, double *pu, int *npu, double atol, double btol,
             double ctol, int maxiter, unsigned epoch) {
    __shared__ double sh[LSQ_NT / 64];
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // no other kernel touches the state concurrently
        st->iter = 0;
        st->istop = 0;
        st->done = 0;
        st->notdone = 1;
        st->first = 1;
        st->atol = atol;
        st->btol = btol;
        st->ctol = ctol;
        st->maxiter = maxiter;
        st->epoch = epoch;
        st->cu = 0.0;
        *npu = (int)gridDim.x;
    }
    double acc = 0.0;
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < m; i += (long long)gridDim.x * LSQ_NT) {
        double v = y[i];
        acc += v * v;
    }
    double bv = block_sum<LSQ_NT>(acc, sh);
    if (threadIdx.x == 0) pu[blockIdx.x] = bv;
}

// ---- K1: u~ <- J t - cu u~ ; publishes the block partials of sum(u~^2) --------------------------
struct EpiU {
    static constexpr bool REDUCE = true;
    using defer = void;
    const int *done;
    int extra_blocks;
    LsmrState *st;
    const double *uold;
    double *unew;
    // damped rows
    int n;
    const double *dg;
    const double *t;
    double *ux;
    double *partials;   // pu
    int *npartials;
    unsigned *counter;  // unused (deferred)
    double cu;          // cached st->cu
    using has_prepare = void;
    __device__ void prepare() { cu = st->cu; }
    __device__ void seg(int s, double dot, double &racc) const {
        double un = dot - st->cu * uold[s];
        unew[s] = un;
        racc += un * un;
    }
    using has_pre = void;
    __device__ double pre(int s) const { return uold[s]; }
    __device__ void seg_pre(int s, double dot, double uo, double &racc) const {
        double un = dot - cu * uo;
        unew[s] = un;
        racc += un * un;
    }
    __device__ void extra(int, double &) const {}   // (the damped rows are updated by K3)
    __device__ void finalize(double) const {}
};

// ---- K2: v~ <- P (J'u~ + d ux~)/beta - beta v ; publishes the block partials of sum(v~^2) -------
// norm(::DampenedVector) exactly as the reference writes it: sqrt(norm(y)^2 + norm(x)^2), iterative_lsmr.jl:72 -- the
// square of a square root is not the sum it came from, so this is NOT sqrt(sum y^2 + sum x^2) in the last bit.
__device__ __forceinline__ double dampened_norm(double sumsq_y, double sumsq_x) {
    const double ny = sqrt(sumsq_y), nx = sqrt(sumsq_x);
    return sqrt(ny * ny + nx * nx);
}

// beta = norm(u~) is formed by every block from K1's partials (lsmr.jl:119, il:72).
struct EpiV {
    static constexpr bool REDUCE = true;
    using defer = void;
    const int *done;
    int extra_blocks;
    LsmrState *st;
    const double *pu;   // K1's (or k_lsmr_begin's) partials of sum(u~_y^2)
    const int *npu;
    const double *px;   // K3's partials of sum(u~_x^2) (damped rows), or null
    const int *npx;
    const double *P;
    const double *dg;
    const double *ux;   // null during setup (zerosvector)
    double *v;
    double *partials;   // pv
    int *npartials;
    unsigned *counter;  // unused (deferred)
    double beta, inv_beta;
    int beta_zero, first;
    double *vout = nullptr;   // where v~ goes (null: in place, over v).  The three-launch iteration (lsq_lsmr3.h) keeps the
                              // normalised v and the new v~ apart: its next launch reads v~ while it rewrites v
    const double *cs = nullptr;   // ... and takes the gather vector of its next J*v straight from here: wout = (v~ .* P) .* s
    double *wout = nullptr;       // (s: the column scale of J = V diag(s), or null)
    // k_combine requests the column's own operands with its first round of partials (has_col_prefetch): seg() finds them in
    // registers when it is called for THAT column (the workgroup's first and, on the LSMR path, only column block)
    using has_col_prefetch = void;
    int pf_j = -1;
    double pf_P = 1.0, pf_dg = 0.0, pf_ux = 0.0, pf_v = 0.0, pf_cs = 1.0;
    __device__ void col_prefetch(int j) {
        pf_j = j;
        pf_P = P ? P[j] : 1.0;
        pf_dg = dg ? dg[j] : 0.0;
        pf_ux = (dg && ux) ? ux[j] : 0.0;
        pf_v = v[j];
        pf_cs = (wout && cs) ? cs[j] : 1.0;
    }
    using has_block_prepare = void;
    __device__ void block_prepare() {
        double b2, bx, unused;
        ordered_sum256x3(pu, npu, px, npx, nullptr, nullptr, b2, bx, unused);
        beta = dg ? dampened_norm(b2, px ? bx : 0.0) : sqrt(b2);   // (px null in the setup pass: u~x == 0)
        beta_zero = !(beta > 0.0);
        inv_beta = beta > 0.0 ? 1.0 / beta : 1.0;   // lsmr.jl:120-121 rmul!(u, inv(beta))
        first = st->first;
    }
    __device__ void seg(int j, double dot, double &racc) const {
        if (beta_zero) {                            // lsmr.jl:120: v (and alpha) stay as they are
            // The three-launch iteration reads v~ and the gather vector from buffers of their own and rescales them by 1/alpha:
            // with beta == 0 they must hold the NORMALISED v (the consumer's scale is 1 then, lsmr_scalars), not the previous
            // iteration's unnormalised v~.  Setup pass (first): v is still zero (lsmr.jl:76 never ran).
            if (vout) {
                const double vj = first ? 0.0 : v[j];
                vout[j] = vj;
                if (wout) {
                    const double t = P ? vj * P[j] : vj;
                    wout[j] = cs ? t * cs[j] : t;
                }
                racc += vj * vj;
            }
            return;
        }
        const bool pf = j == pf_j;                  // (the same operands, fetched a round trip earlier)
        double w = dot;
        if (dg && ux) w += (pf ? pf_ux : ux[j]) * (pf ? pf_dg : dg[j]);           // iterative_lsmr.jl:107
        w *= inv_beta;                              // u = u~/beta
        const double Pj = P ? (pf ? pf_P : P[j]) : 1.0;
        if (P) w *= Pj;                             // :41
        double vn = first ? w : w - beta * (pf ? pf_v : v[j]);    // :42-49 (beta == 0 => fill!)
        (vout ? vout : v)[j] = vn;
        if (wout) {
            const double t = P ? vn * Pj : vn;
            wout[j] = cs ? t * (pf ? pf_cs : cs[j]) : t;
        }
        racc += vn * vn;
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

// alpha, beta and the rotations of one iteration on a private copy of the state (one thread)
__device__ __forceinline__ void lsmr_scalars(LsmrState &ns, double beta2, double betax2, double alpha2, bool damped, bool have_px) {
    const double beta = damped ? dampened_norm(beta2, have_px ? betax2 : 0.0) : sqrt(beta2);   // (no px in the setup pass)
    ns.beta = beta;
    ns.beta_zero = !(beta > 0.0);
    if (!ns.beta_zero) {
        ns.alpha = sqrt(alpha2);                       // lsmr.jl:77,123
        ns.vscale = ns.alpha > 0.0 ? 1.0 / ns.alpha : 1.0;
    } else {
        if (ns.first) ns.alpha = 0.0;
        ns.vscale = 1.0;                               // v was left untouched (lsmr.jl:120)
    }
    const double alpha = ns.alpha;
    if (ns.first) {                                    // lsmr.jl:82-113
        ns.zetabar = alpha * beta;
        ns.alphabar = alpha;
        ns.rho = 1.0; ns.rhobar = 1.0; ns.cbar = 1.0; ns.sbar = 0.0;
        ns.betadd = beta; ns.betad = 0.0; ns.rhodold = 1.0; ns.tautildeold = 0.0;
        ns.thetatilde = 0.0; ns.zeta = 0.0; ns.d = 0.0;
        ns.normA = -1.0; ns.condA = -1.0; ns.normx = -1.0;
        ns.normA2 = alpha * alpha;
        ns.maxrbar = 0.0; ns.minrbar = 1e100;
        ns.normb = beta; ns.normr = beta; ns.normAr = alpha * beta;
    } else {
        lsmr_rotate_inline(ns, alpha, beta);
    }
    ns.cu = beta > 0.0 ? alpha / beta : alpha;         // next K1: u~_new = A v - (alpha/beta) u~
}

// ||x||, the 7 stopping rules and the commit of the new state (one thread of the last block)
__device__ __forceinline__ void lsmr_commit(LsmrState &s, double total, LsmrState *st, LsqMailbox *mail) {
    if (s.first) {
        s.first = 0;
        if (!(s.normAr != 0.0)) { s.done = 1; s.notdone = 0; }      // lsmr.jl:115: exit if b = 0 or A'b = 0
        *st = s;
        publish(mail, st);
        return;
    }
    s.iter += 1;
    s.normx = sqrt(total);                       // lsmr.jl:206
    double test1 = s.normr / s.normb;
    double test2 = s.normAr / (s.normA * s.normr);
    double test3 = 1.0 / s.condA;
    double t1 = test1 / (1.0 + s.normA * s.normx / s.normb);
    double rtol = s.btol + s.atol * s.normA * s.normx / s.normb;
    int istop = 0;                               // :224-231, first hit wins
    if (s.iter >= s.maxiter) istop = 7;
    else if (1.0 + test3 <= 1.0) istop = 6;
    else if (1.0 + test2 <= 1.0) istop = 5;
    else if (1.0 + t1 <= 1.0) istop = 4;
    else if (test3 <= s.ctol) istop = 3;
    else if (test2 <= s.atol) istop = 2;
    else if (test1 <= rtol) istop = 1;
    s.istop = istop;
    if (istop) { s.done = 1; s.notdone = 0; }
    *st = s;
    // (hints for the host's prediction of the stop iteration: in front of the word that announces the iteration)
    __hip_atomic_store((double *)&mail->test1, test1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((double *)&mail->test2, test2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    publish(mail, st);
}

// ---- K3: alpha, rotations (every block, redundantly and identically), n-vector updates, ||x||,
//          stopping rules and the commit of the new state (last block) --------------------------
__global__ void __launch_bounds__(LSQ_NT)
k_lsmr_update(int n, LsmrState *st, LsqMailbox *mail, const double *pu, const int *npu, const double *pv,
              const int *npv, const double *px_in, const int *npx_in, double *px, int *npx,
              const double *__restrict__ dg, double *__restrict__ ux,
              const double *__restrict__ P, double *__restrict__ v, double *__restrict__ h,
              double *__restrict__ hbar, double *__restrict__ x, double *__restrict__ xout,
              double *__restrict__ t, double *partials, unsigned *counter) {
    __shared__ double sh[LSQ_NT / 64];
    __shared__ LsmrState ns;   // this iteration's state, computed from the committed one
    // This kernel is a chain of memory latencies (it moves 0.5 MB): everything that does not depend on the scalars is
    // fetched up front, together -- the thread's vector elements, the committed state (one coalesced load into LDS; it
    // also carries the `done` flag of a finished solve) and, inside ordered_sum256x3, counts and partials -- so that one
    // round trip is paid instead of five.
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * LSQ_NT + tid, jc = j0 < n ? j0 : n - 1;
    const double e_v = v[jc], e_h = h[jc], e_hb = hbar[jc], e_x = x[jc];
    const double e_P = P ? P[jc] : 1.0, e_dg = dg ? dg[jc] : 0.0, e_ux = dg ? ux[jc] : 0.0;
    static_assert(sizeof(LsmrState) % 8 == 0 && sizeof(LsmrState) / 8 <= LSQ_NT, "state copy: one 8-byte word per thread");
    if (tid < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)&ns)[tid] = ((const unsigned long long *)st)[tid];
    double beta2, betax2, alpha2;
    ordered_sum256x3(pu, npu, px_in, npx_in, pv, npv, beta2, betax2, alpha2);   // (its barriers also publish ns)
    if (ns.done) return;
    if (threadIdx.x == 0) lsmr_scalars(ns, beta2, betax2, alpha2, dg != nullptr, px_in != nullptr);
    __syncthreads();
    const bool first = ns.first;
    const double vs = ns.vscale, c1 = ns.c1, c2 = ns.c2, c3 = ns.c3, cu = ns.cu;
    double acc = 0.0, accx = 0.0;
    for (int j = j0; j < n; j += gridDim.x * LSQ_NT) {
        const bool pf = j == j0;                     // first trip: the prefetched elements
        const double Pj = P ? (pf ? e_P : P[j]) : 1.0;
        double vj = (pf ? e_v : v[j]) * vs;          // lsmr.jl:78,124 rmul!(v, inv(alpha))
        v[j] = vj;
        if (first) {                                 // :89-90, iterative_lsmr.jl:183,242
            h[j] = vj;
            hbar[j] = 0.0;
            x[j] = 0.0;
            xout[j] = 0.0;
        } else {
            const double hj = pf ? e_h : h[j];
            double hb = (pf ? e_hb : hbar[j]) * c1 + hj;   // :152-153
            hbar[j] = hb;
            double xj = (pf ? e_x : x[j]) + c2 * hb;       // :154
            x[j] = xj;
            h[j] = hj * c3 + vj;                     // :155-156
            acc += xj * xj;
            // the caller's x always holds P.*x of the newest iterate (iterative_lsmr.jl:195-196,
            // 256-257), so no launch is needed once the stopping rule fires
            xout[j] = P ? xj * Pj : xj;
        }
        const double tj = P ? vj * Pj : vj;          // iterative_lsmr.jl:31 ldiv!(tmp, P, a)
        t[j] = tj;
        if (dg) {   // damped rows of the NEXT u (iterative_lsmr.jl:92): u~x <- d.*t - cu*u~x
            double un = tj * (pf ? e_dg : dg[j]) - cu * (pf ? e_ux : ux[j]);
            ux[j] = un;
            accx += un * un;
        }
    }
    if (dg) {       // deferred partials of sum(u~x^2): read by the next K2 / K3
        double bx = block_sum<LSQ_NT>(accx, sh);
        if (threadIdx.x == 0) {
            px[blockIdx.x] = bx;
            if (blockIdx.x == 0) *npx = (int)gridDim.x;
        }
    }
    double bv = block_sum<LSQ_NT>(acc, sh);
    // last block: every other block is past its prologue, so the state can be committed
    grid_reduce<LSQ_NT>(bv, partials, counter, gridDim.x, sh, [&](double total) { lsmr_commit(ns, total, st, mail); });
}

// (Measured and dropped in round 2: K2's two passes + K3 as ONE launch for the sliced columns, one resident workgroup per CU
//  with two grid barriers -- hierarchical tickets + a polled flag, ~2.2 us each -- and v~ kept in registers.  42.7 us against
//  25.3 + 5.9 + 9.0 us in three launches: the barriers cost what the two launch ramps cost; 2885 vs 2874-2895 LM it/s on C4.)

#include "lsq_lsmr3.h"

// ---- setup from the caller's J'y in one launch: P, sqrt(damp), state reset and
// v~ = P.*(J'y)/beta_1 with the block partials of sum(v~^2)  (k_lsmr_prep + k_lsmr_begin +
// k_combine<EpiV> of the general path; k_lsmr_update in "first" mode follows).
// beta_1^2 = sum(y^2) comes from the caller when it already holds it (the LM loop's ssr), else
// from k_lsmr_begin's partials.
__global__ void __launch_bounds__(LSQ_NT)
k_lsmr_setup(int n, const double *__restrict__ colsum, double *__restrict__ damp, double *__restrict__ P,
             double *__restrict__ dg, double *__restrict__ ux, const double *__restrict__ Jty,
             double *__restrict__ v, LsmrState *st, double *pu, int *npu, double ysumsq, double *pv, int *npv,
             double atol, double btol, double ctol, int maxiter, unsigned epoch, int custom_p,
             const double *__restrict__ cs = nullptr, double *__restrict__ wout = nullptr) {
    __shared__ double sh[LSQ_NT / 64];
    const double beta2 = ysumsq >= 0.0 ? ysumsq : ordered_sum256(pu, *npu);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->iter = 0; st->istop = 0; st->done = 0; st->notdone = 1; st->first = 1;
        st->atol = atol; st->btol = btol; st->ctol = ctol;
        st->maxiter = maxiter; st->epoch = epoch; st->cu = 0.0;
        if (ysumsq >= 0.0) {
            pu[0] = ysumsq;
            *npu = 1;
        }
    }
    const double beta = damp ? dampened_norm(beta2, 0.0) : sqrt(beta2);   // u_x = 0 (zerosvector, il:246)
    const bool beta_zero = !(beta > 0.0);
    const double inv_beta = beta > 0.0 ? 1.0 / beta : 1.0;
    double acc = 0.0;
    for (int j = blockIdx.x * LSQ_NT + threadIdx.x; j < n; j += gridDim.x * LSQ_NT) {
        double s = colsum[j];
        if (damp) {
            double d = damp[j];
            s += d;                  // iterative_lsmr.jl:251
            double r = sqrt(d);      // :252
            dg[j] = r;
            damp[j] = r;
            ux[j] = 0.0;             // zerosvector (:246)
        }
        const double Pj = custom_p ? P[j] : (s > 0.0 ? 1.0 / sqrt(s) : 0.0);
        P[j] = Pj;
        if (!beta_zero) {            // lsmr.jl:76 (beta == 0: v is left untouched, :120)
            const double w = Jty[j] * inv_beta * Pj;
            v[j] = w;
            if (wout) wout[j] = cs ? (w * Pj) * cs[j] : w * Pj;     // the gather vector of the first J*v (lsq_lsmr3.h)
            acc += w * w;
        }
    }
    double bv = block_sum<LSQ_NT>(acc, sh);
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = bv;
        if (blockIdx.x == 0) *npv = (int)gridDim.x;
    }
}

// ---- LM + LSMR, n <= LSMR_LM_PREP_MAX_N: the LM loop's damping + projected gradient norm (k_lm_damp_grad,
// levenberg_marquardt.jl:82-86, 102-104) and k_lsmr_setup as ONE launch of 1024-thread workgroups.  The damping needs
// mean(colsum) -- a reduction over all n -- before the first element can be written: every workgroup takes it itself, over
// all n, in exactly the order of the one-workgroup kernel it replaces (thread-sequential with stride 1024, wave tree, 16
// waves in order: same bits), then handles its own 1024 elements.  One n-length launch (~6 us of latency) less per outer
// iteration; sum(v~^2) is taken per 1024 elements instead of per 256.
__global__ void __launch_bounds__(1024)
k_lm_lsmr_setup(int n, LsmrLmPrep lm, double *__restrict__ damp, double *__restrict__ P, double *__restrict__ dg,
                double *__restrict__ ux, const double *__restrict__ g, double *__restrict__ v, LsmrState *st, double *pu, int *npu,
                double ysumsq, double *pv, int *npv, double atol, double btol, double ctol, int maxiter, unsigned epoch,
                const double *__restrict__ colscale = nullptr, double *__restrict__ wout = nullptr) {
    constexpr int R = LSMR_LM_PREP_MAX_N / 1024;
    __shared__ double sh[16], shm[16];
    __shared__ double s_mean;
    const int tid = threadIdx.x;
    double cs[R], gv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {   // every load is issued before the first use
        const int i = tid + k * 1024;
        cs[k] = i < n ? lm.colsum[i] : 0.0;
        gv[k] = i < n ? g[i] : 0.0;
    }
    double acc = 0.0, mg = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = tid + k * 1024;
        if (i < n) {
            acc += cs[k];
            double gi = gv[k];
            if (lm.lo && lm.x[i] <= lm.lo[i] && gi > 0.0) gi = 0.0;
            else if (lm.hi && lm.x[i] >= lm.hi[i] && gi < 0.0) gi = 0.0;
            double a = fabs(gi);
            if (isnan(a)) a = INFINITY;
            mg = fmax(mg, a);
        }
    }
    acc = wave_sum(acc);
    mg = wave_max(mg);
    if ((tid & 63) == 0) {
        sh[tid >> 6] = acc;
        shm[tid >> 6] = mg;
    }
    __syncthreads();
    if (tid == 0) {
        double tt = 0.0, tm = shm[0];
        for (int w = 0; w < 16; ++w) tt += sh[w];
        for (int w = 1; w < 16; ++w) tm = fmax(tm, shm[w]);
        s_mean = tt / n;
        if (blockIdx.x == 0) {
            *lm.out_grad = tm;
            st->iter = 0; st->istop = 0; st->done = 0; st->notdone = 1; st->first = 1;
            st->atol = atol; st->btol = btol; st->ctol = ctol;
            st->maxiter = maxiter; st->epoch = epoch; st->cu = 0.0;
            pu[0] = ysumsq;
            *npu = 1;
        }
    }
    __syncthreads();
    const double lo_d = lm.min_diag * s_mean, hi_d = lm.max_diag * s_mean;
    const double beta = dampened_norm(ysumsq, 0.0);   // u_x = 0 (zerosvector, il:246)
    const bool beta_zero = !(beta > 0.0);
    const double inv_beta = beta > 0.0 ? 1.0 / beta : 1.0;
    double a2 = 0.0;
    const int j = blockIdx.x * 1024 + tid;             // (this workgroup's own elements: blockIdx.x-th group of 1024)
    if (j < n) {
        // the values of element j sit in this thread's registers when blockIdx.x < R: cs / gv are indexed by a constant after
        // unrolling, so pick by selection
        double c0 = 0.0, g0 = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (k == (int)blockIdx.x) { c0 = cs[k]; g0 = gv[k]; }
        double c = c0;
        c = c > hi_d ? hi_d : (c < lo_d ? lo_d : c);
        const double d = c * lm.inv_delta;     // rmul!(dtd, 1/Delta)
        const double s = c0 + d;               // iterative_lsmr.jl:251
        const double r = sqrt(d);              // :252
        dg[j] = r;
        damp[j] = r;
        ux[j] = 0.0;                           // zerosvector (:246)
        const double Pj = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
        P[j] = Pj;
        if (!beta_zero) {                      // lsmr.jl:76 (beta == 0: v is left untouched, :120)
            const double w = g0 * inv_beta * Pj;
            v[j] = w;
            if (wout) wout[j] = colscale ? (w * Pj) * colscale[j] : w * Pj;     // the gather vector of the first J*v (lsq_lsmr3.h)
            a2 = w * w;
        }
    }
    __syncthreads();
    const double bv = block_sum<1024>(a2, sh);
    if (tid == 0) {
        pv[blockIdx.x] = bv;
        if (blockIdx.x == 0) *npv = (int)gridDim.x;
    }
}

// ---- row-sharded adjoint product: buf[0..n) = J_p'u_p, buf[n] = sum(u_p.^2) -> all-reduce -> epilogue ----
struct EpiBuf {   // out[j] = dot
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int j, double dot, double &) const { out[j] = dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};
__global__ void __launch_bounds__(LSQ_NT) k_rowshard_sumsq(const double *pu, const int *npu, double *out, const int *done,
                                                            double *xbuf, int n) {
    if (done && *done) {
        // an iteration queued behind the stop (the rest of a look-ahead chunk): its product returned at once and the buffer still
        // holds the LAST all-reduced vector, which the coming collective would sum again (x world per queued iteration, for
        // nobody to read -- but it must not run off to Inf): clear it
        for (int i = threadIdx.x; i <= n; i += LSQ_NT) xbuf[i] = 0.0;
        return;
    }
    double a, b, c2;
    ordered_sum256x3(pu, npu, nullptr, nullptr, nullptr, nullptr, a, b, c2);
    if (threadIdx.x == 0) *out = a;
}
extern "C" int lsq_solver_set_row_allreduce(lsq_solver *s, lsq_device_allreduce_callback cb, void *user, long long global_rows) {
    if (!s || s->kind != LSQ_LSMR) {
        lsq_set_error("row-sharded solves: LSMR() only");
        return LSQ_EARG;
    }
    s->row_cb = cb;
    s->row_user = user;
    s->global_rows = cb ? global_rows : 0;
    s->colsum_g_uid = 0;                // (another hook = another group of ranks: the summed colsumabs2 is theirs, not ours)
    s->colsum_g_version = ~0ull;
    if (cb && !s->d_xbuf) {
        // (n + 1 doubles travel; the slot behind them is read as a partial-sum array of count 1, and ordered_sum256x3
        //  fetches the first 256 entries of such an array before it looks at the count)
        LSQ_HIP(hipMalloc(&s->d_xbuf, (size_t)(s->n + 2 + 256) * sizeof(double)));
        LSQ_ZERO(s->d_xbuf, 0, (size_t)(s->n + 2 + 256) * sizeof(double));
        LSQ_HIP(hipMalloc(&s->d_one, sizeof(int)));
        const int one = 1;
        LSQ_HIP(hipMemcpy(s->d_one, &one, sizeof(int), hipMemcpyHostToDevice));
    }
    return LSQ_OK;
}
// colsumabs2 of the whole Jacobian for a row-sharded solver (lsq_solver.h: d_colsum_g): one collective per Jacobian version,
// issued by every rank at the same point of its call sequence; unsharded solvers get the handle's cache
int lsq_rowshard_colsum(lsq_solver *s, lsq_mat *J, const double **out) {
    const double *local = lsq_cached_colsum(J);
    if (!local) return LSQ_EHIP;
    *out = local;
    if (!s->row_cb) return LSQ_OK;
    lsq_ctx *c = s->ctx;
    if (!s->d_colsum_g) LSQ_HIP(hipMalloc(&s->d_colsum_g, (size_t)(s->n > 0 ? s->n : 1) * sizeof(double)));
    if (s->colsum_g_uid != J->uid || s->colsum_g_version != J->version) {
        LSQ_TRY(lsq_d2d(c, s->d_colsum_g, local, (size_t)J->n * sizeof(double)));   // (a copy kernel: no runtime-side staging hole)
        if (s->row_cb(s->d_colsum_g, J->n, (void *)c->stream, s->row_user) != 0) {
            lsq_set_error("row all-reduce callback reported failure");
            return LSQ_ECALLBACK;
        }
        s->colsum_g_uid = J->uid;
        s->colsum_g_version = J->version;
    }
    *out = s->d_colsum_g;
    return LSQ_OK;
}
// v~ <- epilogue(sum over ranks of J_p'src_p) with beta from the summed sum(src.^2); `ev` is the unsharded epilogue
static int rowshard_adjoint(lsq_solver *s, lsq_mat *J, const double *src, EpiV ev, const double *pu, const int *npu) {
    lsq_ctx *c = s->ctx;
    const int n = J->n;
    EpiBuf eb{ev.done, 0, s->d_xbuf, nullptr, nullptr};
    LSQ_TRY(launch_product(J, 1, src, eb));
    LSQ_LAUNCH(k_rowshard_sumsq, dim3(1), dim3(LSQ_NT), 0, c->stream, pu, npu, s->d_xbuf + n, ev.done, s->d_xbuf, n);
    LSQ_HIP(hipGetLastError());
    if (s->row_cb(s->d_xbuf, n + 1, (void *)c->stream, s->row_user) != 0) {
        lsq_set_error("row all-reduce callback reported failure");
        return LSQ_ECALLBACK;
    }
    ev.pu = s->d_xbuf + n;
    ev.npu = s->d_one;
    const int nb = lsq_div_up(n, LSQ_CMB_COLS);
    LSQ_LAUNCH((k_combine<EpiV>), dim3(std::min(nb, 2048)), dim3(LSQ_NT), 0, c->stream, s->d_xbuf, n, 1, ev, nb);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_lsmr_alloc(lsq_solver *s) {
    size_t nb = (size_t)(s->n > 0 ? s->n : 1) * sizeof(double);
    LSQ_HIP(hipMalloc(&s->d_state, 2 * sizeof(LsmrState)));    // (two: the three-launch iteration double-buffers it)
    LSQ_ZERO(s->d_state, 0, 2 * sizeof(LsmrState));
    // the three-launch iteration (lsq_lsmr3.h): second copies of x, hbar, h; sum(u~^2) partials x 2; the three norms x 2; counts
    s->f3_elems = 4 * (size_t)(s->n > 0 ? s->n : 1) + 2 * 4096 + 8 + 16 + 8;
    LSQ_HIP(hipMalloc(&s->d_f3, s->f3_elems * sizeof(double)));
    LSQ_ZERO(s->d_f3, 0, s->f3_elems * sizeof(double));
    LSQ_HIP(hipMalloc(&s->d_u, (size_t)(s->m > 0 ? s->m : 1) * sizeof(double)));
    LSQ_HIP(hipMalloc(&s->d_ux, nb));
    LSQ_HIP(hipMalloc(&s->d_v, nb));
    LSQ_HIP(hipMalloc(&s->d_h, nb));
    LSQ_HIP(hipMalloc(&s->d_hbar, nb));
    LSQ_HIP(hipMalloc(&s->d_t, nb));
    LSQ_HIP(hipMalloc(&s->d_P, nb));
    LSQ_HIP(hipMalloc(&s->d_dg, nb));
    LSQ_HIP(hipMalloc(&s->d_rhs, nb));  // LSMR iterate (un-preconditioned space)
    LSQ_HIP(hipMalloc(&s->d_red, 4 * 4096 * sizeof(double) + 64));  // pu, pv, px[2] (4096 each), counts
    LSQ_ZERO(s->d_red, 0, 4 * 4096 * sizeof(double) + 64);
    return LSQ_OK;
}

void lsq_lsmr_free(lsq_solver *s) {
    hipFree(s->d_state); hipFree(s->d_u); hipFree(s->d_ux); hipFree(s->d_v); hipFree(s->d_h);
    hipFree(s->d_hbar); hipFree(s->d_t); hipFree(s->d_P); hipFree(s->d_dg); hipFree(s->d_rhs); hipFree(s->d_red);
    hipFree(s->d_xbuf); hipFree(s->d_one); hipFree(s->d_colsum_g); hipFree(s->d_f3);
}

static inline int nvec_grid(const lsq_ctx *c, int n) {
    int g = lsq_div_up(n > 0 ? n : 1, LSQ_NT);
    int cap = c->num_cus * 4;
    return g > cap ? cap : g;
}

// d_damp == nullptr: undamped (Dogleg, atol = btol = 1e-6); else LM (btol = 0.5).
bool lsq_lsmr_takes_lm_prep(const lsq_solver *s, const lsq_mat *J) {
    // (row-sharded: every rank must take the same kernels -- their summation orders differ in the last bits, and ranks whose
    //  replicated scalars differ stop at different inner iterations, i.e. issue different numbers of collectives; so the choice
    //  may depend on n only, never on the size of this rank's row block)
    return s->kind == LSQ_LSMR && J->n <= LSMR_LM_PREP_MAX_N && (s->row_cb || !lsq_small_mat(J)) && !s->precond_cb && !s->gen_ldiv &&
           !getenv("LSQ_LSMR_SEPARATE_SETUP");
}

int lsq_lsmr_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul,
                   const double *d_Jty, double y_sumsq, const LsmrLmPrep *lm, const LsmrTail *tail) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n;
    if (m != s->m || n != s->n) {
        lsq_set_error("lsmr: solver allocated for %dx%d, Jacobian is %dx%d", s->m, s->n, m, n);
        return LSQ_EDIM;
    }
    if (s->gen_ldiv) {     // LSMR(preconditioner!, P) with a general P: the operator-level recurrence (lsq_lsmr_general.hip)
        if (s->row_cb) {
            lsq_set_error("row-sharded solves take diagonal preconditioners only");
            return LSQ_EARG;
        }
        LSQ_TRY(lsq_lsmr_general_solve(s, J, d_y, d_damp, d_x, nmul));
        return tail && tail->fn ? tail->fn(nullptr, tail->user) : LSQ_OK;
    }
    const bool sharded = s->row_cb != nullptr;     // J is a row block: the adjoint product is summed over the ranks
    if (lsq_small_mat(J) && !s->precond_cb && !sharded) {   // reference-order kernel
        LSQ_TRY(lsq_lsmr_exact_solve(s, J, d_y, d_damp, d_x, nmul));
        return tail && tail->fn ? tail->fn(nullptr, tail->user) : LSQ_OK;
    }
    static const int lookahead_env = [] {
        const char *e = getenv("LSQ_LOOKAHEAD");
        int v = e ? atoi(e) : LSQ_LOOKAHEAD_DEFAULT;
        return v < 1 ? 1 : v;
    }();
    const int lookahead = lsq_dbg_serial ? 1 : lookahead_env;   // (serial debug mode: nothing queued behind an undecided iteration)
    const bool damped = d_damp != nullptr;
    const double atol = 1e-6, btol = damped ? 0.5 : 1e-6, conlim = 1e8;  // lsmr.jl:54, il:255
    const long long rows = (sharded ? s->global_rows : (long long)m) + (damped ? n : 0);
    const int maxiter = (int)std::max<long long>(rows, n);               // lsmr.jl:55
    const unsigned epoch = (++c->mail_epoch) & 0x7fffffu;
    LsmrState *st = s->d_state;
    const int *done = &st->done;
    double *xs = s->d_rhs;
    double *pu = s->d_red, *pv = s->d_red + 4096;   // deferred-reduction partials (sum u~^2 / sum v~^2)
    // sum(u~x^2) partials are double-buffered: K3 reads iteration k's while writing k+1's
    double *pxb[2] = {s->d_red + 2 * 4096, s->d_red + 3 * 4096};
    int *npu = (int *)(s->d_red + 4 * 4096), *npv = npu + 1;
    int *npxb[2] = {npu + 2, npu + 3};

    if (J->kind == LSQ_MAT_CSC) LSQ_TRY(lsq_ensure_csr(J));
    // The three-launch iteration (lsq_lsmr3.h): both sliced layouts, x resident in LDS, one rank.  Otherwise K1 | K2 | K3 below.
    const bool four_env = getenv("LSQ_LSMR_FOUR_LAUNCHES") != nullptr;      // (read per solve: the tests flip it)
    const bool fused = !four_env && !sharded && J->kind == LSQ_MAT_CSC && J->srows.active && J->srows.ncw == 1 && J->scols.active &&
                       n >= 1 && m >= 1 && c->num_cus >= 8;
    double *const f3 = s->d_f3;
    double *const fx[2] = {xs, f3}, *const fhbar[2] = {s->d_hbar, f3 + n}, *const fh[2] = {s->d_h, f3 + 2 * (size_t)n};
    double *const fw = f3 + 3 * (size_t)n;                          // gather vector of the next J*v: (v~ .* P) .* s
    double *const fpu[2] = {f3 + 4 * (size_t)n, f3 + 4 * (size_t)n + 4096};
    int *const fnpu = (int *)(fpu[1] + 4096);                      // two counts
    LsmrHandoff *const fho = (LsmrHandoff *)(fpu[1] + 4096 + 8);   // the in-launch record of k_lsmr_fused
    double *const vset = fused ? s->d_t : s->d_v;                  // where the setup (and every K2) leaves v~
    if (fused) { pu = fpu[0]; npu = fnpu; }
    // computed once per Jacobian (reference: twice); row-sharded: the sum over the ranks' blocks -- the preconditioner is a
    // replicated n-vector and has to be the same on every rank
    const double *colsum = nullptr;
    LSQ_TRY(lsq_rowshard_colsum(s, J, &colsum));
    const int custom_p = s->precond_cb ? 1 : 0;
    if (custom_p) {   // preconditioner!(P, x, J, damp) on the host, damp still un-rooted (iterative_lsmr.jl:251-252)
        LSQ_HIP(hipStreamSynchronize(c->stream));
        if (s->precond_cb(s->d_P, J, d_damp, s->precond_user) != 0) {
            lsq_set_error("preconditioner callback reported failure");
            return LSQ_ECALLBACK;
        }
    }
    *(volatile unsigned long long *)c->h_mail = 0ull;
    const int gn = nvec_grid(c, n);
    const double *dgk = damped ? s->d_dg : nullptr;
    EpiV ev{done, 0, st, pu, npu, nullptr, nullptr, s->d_P, damped ? s->d_dg : nullptr, nullptr, s->d_v, pv, npv,
            nullptr, 0.0, 0.0, 0, 0};
    auto launch_begin = [&]() {
        long long gb = std::min<long long>(lsq_div_up(m > 0 ? m : 1, LSQ_NT), (long long)c->num_cus * 8);
        if (gb > 4096) gb = 4096;
        LSQ_LAUNCH(k_lsmr_begin, dim3((int)gb), dim3(LSQ_NT), 0, c->stream, m, d_y, st, pu, npu, atol,
                           btol, 1.0 / conlim, maxiter, epoch);
    };
    if (lm) {
        if (!(d_Jty && y_sumsq >= 0.0 && damped && lsq_lsmr_takes_lm_prep(s, J))) {
            lsq_set_error("lsmr: LM preparation requested where it does not apply");
            return LSQ_EARG;
        }
        LSQ_LAUNCH(k_lm_lsmr_setup, dim3(lsq_div_up(n, 1024)), dim3(1024), 0, c->stream, n, *lm, d_damp, s->d_P, s->d_dg,
                           s->d_ux, d_Jty, vset, st, pu, npu, y_sumsq, pv, npv, atol, btol, 1.0 / conlim, maxiter, epoch,
                           (const double *)(fused ? J->d_colscale : nullptr), (double *)(fused ? fw : nullptr));
        if (!fused)
        LSQ_LAUNCH(k_lsmr_update, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, st, c->d_mail, pu, npu, pv, npv,
                           (const double *)nullptr, (const int *)nullptr, pxb[0], npxb[0], dgk, s->d_ux, s->d_P,
                           s->d_v, s->d_h, s->d_hbar, xs, d_x, s->d_t, c->d_partials, lsq_ctr(c, 3));
        LSQ_HIP(hipGetLastError());
    } else if (d_Jty) {
        if (!(y_sumsq >= 0.0)) launch_begin();
        LSQ_LAUNCH(k_lsmr_setup, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, colsum, d_damp, s->d_P, s->d_dg,
                           s->d_ux, d_Jty, vset, st, pu, npu, y_sumsq >= 0.0 ? y_sumsq : -1.0, pv, npv, atol, btol,
                           1.0 / conlim, maxiter, epoch, custom_p, (const double *)(fused ? J->d_colscale : nullptr),
                           (double *)(fused ? fw : nullptr));
        if (!fused)
        LSQ_LAUNCH(k_lsmr_update, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, st, c->d_mail, pu, npu, pv, npv,
                           (const double *)nullptr, (const int *)nullptr, pxb[0], npxb[0], dgk, s->d_ux, s->d_P,
                           s->d_v, s->d_h, s->d_hbar, xs, d_x, s->d_t, c->d_partials, lsq_ctr(c, 3));
        LSQ_HIP(hipGetLastError());
    } else {
        LSQ_LAUNCH(k_lsmr_prep, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, colsum, d_damp, s->d_P,
                           s->d_dg, s->d_ux, custom_p);
        launch_begin();
        LSQ_HIP(hipGetLastError());
        // v~ = A'u (setup), then K3 in "first" mode
        if (fused) { ev.vout = vset; ev.cs = J->d_colscale; ev.wout = fw; }
        if (sharded) LSQ_TRY(rowshard_adjoint(s, J, d_y, ev, pu, npu));
        else LSQ_TRY(launch_product(J, 1, d_y, ev));
        // setup K3: u~x is still zero, so only sum(u~_y^2) enters beta_1; it already forms iteration 1's u~x
        if (!fused)
        LSQ_LAUNCH(k_lsmr_update, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, st, c->d_mail,
                           sharded ? (const double *)(s->d_xbuf + n) : (const double *)pu, sharded ? (const int *)s->d_one : (const int *)npu, pv, npv,
                           (const double *)nullptr, (const int *)nullptr, pxb[0], npxb[0], dgk, s->d_ux, s->d_P,
                           s->d_v, s->d_h, s->d_hbar, xs, d_x, s->d_t, c->d_partials, lsq_ctr(c, 3));
        LSQ_HIP(hipGetLastError());
    }

    int enq = 0, it = 0, istop = 0;
    bool finished = false;
    if (fused) {
        // ---------------------------------------------------------------------------------------------------------------
        // three launches per inner iteration: K1' (k_lsmr_fused: decides iteration j-1, updates the n-vectors, J*v of
        // iteration j) | K2a (k_sell_cols) | K2b (k_combine<EpiV>).  Launch j of K1' reads state / partials / x, hbar, h
        // number (j-1)&1 and writes number j&1; it reports iteration j-1 in the progress word.
        // ---------------------------------------------------------------------------------------------------------------
        static const bool no_dynamic3 = getenv("LSQ_NO_DYNAMIC_TAIL") != nullptr;
        const bool dynamic = tail && tail->fn && tail->dynamic && !no_dynamic3;
        const bool spec = tail && tail->fn && (tail->predict > 0 || dynamic);
        int planned = spec ? tail->predict : 0;
        int tail_at = 0, tail_ran_at = 0, hint_it = 0;
        // REACT instead of PRE-PLACE (round 6, LSQ_TAIL_REACT; profiles/r06/ab_tail_react.txt).  A guarded tail queued behind the
        // launch that is expected to commit the stop costs nothing when the guess holds, but ~50 us when it does not: five
        // launches of full grids that skip themselves stand between the solve's next product and the queue's head (measured:
        // perfect first guesses would gain 3.6 % on C4, LSQ_TAIL_ORACLE_SEQ).  In this mode the expected stop only makes that
        // launch CAUTIOUS, and nothing more is queued until it has reported (~8 us into it): if the solve is over the host
        // queues the tail, unguarded, while the cautious launch winds down; if not, the product of that launch is still
        // streaming for 20 us and the next launches are queued behind it as usual.  A wrong guess then costs a late product.
        static const bool react = [] { const char *e = getenv("LSQ_TAIL_REACT"); return e ? atoi(e) != 0 : LSQ_TAIL_REACT_DEFAULT; }();
        int wait_at = 0;             // react: the iteration whose commit (by a cautious launch) is awaited before anything else is queued
        bool wait_is_guess = false;  //   ... and that launch was cautious because of a PREDICTION (counted), not only because nothing was known
        double t1_prev = -1.0, t2_prev = -1.0, ratio2 = s->lsmr_ratio2;
        LsmrState *stb[2] = {s->d_state, s->d_state + 1};
        const LsqSell &S = J->srows;
        const int nxpad = (n + 1) & ~1;
        const size_t lds = (size_t)(nxpad + LSQ_SELL_ROWS_MAX) * sizeof(double);
        LSQ_TRY(lsq_set_lds(c, (const void *)k_lsmr_fused<0>, (LSQ_LDS_X_MAX + LSQ_SELL_ROWS_MAX) * sizeof(double)));
        // update workgroups: the CUs the sliced rows leave without a block (C4: 253 blocks on 256 CUs -> 3), at least 3
        int ub = c->num_cus - S.nblocks;
        ub = ub < 3 ? 3 : (ub > LSQ_FUSED_UB_MAX ? LSQ_FUSED_UB_MAX : ub);
        const int pb = std::max(1, std::min(S.nblocks, c->num_cus - ub));
        int k1 = 0;                  // K1' launches enqueued: enq (whole iterations) or enq + 1
        bool need_product = false;   // launch k1 went out commit-only: its product half is still owed (if the solve goes on)
        static const bool use_halves = getenv("LSQ_LSMR_HALVES") != nullptr;          // (A/B: commit-only + product-only launches)
        static const bool no_cautious = getenv("LSQ_LSMR_NO_CAUTIOUS") != nullptr;    // (A/B: always the plain fused launch)
        const int test_no_record = getenv("LSQ_TEST_EXCHANGE_TIMEOUT") ? 1 : 0;       // (read once per solve, not per launch)
        const size_t prof_base3[2] = {c->prof_ev[0].size(), c->prof_ev[1].size()};
        std::vector<int> prof_it3[2];
        unsigned long long spins = 0;
        // mode 0: the whole launch; 1: commit-only (update workgroups); 2: product-only, the half that a commit-only launch left out;
        // 3: the whole launch, cautious (the product workgroups wait for the decision before they stream)
        auto launch_k1 = [&](int mode) -> int {
            const int j = mode == 2 ? k1 : k1 + 1, in = (j - 1) & 1, out = j & 1;
            LsmrFused a;
            a.st_in = stb[in]; a.st_out = stb[out]; a.mail = c->d_mail;
            a.pu_in = fpu[in]; a.npu_in = fnpu + in; a.pu_out = fpu[out]; a.npu_out = fnpu + out;
            a.px_in = (damped && j > 1) ? pxb[in] : nullptr; a.npx_in = npxb[in];
            a.px_out = pxb[out]; a.npx_out = npxb[out];
            a.pv = pv; a.npv = npv;
            a.vt = vset; a.w = fw; a.P = s->d_P; a.dg = dgk;
            a.h_in = fh[in]; a.hbar_in = fhbar[in]; a.x_in = fx[in];
            a.h_out = fh[out]; a.hbar_out = fhbar[out]; a.x_out = fx[out];
            a.v = s->d_v; a.xout = d_x; a.ux = s->d_ux;
            a.uold = j == 1 ? d_y : s->d_u; a.unew = s->d_u;
            a.n = n; a.ub = mode == 2 ? 0 : ub;
            a.cautious = mode == 3 ? 1 : 0;
            a.test_no_record = test_no_record;                      // (test hook: the in-launch hand-off fails)
            a.ho = fho;
            a.tag_prev = j > 1 ? s->f3_tag : 0u;                    // (mode 2 has no update workgroups: unused there)
            if (mode != 2 && ++s->f3_tag == 0u) s->f3_tag = 1u;    // a counter per solver (= per record buffer): every older record
            a.tag = s->f3_tag;                                      // carries another value; 0 is the zeroed buffer
            if (mode == 2) a.st_in = stb[out];                      // (the state its commit-only half has committed)
            const int grid = mode == 1 ? ub : (mode == 2 ? pb : pb + ub);
            if (mode != 1 && (c->prof_kernels & 1)) lsq_prof_mark(c, 0, 0);
            hipEvent_t e0, e1;
            if (mode != 1 && lsq_prof_take(c, &e0, &e1)) {
                LSQ_LAUNCH_TIMED(k_lsmr_fused<0>, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, e0, e1, 0, sell_dev(S), S.wrows, m, nxpad, a);
                prof_it3[0].push_back(j);
            } else {
                LSQ_LAUNCH(k_lsmr_fused<0>, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, sell_dev(S), S.wrows, m, nxpad, a);
            }
            if (mode != 1 && (c->prof_kernels & 1)) lsq_prof_mark(c, 0, 1);
            LSQ_HIP(hipGetLastError());
            if (mode != 2) ++k1;
            need_product = mode == 1;
            return LSQ_OK;
        };
        auto launch_k2 = [&]() -> int {      // K2 of iteration enq + 1 (its K1' is launch k1 == enq + 1)
            const int j = enq + 1, cur = j & 1;
            EpiV e2{&stb[cur]->done, 0, stb[cur], fpu[cur], fnpu + cur, damped ? pxb[cur] : nullptr, npxb[cur], s->d_P, dgk,
                    damped ? s->d_ux : nullptr, s->d_v, pv, npv, nullptr, 0.0, 0.0, 0, 0};
            e2.vout = vset; e2.cs = J->d_colscale; e2.wout = fw;
            if (c->prof_kernels & 2) lsq_prof_mark(c, 1, 0);
            const size_t before = c->prof_ev[1].size();
            LSQ_TRY(launch_sell_cols<false>(J, s->d_u, &stb[cur]->done));
            if (c->prof_ev[1].size() > before) prof_it3[1].push_back(j);
            if (c->prof_kernels & 2) lsq_prof_mark(c, 1, 1);
            const int nb = lsq_div_up(n, LSQ_CMB_COLS);
            LSQ_LAUNCH((k_combine<EpiV>), dim3(std::min(nb, 2048)), dim3(LSQ_NT), 0, c->stream, J->scols.d_part, n, J->scols.ngw, e2, nb,
                       J->d_colscale);
            LSQ_HIP(hipGetLastError());
            ++enq;
            return LSQ_OK;
        };
        while (!finished) {
            unsigned long long w = *(volatile unsigned long long *)c->h_mail;
            const bool reported = (unsigned)(w >> 41) == epoch;
            if (reported) {
                it = (int)(w & 0xffffffffu);
                if ((w >> 40) & 1ull) {
                    istop = (int)((w >> 32) & 0xff);
                    finished = true;
                    break;
                }
            }
            if (spec && reported && tail_at > 0 && it >= tail_at) {   // that tail found the solve unfinished and skipped itself
                c->tail_spec[0]++;
                c->tail_spec[1]++;
                tail_at = 0;
                planned = 0;
            }
            if (wait_at > 0 && reported && it >= wait_at) {           // react: the cautious launch found the solve unfinished
                if (wait_is_guess) {
                    c->tail_spec[0]++;
                    c->tail_spec[1]++;
                    planned = 0;
                }
                wait_at = 0;
            }
            if (dynamic && reported && it > hint_it && it >= 1) {
                const double h1 = c->h_mail->test1, h2 = c->h_mail->test2;
                const unsigned long long w2 = *(volatile unsigned long long *)c->h_mail;
                if (w2 == w && h1 > 0.0 && h2 > 0.0) {
                    if (t2_prev > 0.0 && h2 < t2_prev) ratio2 = std::min(0.95, std::max(1e-4, h2 / t2_prev));
                    int k = 1 << 20;
                    if (h2 > atol && ratio2 < 1.0) k = std::min(k, (int)std::ceil(std::log(atol / h2) / std::log(ratio2) - 1e-9));
                    if (t1_prev > 0.0 && h1 < 0.98 * t1_prev && h1 > btol)
                        k = std::min(k, (int)std::ceil(std::log(btol / h1) / std::log(h1 / t1_prev) - 1e-9));
                    if (k < (1 << 20) && tail_at == 0) planned = it + std::max(1, k);
                    t1_prev = h1; t2_prev = h2; hint_it = it;
                }
            }
            // a guarded tail sits behind the commit of iteration tail_at / react: a cautious launch is deciding iteration wait_at
            const bool hold = (tail_at > 0 && it < tail_at) || (wait_at > 0 && it < wait_at);
            if (!hold) {
                if (k1 == enq + 1 && need_product) {             // the tail behind a commit-only launch skipped itself: the solve goes on
                    LSQ_TRY(launch_k1(2));
                    spins = 0;
                    continue;
                }
                if (k1 == enq + 1) {                             // the rest of iteration enq + 1
                    if (!react && spec && tail_at == 0 && planned > 0 && enq >= planned && enq >= 1 && enq > it) {
                        // (the prediction arrived between the two halves: the launch that commits its iteration is the newest one)
                        LSQ_TRY(tail->fn(&stb[k1 & 1]->notdone, tail->user));
                        tail_at = enq;
                        continue;
                    }
                    LSQ_TRY(launch_k2());
                    spins = 0;
                    continue;
                }
                // K1' number enq + 1 commits iteration enq: it does not count against the look-ahead (nothing is decided without it)
                if (enq - it < lookahead + 1 && enq < maxiter + 1) {
                    // the caller's tail right behind the launch that commits the planned last iteration (or the newest one, if
                    // the prediction names an iteration whose commit is already in the queue) -- and that launch commit-only: if
                    // the plan holds there is no next product
                    const bool place = spec && tail_at == 0 && planned > 0 && enq >= planned && enq >= 1;
                    // ... and cautious also where nothing is known yet: the launch that decides iteration 1, unless the previous
                    // solve was a long one
                    const bool unknown = spec && enq == 1 && hint_it == 0 && tail->predict <= 3;
                    const int mode = no_cautious ? 0 : (place ? (use_halves ? 1 : 3) : (unknown && !use_halves ? 3 : 0));
                    LSQ_TRY(launch_k1(mode));
                    if (react && mode == 3) {                    // nothing behind it until it has decided
                        wait_at = enq;
                        wait_is_guess = place;
                    } else if (place) {
                        LSQ_TRY(tail->fn(&stb[k1 & 1]->notdone, tail->user));
                        tail_at = enq;
                    }
                    spins = 0;
                    continue;
                }
            }
            if (c->idle_hook) lsq_run_idle_hook(c);
            if ((++spins & 0x3ffu) != 0) continue;
            const hipError_t sq = hipStreamQuery(c->stream);
            if (sq != hipSuccess && sq != hipErrorNotReady) {
                lsq_set_error("HIP error inside the LSMR iteration: %s", hipGetErrorString(sq));
                return LSQ_EHIP;
            }
            if (sq == hipSuccess) {
                w = *(volatile unsigned long long *)c->h_mail;
                if ((unsigned)(w >> 41) == epoch && ((w >> 40) & 1ull)) continue;
                if ((unsigned)(w >> 41) == epoch && (int)(w & 0xffffffffu) >= k1 - 1) continue;
                LsmrState hs;      // mailbox not visible (should not happen with coherent host memory): read the state
                LSQ_HIP(hipMemcpy(&hs, stb[k1 & 1], sizeof(hs), hipMemcpyDeviceToHost));
                it = hs.iter;
                if (hs.done) {
                    istop = hs.istop;
                    finished = true;
                } else if (enq >= maxiter + 1) {
                    lsq_set_error("lsmr: iteration budget exhausted without a stop rule");
                    return LSQ_EHIP;
                }
            }
        }
        for (int k = 0; k < 2; ++k) {     // timed launches queued beyond the stop did no work: drop their samples
            auto &v = c->prof_ev[k];
            const int last_working = k == 0 ? it + 1 : it;   // K1' number it + 1 committed the stop but skipped its product
            while (v.size() > prof_base3[k] && !prof_it3[k].empty() && prof_it3[k].back() > (k == 0 ? last_working - 1 : last_working)) {
                hipEventDestroy(v.back()); v.pop_back();
                hipEventDestroy(v.back()); v.pop_back();
                prof_it3[k].pop_back();
            }
        }
        if (istop == 99) {     // (a product workgroup gave up waiting for workgroup 0's record: lsq_lsmr3.h)
            lsq_set_error("lsmr: the in-launch hand-off of k_lsmr_fused timed out");
            return LSQ_EHIP;
        }
        s->last_iter = it;
        s->last_istop = istop;
        if (nmul) *nmul = 2 * it;
        if (tail_at > 0) {
            c->tail_spec[0]++;
            if (it > tail_at) c->tail_spec[1]++;
            else tail_ran_at = tail_at;
        }
        if (wait_at > 0 && wait_is_guess) c->tail_spec[0]++;      // react: the predicted stop was the stop (a right guess)
        if (dynamic && t2_prev > 0.0) s->lsmr_ratio2 = ratio2;
        if (tail && tail->fn && tail_ran_at == 0) LSQ_TRY(tail->fn(nullptr, tail->user));
        return LSQ_OK;
    }

    EpiU eu{done, 0, st, d_y, s->d_u, n, s->d_dg, s->d_t, s->d_ux, pu, npu, nullptr, 0.0};
    ev.ux = damped ? s->d_ux : nullptr;

    // Where the caller's tail goes (LsmrTail, lsq_solver.h).  `planned`: the iteration behind which it should be queued --
    // first the caller's guess (the previous solve's count), from iteration 1 on the solve's own prediction: the hints K3
    // publishes are test1 = |r|/|b| and test2 = |A'r|/(|A||r|); a solve is over when test1 <= btol (+ a 1e-6-sized term) or
    // test2 <= atol (lsmr.jl:224-231; the other rules do not fire on these operators), and on the damped, Jacobi-
    // preconditioned operators of an LM run test2 falls geometrically (C4: x 0.07-0.09 per iteration, every iteration).
    static const bool no_dynamic = getenv("LSQ_NO_DYNAMIC_TAIL") != nullptr;
    const bool dynamic = tail && tail->fn && tail->dynamic && !sharded && !no_dynamic;
    const bool spec = tail && tail->fn && !sharded && (tail->predict > 0 || dynamic);
    int planned = spec ? tail->predict : 0;
    int tail_at = 0;             // iteration behind which the latest guarded tail was enqueued (0: none that is still undecided)
    int tail_ran_at = 0;         // ... and the same once it is known (or bound) to have run: the stop iteration was <= tail_at
    int hint_it = 0;             // last iteration whose hints went into the prediction
    double t1_prev = -1.0, t2_prev = -1.0, ratio2 = s->lsmr_ratio2;
    const size_t prof_base[2] = {c->prof_ev[0].size(), c->prof_ev[1].size()};
    std::vector<int> prof_iter[2];   // iteration number of every timed launch of this solve
    unsigned long long spins = 0;
    while (!finished) {
        unsigned long long w = *(volatile unsigned long long *)c->h_mail;
        if ((unsigned)(w >> 41) == epoch) {
            it = (int)(w & 0xffffffffu);
            if ((w >> 40) & 1ull) {
                istop = (int)((w >> 32) & 0xff);
                finished = true;
                break;
            }
        }
        const bool reported_now = (unsigned)(w >> 41) == epoch;
        if (spec && reported_now && tail_at > 0 && it >= tail_at) {   // (not done: `finished` left the loop above)
            // the tail behind iteration tail_at found the solve unfinished and skipped itself: a wrong guess
            c->tail_spec[0]++;
            c->tail_spec[1]++;
            tail_at = 0;
            planned = 0;
        }
        if (dynamic && reported_now && it > hint_it && it >= 1) {
            const double h1 = c->h_mail->test1, h2 = c->h_mail->test2;
            const unsigned long long w2 = *(volatile unsigned long long *)c->h_mail;
            if (w2 == w && h1 > 0.0 && h2 > 0.0) {        // (the pair belongs to iteration `it`: the word did not move meanwhile)
                if (t2_prev > 0.0 && h2 < t2_prev) ratio2 = std::min(0.95, std::max(1e-4, h2 / t2_prev));
                int k = 1 << 20;
                if (h2 > atol && ratio2 < 1.0) k = std::min(k, (int)std::ceil(std::log(atol / h2) / std::log(ratio2) - 1e-9));
                if (t1_prev > 0.0 && h1 < 0.98 * t1_prev && h1 > btol)       // test1 still falling: when does it reach btol?
                    k = std::min(k, (int)std::ceil(std::log(btol / h1) / std::log(h1 / t1_prev) - 1e-9));
                if (k < (1 << 20) && tail_at == 0) planned = it + std::max(1, k);
                t1_prev = h1; t2_prev = h2; hint_it = it;
            }
        }
        // the guarded tail: right behind the planned last iteration, as soon as that one is enqueued (or behind the newest
        // enqueued, still unreported one if the prediction names an iteration that is already in the queue)
        if (spec && tail_at == 0 && planned > 0 && enq >= planned && enq > it) {
            LSQ_TRY(tail->fn(&st->notdone, tail->user));
            tail_at = enq;
        }
        // Row-sharded: every rank must enqueue the SAME number of inner iterations (each carries a collective), so the count
        // cannot depend on when this host happens to see the mailbox: iterations go out in chunks of `lookahead`, and the
        // next chunk only once the last one has reported "not done" -- ceil(stop / lookahead) * lookahead on every rank.
        // (and the first chunk only once the SETUP has reported: a solve that is over before it starts -- A'b = 0 -- must not
        //  get a chunk on the ranks whose host looked too early)
        const bool reported = (unsigned)(w >> 41) == epoch;
        // (a speculative tail sits behind iteration tail_at: nothing more is queued until that iteration has reported)
        const bool hold = tail_at > 0 && enq == tail_at && it < tail_at;
        const bool want = sharded ? (reported && enq == it && enq < maxiter) : (!hold && enq - it < lookahead && enq < maxiter);
        if (want) {
          const int chunk = sharded ? (int)std::min<long long>(lookahead, (long long)maxiter - enq) : 1;
          for (int q = 0; q < chunk; ++q) {
            if (c->prof_kernels & 1) lsq_prof_mark(c, 0, 0);
            {
                const size_t before = c->prof_ev[0].size();
                LSQ_TRY(launch_product(J, 0, s->d_t, eu));   // K1
                if (c->prof_ev[0].size() > before) prof_iter[0].push_back(enq + 1);
            }
            if (c->prof_kernels & 1) lsq_prof_mark(c, 0, 1);
            eu.uold = s->d_u;                            // after the first iteration u~ lives in d_u
            const int cur = enq & 1;                     // px buffer holding this iteration's sum(u~x^2)
            ev.px = damped ? pxb[cur] : nullptr;
            ev.npx = npxb[cur];
            if (c->prof_kernels & 2) lsq_prof_mark(c, 1, 0);
            {
                const size_t before = c->prof_ev[1].size();
                if (sharded) LSQ_TRY(rowshard_adjoint(s, J, s->d_u, ev, pu, npu));   // K2 with the ranks' sum in the middle
                else LSQ_TRY(launch_product(J, 1, s->d_u, ev));   // K2
                if (c->prof_ev[1].size() > before) prof_iter[1].push_back(enq + 1);
            }
            if (c->prof_kernels & 2) lsq_prof_mark(c, 1, 1);
            LSQ_LAUNCH(k_lsmr_update, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, st, c->d_mail,
                               sharded ? (const double *)(s->d_xbuf + n) : (const double *)pu,
                               sharded ? (const int *)s->d_one : (const int *)npu,
                               pv, npv, (const double *)(damped ? pxb[cur] : nullptr), (const int *)npxb[cur],
                               pxb[cur ^ 1], npxb[cur ^ 1], dgk, s->d_ux, s->d_P, s->d_v, s->d_h, s->d_hbar, xs,
                               d_x, s->d_t, c->d_partials, lsq_ctr(c, 3));
            LSQ_HIP(hipGetLastError());
            ++enq;
          }
            spins = 0;
            continue;
        }
        if (c->idle_hook) lsq_run_idle_hook(c);   // look-ahead window full: the device has work for a while
        if ((++spins & 0x3ffu) != 0) continue;
        const hipError_t sq = hipStreamQuery(c->stream);
        if (sq != hipSuccess && sq != hipErrorNotReady) {   // launch failure / device fault: the mailbox will never move
            lsq_set_error("HIP error inside the LSMR iteration: %s", hipGetErrorString(sq));
            return LSQ_EHIP;
        }
        if (sq == hipSuccess) {
            // stream drained: whatever the mailbox shows now is final for the enqueued work
            w = *(volatile unsigned long long *)c->h_mail;
            if ((unsigned)(w >> 41) == epoch && ((w >> 40) & 1ull)) continue;
            if ((unsigned)(w >> 41) == epoch && (int)(w & 0xffffffffu) >= enq) continue;
            // mailbox not visible (should not happen with coherent host memory): read the state
            LsmrState hs;
            LSQ_HIP(hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost));
            it = hs.iter;
            if (hs.done) {
                istop = hs.istop;
                finished = true;
            } else if (enq >= maxiter) {
                lsq_set_error("lsmr: iteration budget exhausted without a stop rule");
                return LSQ_EHIP;
            }
        }
    }
    // instrumentation: launches queued beyond the stopping iteration returned on their first
    // instruction -- drop their samples so the reported average is over working launches only
    for (int k = 0; k < 2; ++k) {
        auto &v = c->prof_ev[k];
        // sample j of this solve (j = 0, 1, ...) belongs to iteration prof_iter[k][j] (1-based)
        while (v.size() > prof_base[k] && !prof_iter[k].empty() && prof_iter[k].back() > it) {
            hipEventDestroy(v.back());
            v.pop_back();
            hipEventDestroy(v.back());
            v.pop_back();
            prof_iter[k].pop_back();
        }
    }
    s->last_iter = it;
    s->last_istop = istop;
    if (nmul) *nmul = 2 * it;  // lsmr.jl:236 ch.mvps
    // the latest guarded tail ran iff the solve was over when its kernels reached the device: stop iteration <= tail_at
    // (tails that skipped themselves were counted, as wrong guesses, when their iteration reported)
    if (tail_at > 0) {
        c->tail_spec[0]++;
        if (it > tail_at) c->tail_spec[1]++;
        else tail_ran_at = tail_at;
    }
    if (dynamic && t2_prev > 0.0) s->lsmr_ratio2 = ratio2;
    if (tail && tail->fn && tail_ran_at == 0) LSQ_TRY(tail->fn(nullptr, tail->user));
    return LSQ_OK;
}
