// Solver plug point (ldiv!), the Levenberg-Marquardt / Dogleg trust-region loops on device
// buffers (levenberg_marquardt.jl:39-144, dogleg.jl:41-203: host control flow, device arrays),
// and the built-in device-side tanh model used by the synthetic benchmarks.
#include <chrono>
#include <cmath>
#include <cstring>

#include "lsq_solver.h"
#include "lsq_spmv.h"

// constants that shape trajectories (types.jl:107-111, dogleg.jl:38-39)
static constexpr double MIN_DELTA = 1e-16, MAX_DELTA = 1e16, MIN_STEP_QUALITY = 1e-3;
static constexpr double MIN_DIAGONAL = 1e-6, MAX_DIAGONAL = 1e32;
static constexpr double DECREASE_THRESHOLD = 0.25, INCREASE_THRESHOLD = 0.75;

// scalar slots used by the loops (ctx->d_slots)
enum { SL_GRAD = 8, SL_DX = 9, SL_NONFIN = 10, SL_TRIAL = 11, SL_PRED = 12, SL_SSR = 13, SL_W0 = 14, SL_W1 = 15,
       SL_W2 = 16, SL_SUM = 17 };

// ---------------------------------------------------------------------------------------------
// solver objects
// ---------------------------------------------------------------------------------------------
extern "C" int lsq_solver_create(lsq_ctx *c, lsq_mat *J, int kind, int for_lm, lsq_solver **out) {
    LSQ_RANGE("lsq_solver_create");
    if (!c || !J || !out) return LSQ_EARG;
    if (kind == LSQ_QR && J->kind != LSQ_MAT_DENSE) {
        lsq_set_error("solver QR() is not available for sparse Jacobians. Choose between Cholesky() and LSMR()");
        return LSQ_EARG;  // types.jl:115-117
    }
    if (kind == LSQ_CHOLESKY && J->kind != LSQ_MAT_DENSE) {
        lsq_set_error("MethodError: no AbstractAllocatedSolver for Cholesky() with a sparse Jacobian "
                      "(dense_cholesky.jl:19 requires a StridedVecOrMat)");
        return LSQ_EARG;
    }
    LSQ_HIP(hipSetDevice(c->device));
    lsq_solver *s = new lsq_solver();
    s->ctx = c;
    s->kind = kind;
    s->for_lm = for_lm ? 1 : 0;
    s->m = J->m;
    s->n = J->n;
    int st = (kind == LSQ_LSMR) ? lsq_lsmr_alloc(s) : lsq_dense_solver_alloc(s);
    if (st != LSQ_OK) {
        delete s;
        return st;
    }
    *out = s;
    return LSQ_OK;
}

extern "C" int lsq_solver_destroy(lsq_solver *s) {
    if (!s) return LSQ_OK;
    hipStreamSynchronize(s->ctx->stream);
    if (s->kind == LSQ_LSMR) lsq_lsmr_free(s);
    else lsq_dense_solver_free(s);
    delete s;
    return LSQ_OK;
}

// include/lsqhip.h: "calls that return host scalars synchronise the stream".  The LSMR driver learns about the stop from a
// progress word that the committing workgroup publishes BEFORE its sibling workgroups have finished writing x (lsq_lsmr3.h:
// relaxed publish), and launches of the look-ahead may still be queued: the public entry points drain the stream
// before they hand nmul back, so that x may be read from another stream or with a blocking copy.  (Polled, not
// hipStreamSynchronize: what is left in the queue is microseconds of work, a sleeping wait wakes up after ~50 us.)  The LM /
// Dogleg loops call lsq_lsmr_solve directly and stay stream-ordered.
static int lsmr_drain(lsq_solver *s) {
    for (;;) {
        const hipError_t q = hipStreamQuery(s->ctx->stream);
        if (q == hipSuccess) return LSQ_OK;
        if (q != hipErrorNotReady) {
            lsq_set_error("HIP error behind an LSMR solve: %s", hipGetErrorString(q));
            return LSQ_EHIP;
        }
    }
}

extern "C" int lsq_ldiv(lsq_solver *s, lsq_mat *J, const double *y, double *x, int *nmul) {
    LSQ_RANGE("lsq_ldiv");
    if (!s || !J || !y || !x) return LSQ_EARG;
    if (s->for_lm && s->kind == LSQ_QR) {
        lsq_set_error("ldiv!: this QR solver was allocated for LevenbergMarquardt (damped)");
        return LSQ_EDIM;
    }
    switch (s->kind) {
    case LSQ_LSMR: LSQ_TRY(lsq_lsmr_solve(s, J, y, nullptr, x, nmul)); return lsmr_drain(s);
    case LSQ_CHOLESKY: return lsq_cholesky_solve(s, J, y, nullptr, x, nmul);
    default: return lsq_qr_solve(s, J, y, nullptr, x, nmul);
    }
}

extern "C" int lsq_ldiv_damped(lsq_solver *s, lsq_mat *J, const double *y, double *damp, double *x, int *nmul) {
    LSQ_RANGE("lsq_ldiv_damped");
    if (!s || !J || !y || !x || !damp) return LSQ_EARG;
    switch (s->kind) {
    case LSQ_LSMR: LSQ_TRY(lsq_lsmr_solve(s, J, y, damp, x, nmul)); return lsmr_drain(s);
    case LSQ_CHOLESKY: return lsq_cholesky_solve(s, J, y, damp, x, nmul);
    default: return lsq_qr_solve(s, J, y, damp, x, nmul);
    }
}

extern "C" int lsq_solver_set_preconditioner(lsq_solver *s, lsq_precond_callback cb, void *user) {
    if (!s) return LSQ_EARG;
    if (s->kind != LSQ_LSMR && cb) {
        lsq_set_error("preconditioners apply to the LSMR solver only (types.jl:82-86)");
        return LSQ_EARG;
    }
    s->precond_cb = cb;
    s->precond_user = user;
    return LSQ_OK;
}

extern "C" int lsq_solver_qr_panel(const lsq_solver *s, int *kind) {
    if (!s || !kind) { lsq_set_error("lsq_solver_qr_panel: null argument"); return LSQ_EARG; }
    *kind = s->last_qr_panel;
    return LSQ_OK;
}
extern "C" int lsq_solver_qr_path(const lsq_solver *s, int *path) {
    if (!s || !path) { lsq_set_error("lsq_solver_qr_path: null argument"); return LSQ_EARG; }
    *path = s->last_qr_path;
    return LSQ_OK;
}

extern "C" int lsq_solver_stats(const lsq_solver *s, int h_giveups[4], int h_paused[4]) {
    if (!s) { lsq_set_error("lsq_solver_stats: null argument"); return LSQ_EARG; }
    const LsqFallback *fb[4] = {&s->fb_tiles, &s->fb_pipe, &s->fb_qrx, &s->fb_cholqr};
    for (int i = 0; i < 4; ++i) {
        if (h_giveups) h_giveups[i] = fb[i]->giveups;
        if (h_paused) h_paused[i] = fb[i]->cooldown;
    }
    return LSQ_OK;
}

extern "C" int lsq_solver_chol_path(const lsq_solver *s, int *path) {
    if (!s || !path) { lsq_set_error("lsq_solver_chol_path: null argument"); return LSQ_EARG; }
    *path = s->last_chol_path;
    return LSQ_OK;
}

extern "C" int lsq_solver_info(const lsq_solver *s, int *iter, int *istop, int *rank) {
    if (iter) *iter = s->last_iter;
    if (istop) *istop = s->last_istop;
    if (rank) {
        *rank = -1;
        if (s->kind == LSQ_QR && s->d_info) {
            LSQ_HIP(hipStreamSynchronize(s->ctx->stream));
            LSQ_HIP(hipMemcpy(rank, s->d_info, sizeof(int), hipMemcpyDeviceToHost));
        }
    }
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// fused outer-loop kernels
// ---------------------------------------------------------------------------------------------
// LM damping: dtd = clamp(colsum, 1e-6*mean, 1e32*mean) / Delta  (levenberg_marquardt.jl:82-86)
__global__ void __launch_bounds__(1024)
k_lm_damp(int n, const double *__restrict__ colsum, double inv_delta, double *__restrict__ dtd) {
    __shared__ double sh[16];
    __shared__ double s_mean;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) acc += colsum[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        s_mean = t / n;
    }
    __syncthreads();
    const double lo = MIN_DIAGONAL * s_mean, hi = MAX_DIAGONAL * s_mean;
    for (int i = threadIdx.x; i < n; i += 1024) {
        double v = colsum[i];
        v = v > hi ? hi : (v < lo ? lo : v);
        dtd[i] = v * inv_delta;  // rmul!(dtd, 1/Delta)
    }
}

// One-workgroup variants for n <= LSQ_ONE_WG_N: a launch costs ~4 us whatever it does, so the
// n-length bookkeeping of an outer iteration is packed into as few launches as possible.
constexpr int LSQ_ONE_WG_N = 16384;

// k_lm_damp + k_gradnorm (levenberg_marquardt.jl:82-86 and :102-104's norm)
__global__ void __launch_bounds__(1024)
k_lm_damp_grad(int n, const double *__restrict__ colsum, double inv_delta, double *__restrict__ dtd,
               const double *__restrict__ g, const double *__restrict__ x, const double *__restrict__ lo,
               const double *__restrict__ hi, double *out_grad) {
    constexpr int R = LSQ_ONE_WG_N / 1024;
    __shared__ double sh[16];
    __shared__ double s_mean;
    double cs[R], gv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {   // every load is issued before the first use
        const int i = threadIdx.x + k * 1024;
        cs[k] = i < n ? colsum[i] : 0.0;
        gv[k] = i < n ? g[i] : 0.0;
    }
    double acc = 0.0, mg = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = threadIdx.x + k * 1024;
        if (i < n) {
            acc += cs[k];
            double gi = gv[k];
            if (lo && x[i] <= lo[i] && gi > 0.0) gi = 0.0;
            else if (hi && x[i] >= hi[i] && gi < 0.0) gi = 0.0;
            double a = fabs(gi);
            if (isnan(a)) a = INFINITY;
            mg = fmax(mg, a);
        }
    }
    acc = wave_sum(acc);
    mg = wave_max(mg);
    __shared__ double shm[16];
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = acc;
        shm[threadIdx.x >> 6] = mg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, tm = shm[0];
        for (int w = 0; w < 16; ++w) t += sh[w];
        for (int w = 1; w < 16; ++w) tm = fmax(tm, shm[w]);
        s_mean = t / n;
        *out_grad = tm;
    }
    __syncthreads();
    const double lo_d = MIN_DIAGONAL * s_mean, hi_d = MAX_DIAGONAL * s_mean;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = threadIdx.x + k * 1024;
        if (i < n) {
            double v = cs[k];
            v = v > hi_d ? hi_d : (v < lo_d ? lo_d : v);
            dtd[i] = v * inv_delta;  // rmul!(dtd, 1/Delta)
        }
    }
}

// Dogleg scaling: dtd = clamp(colsum, 1e-6, 1e32) (dogleg.jl:85-90)
__global__ void __launch_bounds__(LSQ_NT)
k_dl_scale(int n, const double *__restrict__ colsum, double *__restrict__ dtd) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        double v = colsum[i];
        dtd[i] = v > MAX_DIAGONAL ? MAX_DIAGONAL : (v < MIN_DIAGONAL ? MIN_DIAGONAL : v);
    }
}

// g = J'f with the projected max-norm fused (levenberg_marquardt.jl:102-104, dogleg.jl:99-101)
struct EpiGrad {
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *g;
    double *partials;
    unsigned *counter;
    __device__ void seg(int j, double dot, double &) const { g[j] = dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

// predicted residual: sum((J dx - f)^2) without storing fpredict (levenberg_marquardt.jl:114-117)
struct EpiPredict {
    static constexpr bool REDUCE = true;
    const int *done;
    int extra_blocks;
    const double *f;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int i, double dot, double &racc) const { seg_pre(i, dot, f[i], racc); }
    using has_pre = void;
    __device__ double pre(int i) const { return f[i]; }
    __device__ void seg_pre(int, double dot, double fi, double &racc) const {
        double r = dot - fi;
        racc += r * r;
    }
    __device__ void extra(int, double &) const {}
    // the last kernel of an outer iteration also hands the iteration's scalars to the host
    // (what k_publish_slots would do in a launch of its own)
    LsqSlotPublish pub;
    __device__ void finalize(double t) const {
        *out = t;
        if (pub.count > 0) {
            for (int i = 0; i < pub.count; ++i)
                __hip_atomic_store(pub.dst + i, pub.src + i == out ? t : pub.src[i], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub.seq_word, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

// sum((J d)^2) for the Cauchy step length (dogleg.jl:109-111)
struct EpiSumsq {
    static constexpr bool REDUCE = true;
    const int *done;
    int extra_blocks;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int, double dot, double &racc) const { racc += dot * dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double t) const { *out = t; }
};

// x_trial = x - dx, max|dx| and first non-finite index of x_trial in one pass
__global__ void __launch_bounds__(LSQ_NT)
k_step(int n, const double *__restrict__ x, const double *__restrict__ dx, double *__restrict__ xt,
       double *partials, unsigned *counters, double *out_dx, double *out_nonfin, double *__restrict__ t_out,
       double *__restrict__ s_out, const int *skip = nullptr) {
    __shared__ double sh[LSQ_NT / 64];
    if (skip && *skip) return;       // (queued behind an LSMR solve that turned out not to be over: LsmrTail)
    double mx = 0.0, code = 0.0;
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        double d = dx[i];
        double v = x[i] + -1.0 * d;  // axpy!(-1, dx, x)
        xt[i] = v;
        if (t_out) {                 // the built-in model's tanh(x_trial) and 1 - tanh^2 (k_tanh_sfac), while x_trial is in a register
            const double th = tanh(v);
            t_out[i] = th;
            s_out[i] = 1.0 - th * th;
        }
        double a = fabs(d);
        if (isnan(a)) a = INFINITY;
        mx = fmax(mx, a);
        if (!isfinite(v) && code == 0.0) code = 1e15 - (double)(i + 1);
    }
    // both maxima ride one ticket round: partials[b] and partials[LSQ_MAX_GRID + b]
    double b2 = block_max<LSQ_NT>(code, sh);
    if (threadIdx.x == 0) __hip_atomic_store(&partials[LSQ_MAX_GRID + blockIdx.x], b2, RLX_AGENT);
    double b1 = block_max<LSQ_NT>(mx, sh);
    const int nb = gridDim.x;
    if (grid_reduce<LSQ_NT, true>(b1, partials, counters, nb, sh, [=](double t) { *out_dx = t; })) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nb; i += LSQ_NT) acc = fmax(acc, __hip_atomic_load(&partials[LSQ_MAX_GRID + i], RLX_AGENT));
        double t = block_max<LSQ_NT>(acc, sh);
        if (threadIdx.x == 0) *out_nonfin = (t == 0.0) ? -1.0 : (1e15 - t) - 1.0;
    }
}

__global__ void __launch_bounds__(LSQ_NT)
k_revert(int n, const double *__restrict__ xt, const double *__restrict__ dx, double *__restrict__ x) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT)
        x[i] = xt[i] + 1.0 * dx[i];  // axpy!(1, dx, x): (x - dx) + dx, as the reference computes it
}

// max |g_i| with the active-bound projection (utils.jl:39-55) -> slot
__global__ void __launch_bounds__(LSQ_NT)
k_gradnorm(int n, const double *__restrict__ g, const double *__restrict__ x, const double *__restrict__ lo,
           const double *__restrict__ hi, double *partials, unsigned *counter, double *out) {
    __shared__ double sh[LSQ_NT / 64];
    double m = 0.0;
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        double gi = g[i];
        if (lo && x[i] <= lo[i] && gi > 0.0) gi = 0.0;
        else if (hi && x[i] >= hi[i] && gi < 0.0) gi = 0.0;
        double a = fabs(gi);
        if (isnan(a)) a = INFINITY;
        m = fmax(m, a);
    }
    double b = block_max<LSQ_NT>(m, sh);
    grid_reduce<LSQ_NT, true>(b, partials, counter, gridDim.x, sh, [=](double t) { *out = t; });
}

__global__ void __launch_bounds__(LSQ_NT)
k_sumsq_slot(long long n, const double *__restrict__ x, double *partials, unsigned *counter, double *out,
             LsqSlotPublish pub) {
    __shared__ double sh[LSQ_NT / 64];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * LSQ_NT;
    long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {   // four loads in flight per lane
        const double v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
        acc += v0 * v0;
        acc += v1 * v1;
        acc += v2 * v2;
        acc += v3 * v3;
    }
    for (; i < n; i += stride) {
        double v = x[i];
        acc += v * v;
    }
    double b = block_sum<LSQ_NT>(acc, sh);
    grid_reduce<LSQ_NT>(b, partials, counter, gridDim.x, sh, [=](double t) {
        *out = t;
        if (pub.count > 0) {   // last kernel of an outer iteration: hand its scalars to the host
            for (int i = 0; i < pub.count; ++i)
                __hip_atomic_store(pub.dst + i, pub.src + i == out ? t : pub.src[i], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub.seq_word, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    });
}

// sum(w .* x .* y) -> slot (wdot, utils.jl:165-175)
__global__ void __launch_bounds__(LSQ_NT)
k_wdot_slot(int n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ w,
            double *partials, unsigned *counter, double *out) {
    __shared__ double sh[LSQ_NT / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) acc += w[i] * x[i] * y[i];
    double b = block_sum<LSQ_NT>(acc, sh);
    grid_reduce<LSQ_NT>(b, partials, counter, gridDim.x, sh, [=](double t) { *out = t; });
}

// dx = a*p + b*q  (dogleg cases 2/3: dogleg.jl:127-129, 141-143)
__global__ void __launch_bounds__(LSQ_NT)
k_lincomb(int n, double a, const double *__restrict__ p, double b, const double *__restrict__ q,
          double *__restrict__ out) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        double v = p[i] * a;          // copyto!(dx, p); rmul!(dx, a)
        if (q) v += b * q[i];         // axpy!(b, q, dx)
        out[i] = v;
    }
}

static inline int ngrid(const lsq_ctx *c, long long n) {
    long long g = (n + LSQ_NT - 1) / LSQ_NT;
    long long cap = (long long)c->num_cus * 8;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

// sum(x^2) -> slot: tree reduction, or the reference's left-to-right order for small problems
static int sumsq_to_slot(lsq_ctx *c, bool exact, long long n, const double *x, int ctr, double *d_out,
                         LsqSlotPublish pub = LsqSlotPublish()) {
    if (exact) return lsq_seq_reduce(c, 1, (int)n, x, nullptr, nullptr, d_out);
    // few blocks: the ticket fan-in of the grid reduction (~12 ns per arrival) outweighs the loads
    long long g = (n + LSQ_NT - 1) / LSQ_NT, cap = (long long)c->num_cus * 2;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    LSQ_LAUNCH(k_sumsq_slot, dim3((int)g), dim3(LSQ_NT), 0, c->stream, n, x, c->d_partials, lsq_ctr(c, ctr), d_out,
                       pub);
    return LSQ_OK;
}
// f!(out, x) followed by sum(out.^2) -> slot.  When f! is the library's own device-side model on the sliced-row layout the
// sum rides in the residual kernel's epilogue (no second pass over the m-vector: 8 MB and a launch less per iteration at C4)
static int model_f(double *out, const double *x, void *user);
static int model_g(lsq_mat *J, const double *x, void *user);
static int model_f_sumsq(void *user, double *out, const double *x, int ctr, double *d_out, LsqSlotPublish pub, bool *done,
                         const int *skip = nullptr);
// predicted residual |J dx - f|^2 AND f!(x_trial) with its sum of squares in ONE pass over the model's matrix
// (k_sell_rows_pair, lsq_sell.h); *done tells whether it applied (else nothing was launched: the caller takes the two passes)
// sg (optional): also queue the NEXT iteration's gradient + colsumabs2 pass behind it, guarded by the acceptance test the
// kernel takes on the device (lsq_sparse_grad_colsum_spec); sg->launched tells whether that pass was queued
struct SpecGrad {
    int *gate;          // this pass's skip word (non-zero until the pair kernel clears it)
    double ssr;         // the current sum of squares (what the host will test rho against)
    double *grad;       // where the gradient goes
    bool launched;
};
static int model_pair_tail(void *user, lsq_mat *J, const double *dx, const double *fcur, double *ftrial, const double *xt,
                           double *slot_pred, double *slot_trial, LsqSlotPublish pub, const int *skip, bool *done,
                           SpecGrad *sg = nullptr);
// buffers for tanh(xt) and 1 - tanh(xt)^2 when the model's next f!(., xt) can take them from the step kernel (else nulls)
static void model_trial_buffers(void *user, const double *xt, double **t_out, double **s_out);
static int f_then_sumsq(lsq_ctx *c, bool exact, lsq_f_callback f, void *user, long long m, double *out, const double *x, int ctr,
                        double *d_out, LsqSlotPublish pub = LsqSlotPublish(), const int *skip = nullptr) {
    if (!exact && f == model_f) {
        bool fused = false;
        if (model_f_sumsq(user, out, x, ctr, d_out, pub, &fused, skip) != 0) {
            lsq_set_error("user callback reported failure");
            return LSQ_ECALLBACK;
        }
        if (fused) return LSQ_OK;
    } else if (f(out, x, user) != 0) {
        lsq_set_error("user callback reported failure");
        return LSQ_ECALLBACK;
    }
    return sumsq_to_slot(c, exact, m, out, ctr, d_out, pub);
}
// sum((J d - f)^2) -> slot (f may be null: sum((J d)^2))
static int predicted_to_slot(lsq_ctx *c, bool exact, lsq_mat *J, const double *d, const double *f, double *scratch,
                             int ctr, double *d_out, LsqSlotPublish pub = LsqSlotPublish(), const int *skip = nullptr);
static int wdot_to_slot(lsq_ctx *c, bool exact, int n, const double *x, const double *y, const double *w, int ctr,
                        double *d_out);
// g = J'f
static int gradient_into(lsq_ctx *c, bool exact, lsq_mat *J, const double *f, double *g);

struct LoopBuffers {
    lsq_ctx *c;
    int m, n;
    double *dx = nullptr, *dtd = nullptr, *xt = nullptr, *ftrial = nullptr;
    double *dgn = nullptr, *dgr = nullptr, *grad = nullptr, *fpred = nullptr;
    double *lo = nullptr, *hi = nullptr;
    // skip words of gradient passes queued before the host knows whether their step is accepted (k_sell_rows_pair writes 0 for
    // "accepted"): one int per queued pass, handed out in order; every word is non-zero (= skip) until then
    int *gate_ring = nullptr;
    int gate_next = 0;
    static constexpr int GATE_RING = 1024;
    ~LoopBuffers() {
        hipFree(gate_ring);
        hipFree(dx); hipFree(dtd); hipFree(xt); hipFree(ftrial); hipFree(dgn); hipFree(dgr); hipFree(grad); hipFree(fpred); hipFree(lo);
        hipFree(hi);
    }
};

// LeastSquaresProblemAllocated (types.jl:141-160): optimizer and solver buffers are allocated once
// and reused by later optimize! calls on the same problem -- cached in the context.
struct LsqWorkspace {
    lsq_mat *J = nullptr;
    int m = 0, n = 0, optimizer = -1, solver_kind = -1;
    bool has_lo = false, has_hi = false;
    LoopBuffers *buf = nullptr;
    lsq_solver *solver = nullptr;
};
void lsq_workspace_free(void *p) {
    LsqWorkspace *w = (LsqWorkspace *)p;
    if (!w) return;
    delete w->buf;
    if (w->solver) lsq_solver_destroy(w->solver);
    delete w;
}

static int alloc_loop(LoopBuffers &b, lsq_ctx *c, int m, int n, const lsq_options *o, bool dogleg) {
    b.c = c; b.m = m; b.n = n;
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double), mb = (size_t)(m > 0 ? m : 1) * sizeof(double);
    LSQ_HIP(hipMalloc(&b.dx, nb));
    LSQ_HIP(hipMalloc(&b.dtd, nb));
    LSQ_HIP(hipMalloc(&b.xt, nb));
    LSQ_HIP(hipMalloc(&b.ftrial, mb));
    LSQ_HIP(hipMalloc(&b.grad, nb));
    LSQ_HIP(hipMalloc(&b.fpred, mb));
    LSQ_HIP(hipMemsetAsync(b.dx, 0, nb, c->stream));
    LSQ_HIP(hipMalloc(&b.gate_ring, LoopBuffers::GATE_RING * sizeof(int)));
    LSQ_HIP(hipMemsetAsync(b.gate_ring, 1, LoopBuffers::GATE_RING * sizeof(int), c->stream));   // (every word non-zero: skip)
    b.gate_next = 0;
    if (dogleg) {
        LSQ_HIP(hipMalloc(&b.dgn, nb));
        LSQ_HIP(hipMalloc(&b.dgr, nb));
    }
    if (o->h_lower) {
        LSQ_HIP(hipMalloc(&b.lo, nb));
        LSQ_HIP(hipMemcpyAsync(b.lo, o->h_lower, nb, hipMemcpyHostToDevice, c->stream));
    }
    if (o->h_upper) {
        LSQ_HIP(hipMalloc(&b.hi, nb));
        LSQ_HIP(hipMemcpyAsync(b.hi, o->h_upper, nb, hipMemcpyHostToDevice, c->stream));
    }
    LSQ_HIP(hipStreamSynchronize(c->stream));
    return LSQ_OK;
}

static int predicted_to_slot(lsq_ctx *c, bool exact, lsq_mat *J, const double *d, const double *f, double *scratch,
                             int ctr, double *d_out, LsqSlotPublish pub, const int *skip) {
    if (exact) {
        LSQ_TRY(lsq_exact_product(J, 0, d, scratch));
        return lsq_seq_reduce(c, f ? 3 : 1, J->m, scratch, f, nullptr, d_out);
    }
    if (f) {
        EpiPredict ep{skip, 0, f, d_out, c->d_partials, lsq_ctr(c, ctr), pub};
        return launch_product(J, 0, d, ep);
    }
    EpiSumsq es{nullptr, 0, d_out, c->d_partials, lsq_ctr(c, ctr)};
    return launch_product(J, 0, d, es);
}
static int wdot_to_slot(lsq_ctx *c, bool exact, int n, const double *x, const double *y, const double *w, int ctr,
                        double *d_out) {
    if (exact) return lsq_seq_reduce(c, 2, n, x, y, w, d_out);
    int g = lsq_div_up(n > 0 ? n : 1, LSQ_NT), cap = c->num_cus * 8;
    LSQ_LAUNCH(k_wdot_slot, dim3(g > cap ? cap : g), dim3(LSQ_NT), 0, c->stream, n, x, y, w, c->d_partials,
                       lsq_ctr(c, ctr), d_out);
    return LSQ_OK;
}
static int gradient_into(lsq_ctx *c, bool exact, lsq_mat *J, const double *f, double *g) {
    (void)c;
    if (exact) return lsq_exact_product(J, 1, f, g);
    EpiGrad eg{nullptr, 0, g, nullptr, nullptr};
    return launch_product(J, 1, f, eg);
}

// utils.jl:7-31: an if/elseif chain -- at most one flag fires
static bool assess(double maxabs_dx, double maxabs_gr, double ssr, double trial_ssr, double xtol, double ftol,
                   double gtol, bool accepted, int *xc, int *fc, int *gc) {
    *xc = *fc = *gc = 0;
    if (accepted && std::fabs(trial_ssr - ssr) <= ftol * (std::fabs(ssr) + ftol)) *fc = 1;
    else if (maxabs_dx <= xtol) *xc = 1;
    else if (maxabs_gr <= gtol) *gc = 1;
    return *xc || *fc || *gc;
}

static void record(const lsq_options *o, lsq_ctx *c, int it, int n, double ssr, double g, double delta, double rho,
                   int inner, int acc, const double *d_x) {
    if (it > o->trace_cap) return;
    int k = it - 1;
    if (o->trace_ssr) o->trace_ssr[k] = ssr;
    if (o->trace_gnorm) o->trace_gnorm[k] = g;
    if (o->trace_delta) o->trace_delta[k] = delta;
    if (o->trace_rho) o->trace_rho[k] = rho;
    if (o->trace_inner) o->trace_inner[k] = inner;
    if (o->trace_accept) o->trace_accept[k] = acc;
    if (o->trace_x) {
        hipMemcpyAsync(o->trace_x + (size_t)k * n, d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        hipStreamSynchronize(c->stream);
    }
}

extern "C" void lsq_options_default(lsq_options *o) {
    memset(o, 0, sizeof(*o));
    o->x_tol = o->f_tol = o->g_tol = 1e-8;
    o->iterations = 1000;
    o->delta = -1.0;
}

static int check_start(const double *hx, int n, const lsq_options *o) {
    // levenberg_marquardt.jl:49-51 / dogleg.jl:52-54
    for (int i = 0; i < n; ++i) {
        if (o->h_lower && !(hx[i] >= o->h_lower[i])) return LSQ_EBOUNDS;
        if (o->h_upper && !(hx[i] <= o->h_upper[i])) return LSQ_EBOUNDS;
    }
    return LSQ_OK;
}

// the next skip word of the gate ring (all words are non-zero until a pair kernel clears one; a wrapped ring is re-armed in
// stream order: every pass that read the old words was queued before the memset)
static int *next_gate(lsq_ctx *c, LoopBuffers &b) {
    if (!b.gate_ring) return nullptr;
    if (b.gate_next >= LoopBuffers::GATE_RING) {
        if (hipMemsetAsync(b.gate_ring, 1, LoopBuffers::GATE_RING * sizeof(int), c->stream) != hipSuccess) return nullptr;
        b.gate_next = 0;
    }
    return b.gate_ring + b.gate_next++;
}

// One global exchange per outer iteration for sharded problems.
// Callback protocol (include/lsqhip.h, lsq_options.allreduce): vals = {ssr, maxabs_gr, converged}; converged < 0 on
// the way in announces "this rank is leaving with an error"; converged < 0 on the way out (or return code 2) says
// that some rank did -- every rank then leaves its loop with LSQ_ERCCL instead of waiting in a collective forever.
// Every rank issues exactly one exchange per outer iteration (frozen ranks at the top, active ones behind queued work), so
// the exchange count of a loop equals its iteration count on every rank: once `iterations` exchanges have been issued no
// peer will enter another collective.  (Per host thread: a context is driven by one thread at a time.)
static thread_local int t_xchg_issued = 0;          // exchanges issued by the loop running on this thread
static thread_local bool t_xchg_hook_failed = false;  // the hook itself failed: it must not be called again
static int global_exchange(const lsq_options *o, double *ssr, double *gnorm, int *all_converged) {
    if (!o->allreduce) return LSQ_OK;
    double v[3] = {*ssr, *gnorm, (double)*all_converged};
    t_xchg_issued++;
    const int rc = o->allreduce(v, 3, o->allreduce_user);
    if (rc == 2 || (rc == 0 && v[2] < -0.5)) {
        lsq_set_error("sharded run: a peer rank left its loop with an error");
        return LSQ_ERCCL;
    }
    if (rc != 0) {
        t_xchg_hook_failed = true;
        lsq_set_error("allreduce callback failed");
        return LSQ_ECALLBACK;
    }
    *ssr = v[0];
    *gnorm = v[1];
    *all_converged = v[2] > 0.5 ? 1 : 0;
    return LSQ_OK;
}

// Every exit of a sharded loop: the idle hook (which points at a stack object of the loop) is disarmed, and a rank
// that leaves with an error tells its peers so with one last exchange -- unless nobody can be there to match it: the
// failing call was the hook itself, or this rank has already issued the exchange of its LAST allowed iteration (its
// peers are in theirs, or past it, and will not enter another collective: a farewell would wait forever).
struct ShardedExit {
    lsq_ctx *c;
    const lsq_options *o;
    ShardedExit(lsq_ctx *c_, const lsq_options *o_) : c(c_), o(o_) {
        t_xchg_issued = 0;
        t_xchg_hook_failed = false;
    }
    int finish(int st) {
        c->idle_hook = nullptr;
        c->idle_user = nullptr;
        if (st != LSQ_OK && st != LSQ_ERCCL && o->allreduce && !t_xchg_hook_failed && t_xchg_issued < o->iterations) {
            double v[3] = {0.0, 0.0, -1.0};
            (void)o->allreduce(v, 3, o->allreduce_user);
        }
        return st;
    }
};

#define CB(call)                                          \
    do {                                                  \
        if ((call) != 0) {                                \
            lsq_set_error("user callback reported failure"); \
            return LSQ_ECALLBACK;                         \
        }                                                 \
    } while (0)

static int call_g(lsq_g_callback g, lsq_mat *J, const double *x, void *user) {
    J->version++;
    // (a column-scaled handle's g! says itself what it changed: lsq_mat_colscale_changed / lsq_mat_refresh)
    if (J->kind == LSQ_MAT_CSC && !J->d_cs_user) J->csr_fresh = false;
    CB(g(J, x, user));
    return lsq_ensure_csr(J);
}

// ---------------------------------------------------------------------------------------------
// levenberg_marquardt.jl:39-144
// ---------------------------------------------------------------------------------------------
// The accepted trial point becomes the iterate by swapping buffers (no copyto!); whatever buffer
// holds the iterate when the loop is left is copied back into the caller's x / fcur.
struct IterateGuard {
    lsq_ctx *c;
    double *&x, *&fcur;
    double *x_user, *fcur_user;
    int m, n;
    ~IterateGuard() {
        if (x != x_user) lsq_d2d(c, x_user, x, (size_t)n * sizeof(double));
        if (fcur != fcur_user) lsq_d2d(c, fcur_user, fcur, (size_t)m * sizeof(double));
    }
};

// The head of both loops -- f!(fcur, x0), ssr = sum(abs2, fcur), check_isfinite(x0) (levenberg_marquardt.jl:58-60,74; dogleg.jl:66-68,
// 79) -- behind ONE host hand-over instead of two (rounds 1-4: the sum and the check each waited for the device: ~40 us per
// solve at C4).  The built-in model's sum rides in its residual kernel.  Small problems keep the reference-order sums.
static int start_residual(lsq_ctx *c, lsq_mat *J, lsq_f_callback f, void *user, double *fcur, const double *x, double *ssr,
                          int *nonfinite_at) {
    const int m = J->m, n = J->n;
    if (lsq_small_mat(J) || lsq_small_vec(m) || m <= 0) {
        CB(f(fcur, x, user));
        LSQ_TRY(lsq_sumsq(c, m, fcur, ssr));
        return lsq_first_nonfinite(c, n, x, nonfinite_at);
    }
    static_assert(SL_TRIAL == SL_NONFIN + 1, "the two scalars of the loop head travel together");
    LSQ_TRY(lsq_first_nonfinite_to_slot(c, n, x, c->d_slots + SL_NONFIN));
    LSQ_TRY(f_then_sumsq(c, false, f, user, m, fcur, x, 7, c->d_slots + SL_TRIAL));
    double sl[2];
    LSQ_TRY(lsq_read_slots(c, SL_NONFIN, 2, sl));
    *nonfinite_at = (int)sl[0];
    *ssr = sl[1];
    return LSQ_OK;
}

static int optimize_lm_loop(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x_user, double *fcur_user,
                            lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r);
static int optimize_lm(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x_user, double *fcur_user,
                       lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r) {
    ShardedExit ex{c, o};
    return ex.finish(optimize_lm_loop(c, sv, b, J, x_user, fcur_user, f, g, user, o, r));
}
static int optimize_lm_loop(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x_user, double *fcur_user,
                            lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r) {
    const int m = J->m, n = J->n;
    double *x = x_user, *fcur = fcur_user, *xt = b.xt, *ftrial = b.ftrial;
    IterateGuard guard{c, x, fcur, x_user, fcur_user, m, n};
    double delta = o->delta > 0 ? o->delta : 10.0;
    double decrease_factor = 2.0;
    int f_calls = 0, g_calls = 0, mul_calls = 0, xc = 0, fc = 0, gc = 0;
    bool converged = false;
    double ssr;
    int iter = 0, nonfinite_at = -1;
    // row-sharded single problem: J, fcur, ftrial are this rank's rows; sums over residuals are completed across the ranks
    // by the hook (a device buffer, in place, ordered on the stream) before the host reads them
    const bool sharded = o->row_allreduce != nullptr;
    if (sharded) CB(f(fcur, x, user));
    else LSQ_TRY(start_residual(c, J, f, user, fcur, x, &ssr, &nonfinite_at));
    f_calls++;
    auto rows_sum = [&](double *d_buf, int count) -> int {
        if (o->row_allreduce(d_buf, count, (void *)c->stream, o->row_allreduce_user) != 0) {
            lsq_set_error("row all-reduce callback reported failure");
            return LSQ_ECALLBACK;
        }
        return LSQ_OK;
    };
    if (sharded) {
        LSQ_TRY(sumsq_to_slot(c, false, m, fcur, 7, c->d_slots + SL_TRIAL));
        LSQ_TRY(rows_sum(c->d_slots + SL_TRIAL, 1));
        LSQ_TRY(lsq_read_slots(c, SL_TRIAL, 1, &ssr));
        LSQ_TRY(lsq_first_nonfinite(c, n, x, &nonfinite_at));
    }
    r->ssr0 = ssr;
    double maxabs_gr = INFINITY;
    bool need_jac = true;
    const int gn = ngrid(c, n);
    // reference summation order for small problems (lsq_exact.hip); a general preconditioner runs the operator-level LSMR
    const bool exact = lsq_small_mat(J) && !sharded && !(sv->kind == LSQ_LSMR && sv->gen_ldiv);
    int last_inner = 0;          // inner iterations of the previous LSMR solve of this run: the guess for the next one (LsmrTail)
    // The gradient + colsumabs2 pass of the NEXT Jacobian queued behind the tail of this iteration, before the host has seen the
    // iteration's scalars (the ~12 us in which the host reads them, decides and launches used to be idle device time): the built-in
    // model on a column-scaled handle with LSMR, no exchange.  spec_grad_ready: that pass has run for the factors g! is about to
    // install -- adopted at the head of the next iteration, or forgotten if the loop is left before (SpecGuard).
    const bool no_spec_grad = getenv("LSQ_NO_SPEC_GRADIENT") != nullptr;       // (read per run: the tests flip it)
    // spec_pending: a pass has been QUEUED and the host has not yet ruled on it (an error return between the two --
    // a departed C5 peer, a failed wait, a callback failure -- must not leave another Jacobian's colsumabs2 marked current)
    bool spec_launched = false, spec_grad_ready = false, spec_pending = false;
    struct SpecGuard { lsq_mat *J; bool &ready; bool &pending; ~SpecGuard() { if (ready || pending) lsq_sparse_colsum_forget(J); } }
        spec_guard{J, spec_grad_ready, spec_pending};
    int local_done = 0;
    double gssr = ssr, ggr = maxabs_gr;
    long long inner_total = 0;
    while (iter < o->iterations) {
        if (!o->allreduce && converged) break;
        if (o->allreduce && converged) {
            // frozen ranks keep taking part in the exchange until every problem has converged
            int all = 1;
            gssr = ssr; ggr = maxabs_gr;
            LSQ_TRY(global_exchange(o, &gssr, &ggr, &all));
            if (all) break;
            ++iter; local_done = 1;
            continue;
        }
        iter++;
        if (nonfinite_at >= 0) {  // check_isfinite(x), utils.jl:70-75
            r->bad_index = nonfinite_at;
            r->iterations = iter - 1;
            lsq_set_error("IsFiniteException: non-finite x at index %d", nonfinite_at);
            return LSQ_ENONFINITE;
        }
        if (need_jac) {
            LSQ_TRY(call_g(g, J, x, user));
            g_calls++;
            need_jac = false;
        }
        // a fresh Jacobian needs colsumabs2 (:82) and J'f (:102): one pass over J gives both
        bool have_grad = false;
        if (spec_grad_ready) {      // ... and that pass has already run behind the previous iteration's tail
            lsq_sparse_grad_colsum_adopt(J);
            have_grad = true;
            spec_grad_ready = false;
        }
        // (is a pass for the NEXT Jacobian worth queueing behind this iteration's tail?  not in the last allowed iteration)
        // (independent problems with their per-iteration exchange, C5, take it too: the exchange neither reads nor decides anything
        //  the pass depends on; a rank frozen after convergence never adopts -- SpecGuard forgets the pass when the loop is left)
        // (f AND g must be the built-in model's: the pass assumes that g! installs the factors the step kernel has put into
        //  d_sspec -- lsq_model_f() is a public export and may be paired with somebody else's g!)
        const bool spec_ok = !no_spec_grad && !exact && !sharded && sv->kind == LSQ_LSMR && f == model_f && g == model_g &&
                             iter < o->iterations && !lsq_dbg_serial;
        spec_launched = false;
        if (!exact && J->colsum_version != J->version && lsq_can_fuse_grad_colsum(J)) {
            LSQ_TRY(lsq_sparse_grad_colsum(J, fcur, b.grad));
            have_grad = true;
        }
        // :82 (and reused by the LSMR preconditioner); row-sharded: colsumabs2(J) = sum_p colsumabs2(J_p), once per g!, in the
        // solver's buffer (the handle's cache keeps this rank's own block)
        const double *cs = nullptr;
        LSQ_TRY(lsq_rowshard_colsum(sv, J, &cs));
        const bool one_wg = !exact && n <= LSQ_ONE_WG_N;
        // (ssr = NaN, i.e. f(x) not finite: the fused preparation takes ssr as the norm of the right-hand side; the
        //  separate kernels re-form it and let the NaN travel to the reference's check_isfinite at the next iteration)
        const bool lm_prep = one_wg && sv->kind == LSQ_LSMR && ssr >= 0.0 && lsq_lsmr_takes_lm_prep(sv, J);
        if (exact) LSQ_TRY(lsq_exact_lm_damp(c, n, cs, 1.0 / delta, b.dtd));
        else if (!one_wg) LSQ_LAUNCH(k_lm_damp, dim3(1), dim3(1024), 0, c->stream, n, cs, 1.0 / delta, b.dtd);
        {   // :102-104 gradient g = J'f at the pre-step x.  The reference forms it AFTER the solve
            // (into dtd); J and fcur do not change in between, so it is formed once, before the
            // solve, and LSMR's setup product A'b = P.*(J'f)/beta reuses it (saves one pass over J).
            if (!have_grad) LSQ_TRY(gradient_into(c, exact, J, fcur, b.grad));
            if (sharded) LSQ_TRY(rows_sum(b.grad, n));                  // J'f = sum_p J_p'f_p
            if (lm_prep) {}   // (damping and gradient norm ride in the LSMR setup launch below)
            else if (one_wg)
                LSQ_LAUNCH(k_lm_damp_grad, dim3(1), dim3(1024), 0, c->stream, n, cs, 1.0 / delta, b.dtd, b.grad,
                                   x, b.lo, b.hi, c->d_slots + SL_GRAD);
            else
                LSQ_LAUNCH(k_gradnorm, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, b.grad, x, b.lo, b.hi,
                                   c->d_partials, lsq_ctr(c, 4), c->d_slots + SL_GRAD);
            mul_calls++;
        }
        // active ranks: the iteration's one exchange (same values as at the top of the iteration: ssr and
        // the gradient norm change only at its end) runs on the host while the device is busy -- inside
        // the LSMR driver once its look-ahead window is full, else right here behind the queued g!,
        // gradient and damping launches.  "all converged" cannot come back true: this rank itself is not.
        struct Xchg { const lsq_options *o; double *gssr, *ggr; } xchg{o, &gssr, &ggr};
        if (o->allreduce) {
            gssr = ssr; ggr = maxabs_gr;
            c->idle_hook = [](void *p) -> int {
                Xchg *q = (Xchg *)p;
                int all = 0;
                return global_exchange(q->o, q->gssr, q->ggr, &all);
            };
            c->idle_user = &xchg;
            c->idle_status = 0;
            if (sv->kind != LSQ_LSMR || exact) lsq_run_idle_hook(c);
        }
        int lmiter = 0;
        // What follows the solve -- x_trial = x - dx, the predicted residual |J dx - f|^2 (:114-117) and f!(x_trial) with its
        // sum of squares (:107, :111) -- depends on the solve only through dx.  With LSMR on the sliced layouts and a
        // device-side f! it is handed to the solve as its TAIL: enqueued right behind the inner iteration at which the
        // previous solve stopped, every kernel skipping itself if LSMR turns out not to be over by then (lsq_solver.h:
        // LsmrTail).  When the guess holds the device runs from the last inner iteration straight into the step kernel: no
        // early-exit launches of the look-ahead, no wait for the host to notice the stop.  (The predicted residual is formed
        // BEFORE f!(x_trial): it does not depend on it, and the row copy of J that the last LSMR iterations streamed is then
        // still partly in the Infinity Cache.)
        struct TailCtx {
            lsq_ctx *c; lsq_mat *J; LoopBuffers *b; lsq_f_callback f; void *user;
            const double *x, *fcur; double *xt, *ftrial;
            int m, n, gn; bool is_model; LsqSlotPublish pub; int launched;
            bool want_spec; double ssr; bool spec_launched; bool *spec_pending;
        } tc{c, J, &b, f, user, x, fcur, xt, ftrial, m, n, gn, !exact && f == model_f, LsqSlotPublish(), 0, spec_ok, ssr, false,
             &spec_pending};
        auto tail_fn = [](const int *skip, void *u) -> int {
            TailCtx &t = *(TailCtx *)u;
            lsq_ctx *c = t.c;
            double *t_out = nullptr, *s_out = nullptr;   // (the built-in model takes tanh(x_trial) from this launch)
            if (t.is_model) model_trial_buffers(t.user, t.xt, &t_out, &s_out);
            LSQ_LAUNCH(k_step, dim3(t.gn), dim3(LSQ_NT), 0, c->stream, t.n, t.x, t.b->dx, t.xt, c->d_partials,
                               lsq_ctr(c, 5), c->d_slots + SL_DX, c->d_slots + SL_NONFIN, t_out, s_out, skip);   // :106
            LSQ_HIP(hipGetLastError());
            // the last kernel of the iteration hands the scalars to the host
            t.pub = lsq_slots_ticket(c, SL_GRAD, 6);      // (the sixth: the device's own acceptance decision, k_sell_rows_pair)
            bool pair = false;
            SpecGrad sg{t.want_spec ? next_gate(c, *t.b) : nullptr, t.ssr, t.b->grad, false};
            if (t.is_model && model_pair_tail(t.user, t.J, t.b->dx, t.fcur, t.ftrial, t.xt, c->d_slots + SL_PRED, c->d_slots + SL_TRIAL,
                                              t.pub, skip, &pair, &sg) != 0) {
                lsq_set_error("user callback reported failure");
                return LSQ_ECALLBACK;
            }
            t.spec_launched = sg.launched;      // (of the LAST time the tail was queued: an earlier one skipped itself)
            if (sg.launched) *t.spec_pending = true;
            if (!pair) {
                LSQ_TRY(predicted_to_slot(c, false, t.J, t.b->dx, t.fcur, t.b->fpred, 8, c->d_slots + SL_PRED, LsqSlotPublish(), skip));
                LSQ_TRY(f_then_sumsq(c, false, t.f, t.user, t.m, t.ftrial, t.xt, 7, c->d_slots + SL_TRIAL, t.pub, skip));
            }
            LSQ_HIP(hipGetLastError());
            t.launched++;
            return LSQ_OK;
        };
        // (with the per-iteration exchange of independent problems too: the hook is issued by the LSMR driver while the device
        //  works off the queued iterations and the tail; LSQ_NO_TAIL_WITH_EXCHANGE=1 restores the round-3 exclusion)
        const bool tail_ok = !exact && !sharded && (!o->allreduce || !getenv("LSQ_NO_TAIL_WITH_EXCHANGE")) && sv->kind == LSQ_LSMR &&
                             !b.lo && !b.hi;
        bool tail_done = false;
        if (sv->kind == LSQ_LSMR) {
            const LsmrLmPrep prep{cs, 1.0 / delta, MIN_DIAGONAL, MAX_DIAGONAL, x, b.lo, b.hi, c->d_slots + SL_GRAD};
            // (speculation needs kernels that honour the skip flag: the device model on the sliced rows)
            static const bool no_spec = getenv("LSQ_NO_TAIL_SPECULATION") != nullptr;
            const bool guardable = tc.is_model && J->kind == LSQ_MAT_CSC && J->srows.active && !no_spec && !lsq_dbg_serial;
            // (diagnostic, LSQ_TAIL_ORACLE_SEQ="1,1,6,5,3,1": the inner counts of a solve that is being REPEATED, fed back as perfect
            //  first guesses -- the A/B that prices a wrong guess: profiles/r06/ab_tail_oracle.txt; never set in a measured run)
            static const std::vector<int> oracle_seq = [] {
                std::vector<int> v;
                if (const char *e = getenv("LSQ_TAIL_ORACLE_SEQ"))
                    for (const char *p = e; *p;) {
                        v.push_back(atoi(p));
                        while (*p && *p != ',') ++p;
                        if (*p == ',') ++p;
                    }
                return v;
            }();
            const int first_guess = oracle_seq.empty() ? last_inner : oracle_seq[(size_t)(iter - 1) % oracle_seq.size()];
            LsmrTail tail{guardable ? first_guess : 0, tail_fn, &tc, guardable && oracle_seq.empty()};
            LSQ_TRY(lsq_lsmr_solve(sv, J, fcur, b.dtd, b.dx, &lmiter, b.grad, ssr, lm_prep ? &prep : nullptr,
                                   tail_ok ? &tail : nullptr));  // :87
            tail_done = tail_ok;
            last_inner = lmiter / 2;
        }
        else LSQ_TRY(lsq_ldiv_damped(sv, J, fcur, b.dtd, b.dx, &lmiter));
        lsq_run_idle_hook(c);   // (a solve that never filled its window)
        if (o->allreduce && c->idle_status != LSQ_OK) return c->idle_status;
        mul_calls += lmiter;
        inner_total += lmiter / 2;
        double sl[6] = {0, 0, 0, 0, 0, 0};
        if (tail_done) {
            f_calls++;
            spec_launched = tc.spec_launched;
            LSQ_TRY(lsq_wait_slots(c, SL_GRAD, 6, tc.pub.seq, sl));
        } else {
        LSQ_TRY(lsq_box_clip(c, n, b.dx, x, b.lo, b.hi));                // :89-98
        double *t_out = nullptr, *s_out = nullptr;   // (the built-in model takes tanh(x_trial) from this launch)
        if (!exact && f == model_f) model_trial_buffers(user, xt, &t_out, &s_out);
        LSQ_LAUNCH(k_step, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, x, b.dx, xt, c->d_partials,
                           lsq_ctr(c, 5), c->d_slots + SL_DX, c->d_slots + SL_NONFIN, t_out, s_out);   // :106
        LSQ_HIP(hipGetLastError());
        if (exact) {
            CB(f(ftrial, xt, user));                                      // :107
            f_calls++;
            LSQ_TRY(sumsq_to_slot(c, exact, m, ftrial, 7, c->d_slots + SL_TRIAL));                 // :111
            LSQ_TRY(predicted_to_slot(c, exact, J, b.dx, fcur, b.fpred, 8, c->d_slots + SL_PRED));   // :114-117
            LSQ_HIP(hipGetLastError());
            LSQ_TRY(lsq_read_slots(c, SL_GRAD, 5, sl));   // the one host sync of the outer iteration
        } else if (sharded) {
            // the two sums over residual rows are completed across the ranks (one all-reduce of 2 doubles) before the
            // iteration's scalars go to the host
            static_assert(SL_PRED == SL_TRIAL + 1, "trial and predicted ssr travel together");
            LSQ_TRY(predicted_to_slot(c, exact, J, b.dx, fcur, b.fpred, 8, c->d_slots + SL_PRED));
            LSQ_TRY(f_then_sumsq(c, exact, f, user, m, ftrial, xt, 7, c->d_slots + SL_TRIAL));
            f_calls++;
            LSQ_TRY(rows_sum(c->d_slots + SL_TRIAL, 2));
            LSQ_TRY(lsq_read_slots(c, SL_GRAD, 5, sl));
        } else {
            // the predicted residual (:114-117), f!(x_trial) and sum(abs2, ftrial) (:107, :111); the last kernel of the iteration
            // hands the scalars to the host.  The built-in model on a column-scaled handle takes all of it in one pass over A.
            LsqSlotPublish pub = lsq_slots_ticket(c, SL_GRAD, 6);
            bool pair = false;
            SpecGrad sg{spec_ok ? next_gate(c, b) : nullptr, ssr, b.grad, false};
            if (f == model_f) CB(model_pair_tail(user, J, b.dx, fcur, ftrial, xt, c->d_slots + SL_PRED, c->d_slots + SL_TRIAL, pub, nullptr, &pair, &sg));
            spec_launched = sg.launched;
            if (sg.launched) spec_pending = true;
            if (!pair) {
                LSQ_TRY(predicted_to_slot(c, exact, J, b.dx, fcur, b.fpred, 8, c->d_slots + SL_PRED));
                LSQ_TRY(f_then_sumsq(c, exact, f, user, m, ftrial, xt, 7, c->d_slots + SL_TRIAL, pub));
            }
            f_calls++;
            LSQ_HIP(hipGetLastError());
            LSQ_TRY(lsq_wait_slots(c, SL_GRAD, 6, pub.seq, sl));
        }
        }
        mul_calls++;
        maxabs_gr = sl[0];
        const double maxabs_dx = sl[1];
        const int trial_nonfinite = (int)sl[2];
        const double trial_ssr = sl[3], predicted_ssr = sl[4];
        const double pred_red = std::fabs(ssr - predicted_ssr);
        const double rho = pred_red > 0 ? (ssr - trial_ssr) / pred_red : 0.0;   // :118-119
        const bool accepted = rho > MIN_STEP_QUALITY;                           // :122 (strict)
        // the gradient + colsumabs2 pass of the next Jacobian was queued behind the tail and took the same decision on the device
        // -- adopted only if the device's decision (the sixth scalar) is the host's; should they ever differ, a pass that ran for a
        // step the host refuses has overwritten the handle's colsumabs2 with another Jacobian's: forget it
        const bool dev_accepted = spec_launched && sl[5] != 0.0;
        spec_grad_ready = spec_launched && accepted && dev_accepted;
        if (spec_launched && dev_accepted && !accepted) lsq_sparse_colsum_forget(J);
        spec_pending = false;       // (ruled on: adopted at the head of the next iteration, forgotten, or never run on the device)
        converged = assess(maxabs_dx, maxabs_gr, ssr, trial_ssr, o->x_tol, o->f_tol, o->g_tol, accepted, &xc, &fc, &gc);
        if (accepted) {
            std::swap(fcur, ftrial);                                            // copyto!(fcur, ftrial)
            std::swap(x, xt);
            ssr = trial_ssr;
            double q = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
            delta = std::min(delta / std::max(1.0 / 3.0, q), MAX_DELTA);        // :130
            decrease_factor = 2.0;
            need_jac = true;
            nonfinite_at = trial_nonfinite;
        } else {
            LSQ_LAUNCH(k_revert, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, xt, b.dx, x);  // :135
            // a non-finite trial component stays non-finite through (x - dx) + dx: the reference's
            // check_isfinite(x) at the top of the next iteration then throws (utils.jl:70-75)
            nonfinite_at = trial_nonfinite;
            delta = std::max(delta / decrease_factor, MIN_DELTA);
            decrease_factor *= 2.0;
        }
        record(o, c, iter, n, ssr, maxabs_gr, delta, rho, lmiter, accepted ? 1 : 0, x);
    }
    (void)local_done;
    LSQ_HIP(hipStreamSynchronize(c->stream));
    r->optimizer = LSQ_LEVENBERG_MARQUARDT;
    r->ssr = o->allreduce ? gssr : ssr;
    r->iterations = iter;
    r->converged = converged; r->x_converged = xc; r->f_converged = fc; r->g_converged = gc;
    r->f_calls = f_calls; r->g_calls = g_calls; r->mul_calls = mul_calls;
    r->lsmr_iterations = inner_total;
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// dogleg.jl:41-203
// ---------------------------------------------------------------------------------------------
static int optimize_dogleg_loop(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x, double *fcur,
                                lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r);
static int optimize_dogleg(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x, double *fcur,
                           lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r) {
    ShardedExit ex{c, o};
    return ex.finish(optimize_dogleg_loop(c, sv, b, J, x, fcur, f, g, user, o, r));
}
static int optimize_dogleg_loop(lsq_ctx *c, lsq_solver *sv, LoopBuffers &b, lsq_mat *J, double *x_user, double *fcur_user,
                                lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *o, lsq_result *r) {
    const int m = J->m, n = J->n;
    double *x = x_user, *fcur = fcur_user, *xt = b.xt, *ftrial = b.ftrial;   // (accepted trial points become the iterate by swapping)
    IterateGuard guard{c, x, fcur, x_user, fcur_user, m, n};
    double delta = o->delta > 0 ? o->delta : 1.0;
    bool reuse = false, converged = false;
    double wnorm_dgn = 0.0, wnorm_dgr = 0.0, alpha = 0.0, wdot_gr_gn = 0.0;
    int f_calls = 0, g_calls = 0, mul_calls = 0, xc = 0, fc = 0, gc = 0;
    double ssr;
    int iter = 0, nonfinite_at = -1;
    LSQ_TRY(start_residual(c, J, f, user, fcur, x, &ssr, &nonfinite_at));
    f_calls++;
    r->ssr0 = ssr;
    double maxabs_gr = INFINITY;
    const int gn = ngrid(c, n);
    const bool exact = lsq_small_mat(J);
    double gssr = ssr, ggr = maxabs_gr;
    long long inner_total = 0;
    while (iter < o->iterations) {
        if (!o->allreduce && converged) break;
        if (o->allreduce && converged) {   // frozen rank (see optimize_lm)
            int all = 1;
            gssr = ssr; ggr = maxabs_gr;
            LSQ_TRY(global_exchange(o, &gssr, &ggr, &all));
            if (all) break;
            ++iter;
            continue;
        }
        iter++;
        if (nonfinite_at >= 0) {
            r->bad_index = nonfinite_at;
            r->iterations = iter - 1;
            lsq_set_error("IsFiniteException: non-finite x at index %d", nonfinite_at);
            return LSQ_ENONFINITE;
        }
        int ls_iter = 0;
        bool exchanged = false;
        if (!reuse) {
            LSQ_TRY(call_g(g, J, x, user));                               // :83
            g_calls++;
            bool have_grad = false;   // colsumabs2 (:85) and J'f (:99) from one pass over J
            if (!exact && J->colsum_version != J->version && lsq_can_fuse_grad_colsum(J)) {
                LSQ_TRY(lsq_sparse_grad_colsum(J, fcur, b.dgr));
                have_grad = true;
            }
            const double *cs = lsq_cached_colsum(J);                      // :85
            if (!cs) return LSQ_EHIP;
            LSQ_LAUNCH(k_dl_scale, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, cs, b.dtd);  // :90
            if (iter == 1) {                                              // :92-97
                double wx2;
                LSQ_TRY(lsq_wdot(c, n, x, x, b.dtd, &wx2));
                double wx = std::sqrt(wx2);
                if (wx > 0) delta *= wx;
            }
            if (!have_grad) LSQ_TRY(gradient_into(c, exact, J, fcur, b.dgr));  // :99
            mul_calls++;
            LSQ_LAUNCH(k_gradnorm, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, b.dgr, x, b.lo, b.hi,
                               c->d_partials, lsq_ctr(c, 4), c->d_slots + SL_GRAD);
            LSQ_TRY(lsq_ediv(c, n, b.dgr, b.dtd, b.dgr));                 // :105
            LSQ_TRY(wdot_to_slot(c, exact, n, b.dgr, b.dgr, b.dtd, 5, c->d_slots + SL_W0));     // :106
            LSQ_TRY(predicted_to_slot(c, exact, J, b.dgr, nullptr, b.fpred, 6, c->d_slots + SL_SUM));  // :109-111
            mul_calls++;
            LSQ_TRY(lsq_fill(c, n, 0.0, b.dgn));
            if (o->allreduce) {   // the iteration's exchange, hidden behind the queued g!/gradient work
                int all = 0;
                gssr = ssr; ggr = maxabs_gr;
                LSQ_TRY(global_exchange(o, &gssr, &ggr, &all));
                exchanged = true;
            }
            // :115 (LSMR: the driver itself -- what follows is ordered on the stream, the public entry's drain is not needed)
            if (sv->kind == LSQ_LSMR) LSQ_TRY(lsq_lsmr_solve(sv, J, fcur, nullptr, b.dgn, &ls_iter));
            else LSQ_TRY(lsq_ldiv(sv, J, fcur, b.dgn, &ls_iter));
            mul_calls += ls_iter;
            inner_total += ls_iter / 2;
            LSQ_TRY(wdot_to_slot(c, exact, n, b.dgn, b.dgn, b.dtd, 5, c->d_slots + SL_W1));     // :117
            LSQ_TRY(wdot_to_slot(c, exact, n, b.dgr, b.dgn, b.dtd, 5, c->d_slots + SL_W2));     // :134 (case 3)
            LSQ_HIP(hipGetLastError());
            double sl10[10];                                              // SL_GRAD .. SL_SUM in ONE hand-over to the host
            static_assert(SL_SUM - SL_GRAD == 9 && SL_W0 - SL_GRAD == 6, "slot layout");
            LSQ_TRY(lsq_read_slots(c, SL_GRAD, 10, sl10));
            maxabs_gr = sl10[0];
            const double *w = sl10 + (SL_W0 - SL_GRAD);
            wnorm_dgr = std::sqrt(w[0]);
            wnorm_dgn = std::sqrt(w[1]);
            alpha = wnorm_dgr * wnorm_dgr / w[3];
            wdot_gr_gn = w[2];  // wdot(dgr, dgn, dtd), needed only if case 3 is taken (:134)
        }
        if (o->allreduce && !exchanged) {   // iteration that reuses the Gauss-Newton step: nothing to hide behind
            int all = 0;
            gssr = ssr; ggr = maxabs_gr;
            LSQ_TRY(global_exchange(o, &gssr, &ggr, &all));
        }
        double wnorm_dx;
        if (wnorm_dgn <= delta) {                                         // :120 case 1
            LSQ_TRY(lsq_d2d(c, b.dx, b.dgn, (size_t)n * sizeof(double)));
            wnorm_dx = wnorm_dgn;
        } else if (wnorm_dgr * alpha >= delta) {                          // :124 case 2
            LSQ_LAUNCH(k_lincomb, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, delta / wnorm_dgr, b.dgr, 0.0,
                               (const double *)nullptr, b.dx);
            wnorm_dx = delta;
        } else {                                                          // :131 case 3
            double b_dot_a = alpha * wdot_gr_gn;
            double a2 = (alpha * wnorm_dgr) * (alpha * wnorm_dgr);
            double bma2 = a2 - 2 * b_dot_a + wnorm_dgn * wnorm_dgn;
            double cc = b_dot_a - a2;
            double d = std::sqrt(cc * cc + bma2 * (delta * delta - a2));
            double beta = (cc <= 0) ? (d - cc) / bma2 : (delta * delta - a2) / (d + cc);
            LSQ_LAUNCH(k_lincomb, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, beta, b.dgn, alpha * (1 - beta),
                               (const double *)b.dgr, b.dx);
            double w2;
            LSQ_TRY(lsq_wdot(c, n, b.dx, b.dx, b.dtd, &w2));              // :144
            wnorm_dx = std::sqrt(w2);
        }
        LSQ_TRY(lsq_box_clip(c, n, b.dx, x, b.lo, b.hi));                 // :148-160
        double *t_out = nullptr, *s_out = nullptr;   // (the built-in model takes tanh(x_trial) from this launch)
        if (!exact && f == model_f) model_trial_buffers(user, xt, &t_out, &s_out);
        LSQ_LAUNCH(k_step, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, x, b.dx, xt, c->d_partials,
                           lsq_ctr(c, 5), c->d_slots + SL_DX, c->d_slots + SL_NONFIN, t_out, s_out);    // :160
        LSQ_HIP(hipGetLastError());
        LSQ_TRY(f_then_sumsq(c, exact, f, user, m, ftrial, xt, 7, c->d_slots + SL_TRIAL));   // :164, :168
        f_calls++;
        LSQ_TRY(predicted_to_slot(c, exact, J, b.dx, fcur, b.fpred, 8, c->d_slots + SL_PRED));  // :171-174
        mul_calls++;
        double sl[4];
        LSQ_TRY(lsq_read_slots(c, SL_DX, 4, sl));
        const double maxabs_dx = sl[0];
        const int trial_nonfinite = (int)sl[1];
        const double trial_ssr = sl[2], predicted_ssr = sl[3];
        const double pred_red = std::fabs(ssr - predicted_ssr);
        const double rho = pred_red > 0 ? (ssr - trial_ssr) / pred_red : 0.0;
        const bool accepted = rho >= MIN_STEP_QUALITY;                    // :178 (non-strict)
        converged = assess(maxabs_dx, maxabs_gr, ssr, trial_ssr, o->x_tol, o->f_tol, o->g_tol, accepted, &xc, &fc, &gc);
        if (accepted) {
            reuse = false;
            std::swap(fcur, ftrial);                                          // copyto!(fcur, ftrial), copyto!(x, x_trial)
            std::swap(x, xt);
            ssr = trial_ssr;
            nonfinite_at = trial_nonfinite;
        } else {
            reuse = true;
            LSQ_LAUNCH(k_revert, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, xt, b.dx, x);
            nonfinite_at = trial_nonfinite;   // (x - dx) + dx keeps a non-finite component non-finite (see optimize_lm)
        }
        if (rho < DECREASE_THRESHOLD) delta = std::max(MIN_DELTA, delta * 0.5);           // :193-197
        else if (rho > INCREASE_THRESHOLD) delta = std::max(delta, 3.0 * wnorm_dx);
        record(o, c, iter, n, ssr, maxabs_gr, delta, rho, ls_iter, accepted ? 1 : 0, x);
    }
    LSQ_HIP(hipStreamSynchronize(c->stream));
    r->optimizer = LSQ_DOGLEG;
    r->ssr = o->allreduce ? gssr : ssr;
    r->iterations = iter;
    r->converged = converged; r->x_converged = xc; r->f_converged = fc; r->g_converged = gc;
    r->f_calls = f_calls; r->g_calls = g_calls; r->mul_calls = mul_calls;
    r->lsmr_iterations = inner_total;
    return LSQ_OK;
}

extern "C" int lsq_optimize(lsq_ctx *c, int optimizer, int solver_kind, lsq_mat *J, double *x, double *fcur,
                            lsq_f_callback f, lsq_g_callback g, void *user, const lsq_options *opt, lsq_result *res) {
    LSQ_RANGE("lsq_optimize");
    if (!c || !J || !x || !fcur || !f || !g || !opt || !res) {
        lsq_set_error("lsq_optimize: null argument");
        return LSQ_EARG;
    }
    memset(res, 0, sizeof(*res));
    res->bad_index = -1;
    LSQ_HIP(hipSetDevice(c->device));
    const int n = J->n;
    if (opt->h_lower || opt->h_upper) {
        std::vector<double> hx(n);
        LSQ_TRY(lsq_d2h(c, hx.data(), x, (size_t)n * sizeof(double)));
        int st = check_start(hx.data(), n, opt);
        if (st != LSQ_OK) {
            lsq_set_error("ArgumentError: Initial guess must be within bounds.");
            res->status = st;
            return st;
        }
    }
    // allocated-problem cache: same Jacobian handle, shape, optimizer and solver => reuse everything
    LsqWorkspace *w = (LsqWorkspace *)c->workspace;
    const bool lm = optimizer == LSQ_LEVENBERG_MARQUARDT;
    if (!(w && w->J == J && w->m == J->m && w->n == J->n && w->optimizer == optimizer && w->solver_kind == solver_kind &&
          w->has_lo == (opt->h_lower != nullptr) && w->has_hi == (opt->h_upper != nullptr))) {
        lsq_workspace_free(w);
        c->workspace = nullptr;
        w = new LsqWorkspace();
        int st0 = lsq_solver_create(c, J, solver_kind, lm, &w->solver);
        if (st0 != LSQ_OK) {
            delete w;
            res->status = st0;
            return st0;
        }
        w->buf = new LoopBuffers();
        st0 = alloc_loop(*w->buf, c, J->m, J->n, opt, !lm);
        if (st0 != LSQ_OK) {
            lsq_workspace_free(w);
            res->status = st0;
            return st0;
        }
        w->J = J; w->m = J->m; w->n = J->n; w->optimizer = optimizer; w->solver_kind = solver_kind;
        w->has_lo = opt->h_lower != nullptr; w->has_hi = opt->h_upper != nullptr;
        c->workspace = w;
    } else {
        const size_t nb = (size_t)(J->n > 0 ? J->n : 1) * sizeof(double);
        LSQ_HIP(hipMemsetAsync(w->buf->dx, 0, nb, c->stream));
        if (opt->h_lower) LSQ_HIP(hipMemcpyAsync(w->buf->lo, opt->h_lower, nb, hipMemcpyHostToDevice, c->stream));
        if (opt->h_upper) LSQ_HIP(hipMemcpyAsync(w->buf->hi, opt->h_upper, nb, hipMemcpyHostToDevice, c->stream));
        if (opt->h_lower || opt->h_upper) LSQ_HIP(hipStreamSynchronize(c->stream));
    }
    if (w->solver->kind == LSQ_LSMR) {   // LSMR(preconditioner!, P): per call, the cached solver may have had another one
        w->solver->precond_cb = opt->preconditioner;
        w->solver->precond_user = opt->preconditioner_user;
        w->solver->gen_update = opt->precond_update;
        w->solver->gen_ldiv = opt->precond_ldiv;
        w->solver->gen_user = opt->precond_general_user;
    }
    if (opt->row_allreduce) {   // row-sharded single problem
        if (!lm || w->solver->kind != LSQ_LSMR || opt->allreduce || J->kind == LSQ_MAT_OP) {
            lsq_set_error("row-sharded runs (lsq_options.row_allreduce): LevenbergMarquardt(LSMR()) on stored Jacobians only, and "
                          "not together with the independent-problems exchange (lsq_options.allreduce)");
            res->status = LSQ_EARG;
            return LSQ_EARG;
        }
        LSQ_TRY(lsq_solver_set_row_allreduce(w->solver, opt->row_allreduce, opt->row_allreduce_user,
                                             opt->global_rows > 0 ? opt->global_rows : (long long)J->m));
    } else if (w->solver->kind == LSQ_LSMR) {
        LSQ_TRY(lsq_solver_set_row_allreduce(w->solver, nullptr, nullptr, 0));
    }
    auto t0 = std::chrono::steady_clock::now();
    int st = lm ? optimize_lm(c, w->solver, *w->buf, J, x, fcur, f, g, user, opt, res)
                : optimize_dogleg(c, w->solver, *w->buf, J, x, fcur, f, g, user, opt, res);
    hipStreamSynchronize(c->stream);
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res->status = st;
    return st;
}

// ---------------------------------------------------------------------------------------------
// built-in model: r(x) = A tanh(x) - b ; J = A diag(1 - tanh(x)^2)      
// ---------------------------------------------------------------------------------------------
struct lsq_model {
    lsq_ctx *ctx;
    lsq_mat *J;
    // fused: J is a COLUMN-SCALED handle on the sliced layouts (lsq_mat_set_colscale): J's own storage holds A once, g! writes
    // the n factors s = 1 - tanh(x)^2 and nothing else; every product applies them on the fly
    bool fused = false;
    double *d_Acsc = nullptr;  // A values, CSC order (or dense column-major)                       } not fused: J's values are
    double *d_Acsr = nullptr;  // A values in J's row-mirror layout (CSR order or sliced rows)      } multiplied out by g!
    double *d_Ab = nullptr;    // A values in J's column-mirror layout (window-blocked CSC or sliced columns)
    double *d_b = nullptr;
    double *d_t = nullptr;     // tanh(x) of the latest f!
    double *d_s = nullptr;     // 1 - tanh(x)^2 of the latest g! (fused: this IS J's column scale)
    double *d_sspec = nullptr; // the same at the latest trial point (written by the step kernel)
    const double *tanh_x = nullptr;   // device vector whose tanh the step kernel has already put into d_t
    const double *sfac_x = nullptr;   // device vector whose 1 - tanh^2 the step kernel has already put into d_sspec
    // k_sell_rows_pair reads its two m-vectors in the SLICE ORDER of J's sliced rows (lsq_sell.h): b once, and two mirrors of
    // residual vectors, each valid for the row-order device vector named in perm_of (nullptr: stale)
    double *d_b_perm = nullptr;
    double *d_perm[2] = {nullptr, nullptr};
    const double *perm_of[2] = {nullptr, nullptr};
};
static void model_perm_invalidate(lsq_model *md, const double *vec) {   // vec == nullptr: all
    for (int k = 0; k < 2; ++k)
        if (!vec || md->perm_of[k] == vec) md->perm_of[k] = nullptr;
}

__global__ void __launch_bounds__(LSQ_NT) k_tanh(int n, const double *__restrict__ x, double *__restrict__ t) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) t[i] = tanh(x[i]);
}

struct EpiResidual {  // out = A t - b
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    const double *b;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int i, double dot, double &) const { out[i] = dot - b[i]; }
    using has_pre = void;
    __device__ double pre(int i) const { return b[i]; }
    __device__ void seg_pre(int i, double dot, double bi, double &) const { out[i] = dot - bi; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

struct EpiResidualSq {  // out = A t - b, and sum(out.^2) -> slot (+ the iteration's scalars to the host, like EpiPredict)
    static constexpr bool REDUCE = true;
    const int *done;
    int extra_blocks;
    const double *b;
    double *out;
    double *partials;
    unsigned *counter;
    double *slot;
    LsqSlotPublish pub;
    __device__ void seg(int i, double dot, double &racc) const { seg_pre(i, dot, b[i], racc); }
    using has_pre = void;
    __device__ double pre(int i) const { return b[i]; }
    __device__ void seg_pre(int i, double dot, double bi, double &racc) const {
        const double v = dot - bi;
        out[i] = v;
        racc += v * v;
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double t) const {
        *slot = t;
        if (pub.count > 0) {
            for (int i = 0; i < pub.count; ++i)
                __hip_atomic_store(pub.dst + i, pub.src + i == slot ? t : pub.src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pub.seq_word, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

// column scaling of a column-segmented value array (CSC nzval or dense columns)
__global__ void __launch_bounds__(LSQ_NT)
k_scale_cols(int n, const int *__restrict__ colptr, int m_dense, const double *__restrict__ A,
             const double *__restrict__ x, double *__restrict__ out) {
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        double t = tanh(x[j]);
        double s = 1.0 - t * t;
        long long k0 = colptr ? colptr[j] : (long long)j * m_dense;
        long long k1 = colptr ? colptr[j + 1] : (long long)(j + 1) * m_dense;
        for (long long k = k0 + threadIdx.x; k < k1; k += LSQ_NT) out[k] = A[k] * s;
    }
}
// dense J = A diag(1 - tanh(x)^2): block (column j, row chunk)
__global__ void __launch_bounds__(LSQ_NT)
k_scale_dense(int m, const double *__restrict__ A, const double *__restrict__ x, double *__restrict__ out) {
    const int j = blockIdx.x;
    const double t = tanh(x[j]);
    const double sfac = 1.0 - t * t;
    const size_t base = (size_t)j * m;
    for (int i = blockIdx.y * LSQ_NT + threadIdx.x; i < m; i += gridDim.y * LSQ_NT) out[base + i] = A[base + i] * sfac;
}
// the CSR mirror: scale by the column of each entry
__global__ void __launch_bounds__(LSQ_NT)
k_scale_csr(long long nnz, const int *__restrict__ colidx, const double *__restrict__ A, const double *__restrict__ sfac,
            double *__restrict__ out) {
    for (long long k = blockIdx.x * (long long)LSQ_NT + threadIdx.x; k < nnz; k += (long long)gridDim.x * LSQ_NT)
        out[k] = A[k] * sfac[colidx[k]];
}
__global__ void __launch_bounds__(LSQ_NT) k_tanh_sfac(int n, const double *__restrict__ x, double *__restrict__ t,
                                                      double *__restrict__ s) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        const double v = tanh(x[i]);
        t[i] = v;
        s[i] = 1.0 - v * v;
    }
}
__global__ void __launch_bounds__(LSQ_NT) k_sfac(int n, const double *__restrict__ x, double *__restrict__ s) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        double t = tanh(x[i]);
        s[i] = 1.0 - t * t;
    }
}

// window-blocked CSC: segment s belongs to column s % n; one wave per segment
__global__ void __launch_bounds__(LSQ_NT)
k_scale_bcsc(int nseg, int n, const int *__restrict__ ptr, const double *__restrict__ A, const double *__restrict__ sfac,
             double *__restrict__ out) {
    const int lane = threadIdx.x & 63, per = LSQ_NT / 64;
    for (int s = blockIdx.x * per + (threadIdx.x >> 6); s < nseg; s += gridDim.x * per) {
        const double f = sfac[s % n];
        const int k1 = ptr[s + 1];
        for (int k = ptr[s] + lane; k < k1; k += 64) out[k] = A[k] * f;
    }
}

// out[k] = A[k] * sfac[col16[k]] with the n scale factors staged in LDS: pure streaming
// (16-byte value loads/stores, 8-byte index loads), one persistent 1024-thread workgroup per CU.
// NT: the output is stored non-temporally (16-byte vector stores: 4.8 TB/s instead of 3.8 -- no
// write-allocate next to the A stream); used for the copy that is not read back right away.
typedef double lsq_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ void __launch_bounds__(1024)
k_scale_lds(long long nnz4, const unsigned short *__restrict__ col16, const double *__restrict__ A,
            const double *__restrict__ sfac, int n, double *__restrict__ out) {
    extern __shared__ double sf[];
    // (the factors 1 - tanh(x)^2 come from k_sfac: ten fp64 tanh per thread in every one of the 256
    //  workgroups cost ~6 us before the first byte was streamed)
    for (int i = threadIdx.x; i < n; i += 1024) sf[i] = sfac[i];
    __syncthreads();
    for (long long q = blockIdx.x * 1024LL + threadIdx.x; q < nnz4; q += (long long)gridDim.x * 1024) {
        const long long k = 4 * q;  // arrays are padded to a multiple of 4 (+8)
        const lsq_d2 a0 = *reinterpret_cast<const lsq_d2 *>(A + k);
        const lsq_d2 a1 = *reinterpret_cast<const lsq_d2 *>(A + k + 2);
        const uint2 c = *reinterpret_cast<const uint2 *>(col16 + k);
        lsq_d2 o0, o1;
        o0.x = a0.x * sf[c.x & 0xffffu];
        o0.y = a0.y * sf[c.x >> 16];
        o1.x = a1.x * sf[c.y & 0xffffu];
        o1.y = a1.y * sf[c.y >> 16];
        if constexpr (NT) {
            __builtin_nontemporal_store(o0, reinterpret_cast<lsq_d2 *>(out + k));
            __builtin_nontemporal_store(o1, reinterpret_cast<lsq_d2 *>(out + k + 2));
        } else {
            *reinterpret_cast<lsq_d2 *>(out + k) = o0;
            *reinterpret_cast<lsq_d2 *>(out + k + 2) = o1;
        }
    }
}

// short segments (LDS-window plan: ~4 entries): one thread per segment
__global__ void __launch_bounds__(LSQ_NT)
k_scale_bcsc_thread(int nseg, int n, const int *__restrict__ ptr, const double *__restrict__ A,
                    const double *__restrict__ sfac, double *__restrict__ out) {
    for (int s = blockIdx.x * LSQ_NT + threadIdx.x; s < nseg; s += gridDim.x * LSQ_NT) {
        const double f = sfac[s % n];
        const int k1 = ptr[s + 1];
        for (int k = ptr[s]; k < k1; ++k) out[k] = A[k] * f;
    }
}

static int model_f(double *out, const double *x, void *user) {
    lsq_model *md = (lsq_model *)user;
    lsq_ctx *c = md->ctx;
    lsq_mat *J = md->J;
    const bool have_tanh = md->tanh_x == x;   // the step kernel formed tanh(x) while it wrote x (model_trial_buffers)
    md->tanh_x = nullptr;
    model_perm_invalidate(md, out);           // (a slice-order mirror of `out` is stale from here on)
    if (!have_tanh) {
        md->sfac_x = nullptr;
        LSQ_LAUNCH(k_tanh, dim3(ngrid(c, J->n)), dim3(LSQ_NT), 0, c->stream, J->n, x, md->d_t);
    }
    EpiResidual e{nullptr, 0, md->d_b, out, nullptr, nullptr};
    if (J->kind == LSQ_MAT_CSC && J->srows.active) {
        // the stored values themselves (fused: J's storage IS A; else A in the same layout), no column scale: r = A t - b
        if (launch_sell_rows(J, nullptr, md->d_t, e, md->fused ? nullptr : md->d_Acsr) != LSQ_OK) return 1;
    } else if (J->kind == LSQ_MAT_CSC) {
        LsqSegs A = J->csr;  // same pattern, A's values
        A.d_val = md->d_Acsr;
        if (launch_segs<false>(c, A, md->d_t, e) != LSQ_OK) return 1;
    } else {
        lsq_mat tmp = *J;
        tmp.d_dense = md->d_Acsc;
        if (launch_product(&tmp, 0, md->d_t, e) != LSQ_OK) return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the step kernel of both loops writes tanh(x_trial) and 1 - tanh(x_trial)^2 next to x_trial itself: the model's f!(., x_trial)
// and -- if the step is accepted -- its g!(., x_trial) then need no n-length launch of their own
static void model_trial_buffers(void *user, const double *xt, double **t_out, double **s_out) {
    lsq_model *md = (lsq_model *)user;
    md->tanh_x = md->sfac_x = nullptr;
    if ((md->J->kind == LSQ_MAT_CSC && md->J->srows.active) || md->J->kind == LSQ_MAT_DENSE) {
        *t_out = md->d_t;
        *s_out = md->d_sspec;
        md->tanh_x = md->sfac_x = xt;
    }
}

// f! + sum(abs2, out) in one pass (sliced rows only; *done tells whether it applied)
static int model_f_sumsq(void *user, double *out, const double *x, int ctr, double *d_out, LsqSlotPublish pub, bool *done,
                         const int *skip) {
    lsq_model *md = (lsq_model *)user;
    lsq_ctx *c = md->ctx;
    lsq_mat *J = md->J;
    *done = false;
    if (!(J->kind == LSQ_MAT_CSC && J->srows.active)) return model_f(out, x, user);
    model_perm_invalidate(md, out);
    EpiResidualSq e{skip, 0, md->d_b, out, c->d_partials, lsq_ctr(c, ctr), d_out, pub};
    const bool have_tanh = md->tanh_x == x;   // k_step formed tanh(x) while it wrote x
    md->tanh_x = nullptr;
    if (!have_tanh) {
        md->sfac_x = nullptr;
        LSQ_LAUNCH(k_tanh, dim3(ngrid(c, J->n)), dim3(LSQ_NT), 0, c->stream, J->n, x, md->d_t);
    }
    if (launch_sell_rows(J, nullptr, md->d_t, e, md->fused ? nullptr : md->d_Acsr) != LSQ_OK) return 1;
    *done = true;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static int model_pair_tail(void *user, lsq_mat *J, const double *dx, const double *fcur, double *ftrial, const double *xt,
                           double *slot_pred, double *slot_trial, LsqSlotPublish pub, const int *skip, bool *done, SpecGrad *sg) {
    lsq_model *md = (lsq_model *)user;
    lsq_ctx *c = md->ctx;
    *done = false;
    // the column-scaled handle on the one-window sliced rows (J = A diag(s): both products stream the same A), both gather
    // vectors resident in LDS, tanh(x_trial) already formed by the step kernel
    if (!(md->fused && J == md->J && J->kind == LSQ_MAT_CSC && J->srows.active && J->srows.ncw == 1 && J->d_colscale &&
          J->n <= LSQ_PAIR_X_MAX && md->tanh_x == xt) || getenv("LSQ_NO_PAIR_TAIL"))
        return 0;
    const LsqSell &S = J->srows;
    const int nxpad = (J->n + 1) & ~1;
    // (two gather vectors; after a block's stream their LDS is the output window of its LSQ_SELL_ROWS_MAX rows)
    const size_t lds = (size_t)std::max(2 * nxpad, LSQ_SELL_ROWS_MAX) * sizeof(double);
    if (lsq_set_lds(c, (const void *)k_sell_rows_pair<0>, (size_t)2 * LSQ_PAIR_X_MAX * sizeof(double)) != LSQ_OK) return 0;
    // slice-order mirrors: b once; f from the slot that mirrors fcur (filled by an earlier launch of this kernel if the step that
    // produced fcur was accepted, else permuted now); the other slot takes this launch's trial residual
    const size_t plen = (size_t)S.nslices * 64;
    const int pgrid = std::max(1, std::min(lsq_div_up((long long)plen, LSQ_NT), c->num_cus * 8));
    if (!md->d_b_perm) {
        double *pb = nullptr, *p0 = nullptr, *p1 = nullptr;     // (all three or none: a partial set must not look initialised)
        if (hipMalloc(&pb, plen * sizeof(double)) != hipSuccess || hipMalloc(&p0, plen * sizeof(double)) != hipSuccess ||
            hipMalloc(&p1, plen * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            hipFree(pb); hipFree(p0); hipFree(p1);
            return 0;                                           // (the caller takes the two-pass tail)
        }
        md->d_b_perm = pb; md->d_perm[0] = p0; md->d_perm[1] = p1;
        LSQ_LAUNCH(k_sell_perm_rows<0>, dim3(pgrid), dim3(LSQ_NT), 0, c->stream, sell_dev(S), S.wrows, S.nslices, (const double *)md->d_b,
                   md->d_b_perm);
    }
    int k = md->perm_of[0] == fcur ? 0 : (md->perm_of[1] == fcur ? 1 : -1);
    if (k < 0) {
        k = 0;
        LSQ_LAUNCH(k_sell_perm_rows<0>, dim3(pgrid), dim3(LSQ_NT), 0, c->stream, sell_dev(S), S.wrows, S.nslices, fcur, md->d_perm[k]);
        md->perm_of[k] = fcur;
    }
    md->perm_of[1 - k] = ftrial;
    md->tanh_x = nullptr;     // (consumed, as model_f_sumsq does)
    const int grid = std::max(1, std::min(S.nblocks, c->num_cus));
    // (the step kernel has put 1 - tanh^2(x_trial) into d_sspec: the factors g! will install if the step is accepted)
    const bool spec = sg && sg->gate && md->sfac_x == xt;
    SellPairEpi e{skip, md->d_perm[k], md->d_b_perm, ftrial, md->d_perm[1 - k], c->d_partials, c->d_partials + 4096, lsq_ctr(c, 7),
                  slot_pred, slot_trial, pub, spec ? sg->gate : nullptr, spec ? sg->ssr : 0.0, MIN_STEP_QUALITY,
                  c->d_slots + SL_SSR};
    LSQ_LAUNCH(k_sell_rows_pair<0>, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, sell_dev(S), S.wrows, J->m, dx, J->d_colscale,
               (const double *)md->d_t, J->n, nxpad, e);
    *done = true;
    if (hipGetLastError() != hipSuccess) return 1;
    if (spec) {
        const int st = lsq_sparse_grad_colsum_spec(J, ftrial, sg->grad, md->d_sspec, sg->gate);
        if (st == LSQ_OK) sg->launched = true;
        else if (st != LSQ_EARG) return 1;
    }
    return 0;
}

static int model_g(lsq_mat *J, const double *x, void *user) {
    lsq_model *md = (lsq_model *)user;
    lsq_ctx *c = md->ctx;
    // s = 1 - tanh(x)^2: already there if x is the trial point the step kernel has just written (dense columns: k_scale_dense
    // forms the factor of its column from x itself)
    if (J->kind == LSQ_MAT_DENSE) {}
    else if (md->sfac_x == x) std::swap(md->d_s, md->d_sspec);
    else LSQ_LAUNCH(k_sfac, dim3(ngrid(c, J->n)), dim3(LSQ_NT), 0, c->stream, J->n, x, md->d_s);
    md->sfac_x = nullptr;
    md->tanh_x = nullptr;
    if (md->fused) {
        // J = A diag(s) is never multiplied out: handing the handle its (new) factor vector is all of g!
        if (lsq_mat_set_colscale(J, md->d_s) != LSQ_OK) return 1;
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    if (J->kind == LSQ_MAT_CSC) {
        // mirrors the products read: rows (CSR or sliced rows) and columns (window-blocked CSC or sliced
        // columns), each with a 16-bit column per stored entry when the LDS-staged scaling applies
        const bool have_cols = J->scols.active || J->nwin > 1;
        const unsigned short *rcol = J->srows.active ? J->srows.d_idx16 : J->csr.d_idx16;
        const unsigned short *ccol = J->scols.active ? J->scols.d_col16 : J->bcsc.d_col16;
        double *rval = J->srows.active ? J->srows.d_val : J->csr.d_val;
        double *cval = J->scols.active ? J->scols.d_val : J->bcsc.d_val;
        const long long rlen = lsq_mirror_rows_len(J), clen = lsq_mirror_cols_len(J);
        const bool lds_ok = J->n <= 12000 && J->nnz >= (1 << 20) && J->srows.ncw == 1;   // (window-relative indices otherwise)
        // big problems: every product (and colsumabs2) reads the mirrors, so only those are written; the
        // CSC-ordered copy is rebuilt on demand (lsq_ensure_csc)
        const bool lazy_csc = lds_ok && rcol && have_cols && ccol && lsq_can_fuse_grad_colsum(J);
        if (!lazy_csc) {
            int grid = J->n < c->num_cus * 16 ? (J->n > 0 ? J->n : 1) : c->num_cus * 16;
            LSQ_LAUNCH(k_scale_cols, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->n, J->csc.d_ptr, 0, md->d_Acsc,
                               x, J->csc.d_val);
        }
        J->csc_fresh = !lazy_csc;
        const size_t lds = (size_t)J->n * sizeof(double);
        if (lds_ok) {
            LSQ_TRY(lsq_set_lds(c, (const void *)k_scale_lds<true>, 12000 * 8));
            LSQ_TRY(lsq_set_lds(c, (const void *)k_scale_lds<false>, 12000 * 8));
        }
        if (J->nnz > 0) {
            if (lds_ok && rcol) {
                LSQ_LAUNCH(k_scale_lds<true>, dim3(c->num_cus), dim3(1024), lds, c->stream, (rlen + 3) / 4, rcol,
                                   md->d_Acsr, md->d_s, J->n, rval);
            } else if (!J->srows.active) {
                long long g2 = std::min<long long>((J->nnz + LSQ_NT - 1) / LSQ_NT, (long long)c->num_cus * 16);
                LSQ_LAUNCH(k_scale_csr, dim3((int)g2), dim3(LSQ_NT), 0, c->stream, J->nnz, J->csr.d_idx,
                                   md->d_Acsr, md->d_s, J->csr.d_val);
            } else {
                // sliced rows without the LDS-staged scaling (forced onto a small pattern): rebuild from the CSC copy
                if (lazy_csc) return 1;
                if (lsq_mirror_rows(J, J->csc.d_val, rval) != LSQ_OK) return 1;
            }
        }
        if (have_cols) {
            if (lds_ok && ccol) {
                LSQ_LAUNCH(k_scale_lds<false>, dim3(c->num_cus), dim3(1024), lds, c->stream, (clen + 3) / 4,
                                   ccol, md->d_Ab, md->d_s, J->n, cval);
            } else if (J->scols.active) {
                // no LDS-staged scaling: rebuild the sliced columns from the CSC copy instead
                if (lazy_csc) return 1;
                if (lsq_mirror_cols(J, J->csc.d_val, cval) != LSQ_OK) return 1;
            } else if (J->nnz < 16LL * J->bcsc.nseg) {
                int nsegs = J->bcsc.nseg;
                int g3 = std::min(lsq_div_up(nsegs, LSQ_NT), c->num_cus * 16);
                LSQ_LAUNCH(k_scale_bcsc_thread, dim3(g3), dim3(LSQ_NT), 0, c->stream, nsegs, J->n,
                                   J->bcsc.d_ptr, md->d_Ab, md->d_s, J->bcsc.d_val);
            } else {
                int nsegs = J->bcsc.nseg;
                int g3 = std::min(lsq_div_up(nsegs, LSQ_NT / 64), c->num_cus * 16);
                LSQ_LAUNCH(k_scale_bcsc, dim3(g3), dim3(LSQ_NT), 0, c->stream, nsegs, J->n, J->bcsc.d_ptr,
                                   md->d_Ab, md->d_s, J->bcsc.d_val);
            }
        }
        J->csr_fresh = true;  // every mirror written directly: no permutation pass needed
    } else {
        // dense columns: every block takes a (column, row chunk) pair, so that a matrix with few columns still fills the chip
        const long long tot = (long long)J->m * J->n;
        const int chunks = std::max(1, (int)std::min<long long>((J->m + 4 * LSQ_NT - 1) / (4 * LSQ_NT),
                                                               std::max<long long>(1, (long long)c->num_cus * 16 / std::max(1, J->n))));
        if (tot > 0)
            LSQ_LAUNCH(k_scale_dense, dim3(J->n, chunks), dim3(LSQ_NT), 0, c->stream, J->m, md->d_Acsc, x, J->d_dense);
    }
    J->version++;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" lsq_f_callback lsq_model_f(void) { return model_f; }
extern "C" lsq_g_callback lsq_model_g(void) { return model_g; }

extern "C" int lsq_model_tanh_create(lsq_ctx *c, lsq_mat *J, const double *hA, const double *hb, lsq_model **out) {
    if (!c || !J || !hA || !hb || !out) return LSQ_EARG;
    LSQ_HIP(hipSetDevice(c->device));
    lsq_model *md = new lsq_model();
    md->ctx = c;
    md->J = J;
    const size_t nb = (size_t)(J->n > 0 ? J->n : 1) * sizeof(double);
    LSQ_HIP(hipMalloc(&md->d_b, (size_t)(J->m > 0 ? J->m : 1) * sizeof(double)));
    LSQ_HIP(hipMemcpy(md->d_b, hb, (size_t)J->m * sizeof(double), hipMemcpyHostToDevice));
    LSQ_HIP(hipMalloc(&md->d_t, nb));
    LSQ_HIP(hipMalloc(&md->d_s, nb));
    LSQ_HIP(hipMalloc(&md->d_sspec, nb));
    if (lsq_colscale_fusable(J)) {
        // J's own storage takes A (once, in every layout); from here on J = A diag(s) through the handle's column scale
        LSQ_TRY(lsq_mat_set_values(J, hA));
        LSQ_TRY(lsq_fill(c, J->n, 1.0, md->d_s));
        LSQ_TRY(lsq_mat_set_colscale(J, md->d_s));
        md->fused = true;
        LSQ_HIP(hipStreamSynchronize(c->stream));
        *out = md;
        return LSQ_OK;
    }
    size_t vb = (size_t)(J->nnz + 8) * sizeof(double);
    LSQ_HIP(hipMalloc(&md->d_Acsc, vb));
    LSQ_ZERO(md->d_Acsc, 0, vb);
    LSQ_HIP(hipMemcpy(md->d_Acsc, hA, (size_t)J->nnz * sizeof(double), hipMemcpyHostToDevice));
    if (J->kind == LSQ_MAT_CSC) {
        // A in the layouts the products of J read (same maps as J's own mirrors), permuted once
        const size_t rb = (size_t)(lsq_mirror_rows_len(J) + 1024) * sizeof(double);
        LSQ_HIP(hipMalloc(&md->d_Acsr, rb));
        LSQ_ZERO(md->d_Acsr, 0, rb);
        LSQ_TRY(lsq_mirror_rows(J, md->d_Acsc, md->d_Acsr));
        if (lsq_mirror_cols_len(J) > 0) {
            const size_t cbytes = (size_t)(lsq_mirror_cols_len(J) + 1024) * sizeof(double);
            LSQ_HIP(hipMalloc(&md->d_Ab, cbytes));
            LSQ_ZERO(md->d_Ab, 0, cbytes);
            LSQ_TRY(lsq_mirror_cols(J, md->d_Acsc, md->d_Ab));
        }
        LSQ_HIP(hipStreamSynchronize(c->stream));
    }
    *out = md;
    return LSQ_OK;
}

extern "C" int lsq_model_destroy(lsq_model *md) {
    if (!md) return LSQ_OK;
    hipStreamSynchronize(md->ctx->stream);
    if (md->fused && md->J) lsq_mat_set_colscale(md->J, nullptr);   // (the factor vector goes away with the model)
    hipFree(md->d_Acsc); hipFree(md->d_Acsr); hipFree(md->d_Ab); hipFree(md->d_b); hipFree(md->d_t); hipFree(md->d_s); hipFree(md->d_sspec);
    hipFree(md->d_b_perm); hipFree(md->d_perm[0]); hipFree(md->d_perm[1]);
    delete md;
    return LSQ_OK;
}
