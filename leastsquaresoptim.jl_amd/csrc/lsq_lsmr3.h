// Round 5: the LSMR inner iteration as THREE launches (included by lsq_lsmr.hip behind k_lsmr_update).
//
// K3 (k_lsmr_update: 8 us of latencies for 0.5 MB -- a launch ramp, a round of loads, the scalar chain, a ticket round, a
// system-scope publish) is folded into the head of the NEXT J*v launch:
//   * every workgroup forms alpha, beta and the rotations of the iteration that has just finished from the deferred partials
//     (lsmr.jl:119-149, as K3's workgroups did) and takes the STOP DECISION (lsmr.jl:205-231) itself;
//   * the J*v workgroups stage the gather vector themselves and UNNORMALISED, w = (P.*s).*v~ (consumer-side: they read v~ and P
//     instead of t), so their stream starts at once; J (P.*v) = (J w)/alpha is finished in the epilogue, by which time wave 0 of the
//     workgroup has worked through the scalar chain (the other waves draw its share of the slices from a counter) -- like u, v
//     is never normalised on the way into a product (lsmr.jl:118,124).  If the finished iteration was the last they stop drawing;
//   * the n-vector updates of lsmr.jl:152-156 (v, hbar, x, h; the caller's P.*x; the damped rows u~x of iterative_lsmr.jl:92)
//     run beside the product in a few extra workgroups of the same launch (the sliced rows of C4 leave 3 of the 256 CUs
//     without a block), which also commit the state and publish the progress word -- at the HEAD of the launch.
// The stop decision needs ||x|| of the iterate those extra workgroups are only just writing (it enters rule 1 through
// rtol = btol + atol ||A|| ||x|| / ||b|| and rule 4 through t1).  Every workgroup BOUNDS it instead,
//     | ||x_k|| - ||x_(k-1)|| |  <=  |c2| (|c1| ||hbar_(k-1)|| + ||h_(k-1)||)        (x_k = x_(k-1) + c2 (c1 hbar_(k-1) + h_(k-1))),
// from three norms the previous launch's update workgroups left as partials, and evaluates the rules on the interval: the
// outcome is the reference's whenever it does not depend on where in the interval ||x_k|| lies.  When it does (test1 within
// ~1e-6 of btol), every workgroup computes sum(x_k^2) itself from x, hbar, h -- the same routine, hence the same bits, in every
// workgroup -- and evaluates the rules exactly as lsmr_commit does.  LSQ_LSMR_EXACT_NORMX=1 forces that path (tests).
// The state is double-buffered (launch k reads st[(k-1)&1] and writes st[k&1]): a workgroup that is dispatched late must not
// read the state its own launch commits; the same holds for the partials a launch both reads (head) and writes (tail).
#pragma once

constexpr int LSQ_FUSED_UB_MAX = 4;
struct LsmrFused {
    const LsmrState *st_in;
    LsmrState *st_out;
    LsqMailbox *mail;
    const double *pu_in; const int *npu_in;      // sum(u~_y^2): the previous launch's product workgroups (or the setup)
    double *pu_out; int *npu_out;
    const double *px_in; const int *npx_in;      // sum(u~_x^2): the previous launch's update workgroups (null: u~_x == 0)
    double *px_out; int *npx_out;
    const double *pv; const int *npv;            // sum(v~^2): the previous K2 (or the setup)
    const double *pn_in;                         // [3][LSQ_FUSED_UB_MAX]: sum(x^2), sum(hbar^2), sum(h^2) left by the previous launch
    double *pn_out;
    const double *vt;                            // v~ (n)
    const double *P, *cs, *dg;                   // preconditioner (or null), column scale of J (or null), sqrt(damp) (or null)
    // x, hbar, h are double-buffered like the state: the exact-norm path lets EVERY workgroup read the old vectors while the
    // update workgroups of the same launch write the new ones
    const double *h_in, *hbar_in, *x_in;
    double *h_out, *hbar_out, *x_out;
    double *v, *xout, *ux;
    const double *uold; double *unew;            // m
    int n, ub, force_exact;
};

// the rules of lsmr.jl:224-231 with ||x|| known only to lie in [nx_lo, nx_hi]; true: decided, state advanced like lsmr_commit
__device__ inline bool lsmr_decide_bounded(LsmrState &s, double nx_lo, double nx_hi) {
    const int iter = s.iter + 1;
    const double test1 = s.normr / s.normb;
    const double test2 = s.normAr / (s.normA * s.normr);
    const double test3 = 1.0 / s.condA;
    const double z_lo = s.normA * nx_lo / s.normb, z_hi = s.normA * nx_hi / s.normb;
    int istop;
    if (iter >= s.maxiter) istop = 7;
    else if (1.0 + test3 <= 1.0) istop = 6;
    else if (1.0 + test2 <= 1.0) istop = 5;
    else {
        const double t1_hi = test1 / (1.0 + z_lo) * (1.0 + 1e-12), t1_lo = test1 / (1.0 + z_hi) * (1.0 - 1e-12);
        if (1.0 + t1_hi <= 1.0) istop = 4;                        // rule 4 fires wherever ||x|| lies
        else if (!(1.0 + t1_lo > 1.0)) return false;              // it may fire (or a NaN is about): exact evaluation
        else if (test3 <= s.ctol) istop = 3;
        else if (test2 <= s.atol) istop = 2;
        else {
            const double rt_lo = (s.btol + s.atol * z_lo) * (1.0 - 1e-12), rt_hi = (s.btol + s.atol * z_hi) * (1.0 + 1e-12);
            if (test1 <= rt_lo) istop = 1;
            else if (test1 > rt_hi) istop = 0;
            else return false;
        }
    }
    s.iter = iter;
    s.normx = -1.0;            // (not formed: nothing reads it once the rules are decided)
    s.istop = istop;
    if (istop) { s.done = 1; s.notdone = 0; }
    return true;
}
// ... and with ||x||^2 = total: lsmr_commit's evaluation, without the store
__device__ inline void lsmr_decide_exact(LsmrState &s, double total) {
    s.iter += 1;
    s.normx = sqrt(total);
    const double test1 = s.normr / s.normb;
    const double test2 = s.normAr / (s.normA * s.normr);
    const double test3 = 1.0 / s.condA;
    const double t1 = test1 / (1.0 + s.normA * s.normx / s.normb);
    const double rtol = s.btol + s.atol * s.normA * s.normx / s.normb;
    int istop = 0;
    if (s.iter >= s.maxiter) istop = 7;
    else if (1.0 + test3 <= 1.0) istop = 6;
    else if (1.0 + test2 <= 1.0) istop = 5;
    else if (1.0 + t1 <= 1.0) istop = 4;
    else if (test3 <= s.ctol) istop = 3;
    else if (test2 <= s.atol) istop = 2;
    else if (test1 <= rtol) istop = 1;
    s.istop = istop;
    if (istop) { s.done = 1; s.notdone = 0; }
}

// ordered_sum256's association (thread t < 256 adds partials t, t + 256, ...; wave tree; waves 0-3 in order) evaluated by ONE
// wave: the same bits as the block-wide version that K2's epilogue uses for the same arrays -- beta is the same number wherever
// it is formed
__device__ __forceinline__ double wave_ordered_sum256(const double *partials, const int *count, int lane) {
    if (!partials) return 0.0;
    const int cnt = *count;
    double g[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = w * 64 + lane;
        const double p0 = partials[t];                // (the arrays hold at least 256 entries: one round of loads)
        double acc = t < cnt ? p0 : 0.0;
        for (int i = t + 256; i < cnt; i += 256) acc += partials[i];
        g[w] = wave_sum(acc);
    }
    return ((g[0] + g[1]) + g[2]) + g[3];
}
// sum(x_k^2), x_k = x + c2 (c1 hbar + h), by one wave: lane l takes elements l, l + 64, ... in order, then the wave tree
__device__ __forceinline__ double wave_exact_normx2(const LsmrFused &a, double c1, double c2, int lane) {
    double acc = 0.0;
    for (int j = lane; j < a.n; j += 64) {
        const double hb = a.hbar_in[j] * c1 + a.h_in[j];
        const double xj = a.x_in[j] + c2 * hb;
        acc += xj * xj;
    }
    return wave_sum(acc);
}
// Wave 0 of every workgroup: alpha, beta, the rotations of the finished iteration and its stop decision, into `ns` (LDS).
// Everything is a function of the same global arrays, evaluated by the same instructions: identical in every workgroup.
__device__ __noinline__ void lsmr_fused_scalars(const LsmrFused &a, LsmrState &ns, int lane) {
    static_assert(sizeof(LsmrState) % 8 == 0 && sizeof(LsmrState) / 8 <= 64, "state copy: one 8-byte word per lane");
    // (all loads of the chain requested together: state word, the three partial arrays, the three norms)
    unsigned long long word = 0;
    if (lane < (int)(sizeof(LsmrState) / 8)) word = ((const unsigned long long *)a.st_in)[lane];
    double pn0 = 0.0, pn1 = 0.0, pn2 = 0.0;
#pragma unroll
    for (int k = 0; k < LSQ_FUSED_UB_MAX; ++k) {
        const bool in = k < a.ub;
        const double t0 = a.pn_in[k], t1 = a.pn_in[LSQ_FUSED_UB_MAX + k], t2 = a.pn_in[2 * LSQ_FUSED_UB_MAX + k];
        pn0 += in ? t0 : 0.0;
        pn1 += in ? t1 : 0.0;
        pn2 += in ? t2 : 0.0;
    }
    const double beta2 = wave_ordered_sum256(a.pu_in, a.npu_in, lane);
    const double betax2 = wave_ordered_sum256(a.px_in, a.npx_in, lane);
    const double alpha2 = wave_ordered_sum256(a.pv, a.npv, lane);
    if (lane < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)&ns)[lane] = word;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int need_exact = 0;
    if (lane == 0 && !ns.done) {
        const bool was_first = ns.first != 0;
        lsmr_scalars(ns, beta2, betax2, alpha2, a.dg != nullptr, a.px_in != nullptr);
        if (was_first) {
            ns.first = 0;
            if (!(ns.normAr != 0.0)) { ns.done = 1; ns.notdone = 0; }      // lsmr.jl:115: exit if b = 0 or A'b = 0
        } else if (a.force_exact) {
            need_exact = 1;
        } else {
            const double nx = sqrt(pn0), step = fabs(ns.c2) * (fabs(ns.c1) * sqrt(pn1) + sqrt(pn2));
            const double nx_hi = (nx + step) * (1.0 + 1e-12), nx_lo = fmax(0.0, nx - step) * (1.0 - 1e-12);
            need_exact = lsmr_decide_bounded(ns, nx_lo, nx_hi) ? 0 : 1;
        }
    }
    need_exact = __builtin_amdgcn_readfirstlane(need_exact);
    if (need_exact) {      // (wave-uniform; rare: test1 within ~1e-6 of btol)
        __threadfence_block();
        const double total = wave_exact_normx2(a, ns.c1, ns.c2, lane);
        if (lane == 0) lsmr_decide_exact(ns, total);
    }
    __threadfence_block();
}

template <int = 0>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_lsmr_fused(SellDev S, int wrows, int m, int nxpad, LsmrFused a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_BIG_NT / 64];
    __shared__ LsmrState ns;
    __shared__ int s_next, s_stop;
    double *xl = smem;            // nxpad doubles
    double *yw = smem + nxpad;    // LSQ_SELL_ROWS_MAX doubles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n;
    const bool upd = (int)blockIdx.x < a.ub;   // update workgroups come first in the dispatch order: never queued behind a product

    if (upd) {
        // ---- lsmr.jl:78/124, 152-156, iterative_lsmr.jl:92,195-196: the n-vector updates; commit + progress word first ----
        const int was_first = a.st_in->first;
        if (wv == 0) lsmr_fused_scalars(a, ns, lane);
        __syncthreads();
        if (a.st_in->done) {    // a launch queued behind a finished solve: hand the state on (kernels behind it read st_out)
            if (blockIdx.x == 0 && tid < (int)(sizeof(LsmrState) / 8))
                ((unsigned long long *)a.st_out)[tid] = ((const unsigned long long *)&ns)[tid];
            return;
        }
        const bool done_now = ns.done != 0;
        if (blockIdx.x == 0) {
            if (tid < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)a.st_out)[tid] = ((const unsigned long long *)&ns)[tid];
            if (tid == 0) {
                if (!was_first) {
                    __hip_atomic_store((double *)&a.mail->test1, ns.normr / ns.normb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store((double *)&a.mail->test2, ns.normAr / (ns.normA * ns.normr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                publish(a.mail, &ns);
            }
        }
        const double vs = ns.vscale, cu = ns.cu, c1 = ns.c1, c2 = ns.c2, c3 = ns.c3;
        double ax = 0.0, ahb = 0.0, ah = 0.0, aux = 0.0;
        for (int j = blockIdx.x * LSQ_BIG_NT + tid; j < n; j += a.ub * LSQ_BIG_NT) {
            const double Pj = a.P ? a.P[j] : 1.0;
            const double vj = a.vt[j] * vs;                 // lsmr.jl:78,124 rmul!(v, inv(alpha))
            a.v[j] = vj;
            if (was_first) {                                // :89-90, iterative_lsmr.jl:183,242
                a.h_out[j] = vj;
                a.hbar_out[j] = 0.0;
                a.x_out[j] = 0.0;
                a.xout[j] = 0.0;
                ah += vj * vj;
            } else {
                const double hj = a.h_in[j];
                const double hb = a.hbar_in[j] * c1 + hj;   // :152-153
                a.hbar_out[j] = hb;
                const double xj = a.x_in[j] + c2 * hb;      // :154
                a.x_out[j] = xj;
                const double hn = hj * c3 + vj;             // :155-156
                a.h_out[j] = hn;
                a.xout[j] = a.P ? xj * Pj : xj;             // the caller's x always holds P.*x of the newest iterate
                ax += xj * xj;
                ahb += hb * hb;
                ah += hn * hn;
            }
            if (a.dg && !done_now) {    // damped rows of the NEXT u (iterative_lsmr.jl:92): u~x <- d.*t - cu*u~x
                const double tj = a.P ? vj * Pj : vj;       // iterative_lsmr.jl:31 ldiv!(tmp, P, a)
                const double un = tj * a.dg[j] - cu * a.ux[j];
                a.ux[j] = un;
                aux += un * un;
            }
        }
        const double bx = block_sum<LSQ_BIG_NT>(ax, sh);
        const double bhb = block_sum<LSQ_BIG_NT>(ahb, sh);
        const double bh = block_sum<LSQ_BIG_NT>(ah, sh);
        const double bux = block_sum<LSQ_BIG_NT>(aux, sh);
        if (tid == 0) {
            a.pn_out[blockIdx.x] = bx;
            a.pn_out[LSQ_FUSED_UB_MAX + blockIdx.x] = bhb;
            a.pn_out[2 * LSQ_FUSED_UB_MAX + blockIdx.x] = bh;
            a.px_out[blockIdx.x] = bux;
            if (blockIdx.x == 0) *a.npx_out = a.ub;
        }
        return;
    }

    // ---- product workgroups: u~ <- vs * (J w) - cu * u~ with w = (P .* s) .* v~  (lsmr.jl:118; J (P.*v) = (J w) / alpha) ----
    // The product is linear in v: the gather vector is staged UNNORMALISED, straight from v~, so the stream starts at once; 1/alpha
    // and cu -- the end of the scalar chain that wave 0 works through while the other fifteen waves stream -- are applied to the
    // finished dot products.  The waves draw their slices from a counter (LDS), so wave 0's late start costs a sixteenth of it.
    constexpr int XR = (LSQ_LDS_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
    {
        double xr[XR], pr[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) xr[q] = a.vt[min(tid + q * LSQ_BIG_NT, n - 1)];
        if (a.P) {
#pragma unroll
            for (int q = 0; q < XR; ++q) pr[q] = a.P[min(tid + q * LSQ_BIG_NT, n - 1)];
        }
        if (a.cs) {     // column-scaled J = V diag(s): one factor array stays live
            double sr[XR];
#pragma unroll
            for (int q = 0; q < XR; ++q) sr[q] = a.cs[min(tid + q * LSQ_BIG_NT, n - 1)];
#pragma unroll
            for (int q = 0; q < XR; ++q) pr[q] = a.P ? pr[q] * sr[q] : sr[q];
        }
        const int dflag = a.st_in->done;
#pragma unroll
        for (int q = 0; q < XR; ++q)
            if (tid + q * LSQ_BIG_NT < n) xl[tid + q * LSQ_BIG_NT] = (a.P || a.cs) ? xr[q] * pr[q] : xr[q];
        if (dflag) return;      // launches queued behind a finished solve stop here
    }
    constexpr int Q = LSQ_SELL_ROWS_MAX / LSQ_BIG_NT;
    const int pb = (int)blockIdx.x - a.ub, npb = (int)gridDim.x - a.ub;
    double racc = 0.0;
    bool have_scalars = false;
    for (int w = pb; w < S.nblocks; w += npb) {
        const int base = w * wrows, rows = min(wrows, m - base);
        const int s0 = w * S.spw;
        double pre[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) pre[q] = a.uold[base + min(tid + q * LSQ_BIG_NT, rows - 1)];
        if (tid == 0) { s_next = 0; if (!have_scalars) s_stop = 0; }
        __syncthreads();      // w staged / the previous window's epilogue is done with yw / the slice counter is reset
        if (!have_scalars && wv == 0) {
            lsmr_fused_scalars(a, ns, lane);
            if (lane == 0 && ns.done) s_stop = 1;       // the finished iteration was the last: no more slices are drawn
        }
        have_scalars = true;
        auto grab = [&]() -> int {
            int v = 0;
            if (lane == 0) v = atomicAdd(&s_next, 1);
            return __builtin_amdgcn_readfirstlane(v);
        };
        int cur = grab();
        SellSliceRef A = sell_slice_ref(S, s0 + cur, s0 + S.spw, lane);
        while (cur < S.spw) {
            if (*(volatile int *)&s_stop) break;
            const int nxt = grab();
            const SellSliceRef r = A;
            A = sell_slice_ref(S, s0 + nxt, s0 + S.spw, lane);
            const size_t oa = (size_t)r.sm.x + lane * 2;
            const unsigned pos = r.inf & LSQ_SELL_POS_MASK;
            double sum = 0.0, sq = 0.0;
            sell_lane_sum<false>(S.val + oa, S.idx16 + oa, r.sm.y, (int)(r.inf >> LSQ_SELL_POS_BITS), xl, sum, sq);
            if (pos != LSQ_SELL_POS_MASK) yw[pos] = sum;
            cur = nxt;
        }
        __syncthreads();
        if (ns.done) return;                 // (decided by this launch: the update workgroups finish x; nothing to multiply)
        const double vs = ns.vscale, cu = ns.cu;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = tid + q * LSQ_BIG_NT;
            if (i < rows) {
                const double un = vs * yw[i] - cu * pre[q];
                a.unew[base + i] = un;
                racc += un * un;
            }
        }
    }
    const double bv = block_sum<LSQ_BIG_NT>(racc, sh);
    if (tid == 0) {
        a.pu_out[pb] = bv;
        if (pb == 0) *a.npu_out = npb;
    }
}
