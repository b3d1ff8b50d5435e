// Round 5: the LSMR inner iteration as THREE launches (included by lsq_lsmr.hip behind k_lsmr_update).
//
// K3 (k_lsmr_update: 8 us of latencies for 0.5 MB -- a launch ramp, a round of loads, the scalar chain, a ticket round, a
// system-scope publish) is folded into the NEXT J*v launch (k_lsmr_fused):
//   * a few extra workgroups at the FRONT of the grid (the sliced rows of C4 leave 3 of the 256 CUs without a block) do what K3
//     did: alpha, beta and the rotations of the iteration that has just finished from the deferred partials (lsmr.jl:119-149),
//     ||x|| and the seven stopping rules (lsmr.jl:205-231) -- each of them over ALL of x, redundantly and identically, so no
//     reduction across workgroups is needed --, the n-vector updates of lsmr.jl:152-156 (v, hbar, x, h; the caller's P.*x; the
//     damped rows u~x of iterative_lsmr.jl:92); workgroup 0 commits the state and publishes the progress word;
//   * the J*v workgroups gather the UNNORMALISED w = (v~.*P).*s, which the producer of v~ (K2's epilogue, or the setup) leaves beside
//     it, so their stream starts at once, exactly like k_sell_rows'; J (P.*v) = (J w)/alpha is finished in the
//     epilogue -- like u, v is never normalised on the way into a product (lsmr.jl:118,124) -- with 1/alpha, alpha/beta and the
//     stop decision taken from a record that workgroup 0 published ~15 us earlier (see "hand-off" below).  If the finished
//     iteration was the last, the record says so: waves stop drawing slices when they see it and the epilogue is skipped -- and
//     where the host expects that outcome the launch is CAUTIOUS (below) and nothing streams at all.
// Hand-off inside the launch: workgroup 0 -> every product workgroup, one record {1/alpha, alpha/beta, done} as
// tagged words (flag-in-data); requested by a reader before the last slice of its stream and checked after it (bounded re-reads
// otherwise: workgroup 0 is the FIRST workgroup of the grid, so it is dispatched before any reader; a reader that still gives
// up -- a second of spinning -- leaves its launch's tag in a word of its own; the next launch's update workgroups end the solve
// with istop = 99 and the host returns LSQ_EHIP).
// The state and x, hbar, h are double-buffered (launch k reads set (k-1)&1 and writes set k&1): the update workgroups read ALL of
// x, hbar, h for ||x|| while their siblings write their own thirds; a late workgroup must not read what its own launch commits.
// Where the host expects the launch to find the solve finished (lsq_lsmr_solve: the predicted last iteration; the first iteration
// when nothing is known yet) it asks for a CAUTIOUS launch: the product workgroups wait for the record before they stream.  (The
// launch also comes in halves -- COMMIT-ONLY, grid = the update workgroups, and PRODUCT-ONLY, ub = 0, the record of the commit-only
// launch already there -- LSQ_LSMR_HALVES=1: the first form of the same idea, kept for A/B.)
#pragma once

constexpr int LSQ_FUSED_UB_MAX = 4;
// flag-in-data, like the other in-launch exchanges of the library: five 64-bit words, each 32 bits of payload under the 32-bit
// tag of the launch that wrote it (1/alpha and alpha/beta as two halves each, the decision).  A reader needs no ordering between
// the words: a word either carries this launch's tag -- then its payload is this launch's -- or it does not, and consecutive
// launches never share a tag.
struct LsmrHandoff {
    unsigned long long w[8];
};
__device__ __forceinline__ void lsmr_handoff_write(LsmrHandoff *ho, unsigned tag, double vs, double cu, int done) {
    const unsigned long long t = (unsigned long long)tag << 32;
    const unsigned long long bv = (unsigned long long)__double_as_longlong(vs), bc = (unsigned long long)__double_as_longlong(cu);
    __hip_atomic_store(&ho->w[0], t | (bv & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ho->w[1], t | (bv >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ho->w[2], t | (bc & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ho->w[3], t | (bc >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ho->w[4], t | (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A reader that gives up waiting for the record leaves its launch's tag in w[6] -- a word nobody else writes (marking
// the committed state itself was unsafe, workgroup 0 of the same launch copies its whole state over it when it finally runs).  The
// update workgroups of the NEXT launch find the tag there and end the solve with istop = 99; the host returns LSQ_EHIP.
__device__ __forceinline__ void lsmr_handoff_fail(LsmrHandoff *ho, unsigned tag) {
    __hip_atomic_store(&ho->w[6], (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct LsmrHandoffWords { unsigned long long w[5]; };
__device__ __forceinline__ LsmrHandoffWords lsmr_handoff_request(const LsmrHandoff *ho) {
    LsmrHandoffWords r;
#pragma unroll
    for (int k = 0; k < 5; ++k) r.w[k] = __hip_atomic_load(&ho->w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}
__device__ __forceinline__ bool lsmr_handoff_valid(const LsmrHandoffWords &r, unsigned tag) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 5; ++k) ok = ok && (unsigned)(r.w[k] >> 32) == tag;
    return ok;
}
struct LsmrFused {
    const LsmrState *st_in;
    LsmrState *st_out;
    LsqMailbox *mail;
    LsmrHandoff *ho;
    unsigned tag;                                // of this launch's record: differs from the previous launch's
    unsigned tag_prev;                           // the previous launch's tag: a reader of THAT launch that gave up left it in ho->w[6]
                                                 // (0: launch 1 of a solve -- the launch in front of it is the setup, nothing to check)
    const double *pu_in; const int *npu_in;      // sum(u~_y^2): the previous launch's product workgroups (or the setup)
    double *pu_out; int *npu_out;
    const double *px_in; const int *npx_in;      // sum(u~_x^2): the previous launch's update workgroups (null: u~_x == 0)
    double *px_out; int *npx_out;
    const double *pv; const int *npv;            // sum(v~^2): the previous K2 (or the setup)
    const double *vt;                            // v~ (n)
    const double *w;                             // (v~ .* P) .* s: the gather vector, left by the producer of v~
    const double *P, *dg;                        // preconditioner (or null), sqrt(damp) (or null)
    const double *h_in, *hbar_in, *x_in;
    double *h_out, *hbar_out, *x_out;
    double *v, *xout, *ux;
    const double *uold; double *unew;            // m
    int n, ub;
    int test_no_record;    // test hook (LSQ_TEST_EXCHANGE_TIMEOUT): workgroup 0 keeps its record to itself and the readers give up early
    int cautious;      // the product workgroups wait for the record BEFORE they stream (this launch may well find the solve finished)
};

// ||x||^2 = total: lsmr_commit's evaluation of the rules (lsmr.jl:205-231), without the store
__device__ inline void lsmr_decide(LsmrState &s, double total) {
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

constexpr long long LSQ_FUSED_SPIN_LIMIT = 1LL << 24;      // x ~64 cycles of s_sleep: about a second

// The progress word WITHOUT release semantics: a system-scope release makes the storing wave write the L2 back and wait for every
// store it has in flight (~2 us measured on the critical path of the commit-only launch).  Nothing the host reads on seeing the
// word depends on that order: results are read behind a stream synchronisation, and the two hints beside the word only feed a
// prediction (the host re-reads the word around them; a stale pair costs a wrong guess).
__device__ __forceinline__ void publish_relaxed(LsqMailbox *mail, const LsmrState *st) {
    unsigned long long w = ((unsigned long long)(st->epoch & 0x7fffffu) << 41) |
                           ((unsigned long long)(st->done ? 1 : 0) << 40) |
                           ((unsigned long long)(st->istop & 0xff) << 32) | (unsigned)st->iter;
    __hip_atomic_store((unsigned long long *)mail, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ordered_sum256x3 (lsq_spmv.h: same association, same bits) with the first 512 entries of every array requested up front: K2's
// 313 partials of sum(v~^2) then cost ONE round of loads -- beside 250 streaming workgroups a round is microseconds.  The arrays
// hold 4096 entries.
__device__ __forceinline__ void ordered_sum512x3(const double *pa, const int *na, const double *pb, const int *nb,
                                                 const double *pc, const int *nc, double &ra, double &rb, double &rc) {
    __shared__ double s_w5[3][4];
    __shared__ double s_tot5[3];
    const int tid = threadIdx.x;
    if (tid < 256) {
        const double a0 = pa ? pa[tid] : 0.0, b0 = pb ? pb[tid] : 0.0, c0 = pc ? pc[tid] : 0.0;
        const double a1 = pa ? pa[tid + 256] : 0.0, b1 = pb ? pb[tid + 256] : 0.0, c1 = pc ? pc[tid + 256] : 0.0;
        const int ca = pa ? *na : 0, cb = pb ? *nb : 0, cc = pc ? *nc : 0;
        double a = tid < ca ? a0 : 0.0, b = tid < cb ? b0 : 0.0, c = tid < cc ? c0 : 0.0;
        if (tid + 256 < ca) a += a1;
        if (tid + 256 < cb) b += b1;
        if (tid + 256 < cc) c += c1;
        for (int i = tid + 512; i < ca; i += 256) a += pa[i];
        for (int i = tid + 512; i < cb; i += 256) b += pb[i];
        for (int i = tid + 512; i < cc; i += 256) c += pc[i];
        a = wave_sum(a);
        b = wave_sum(b);
        c = wave_sum(c);
        if ((tid & 63) == 0) {
            s_w5[0][tid >> 6] = a;
            s_w5[1][tid >> 6] = b;
            s_w5[2][tid >> 6] = c;
        }
    }
    __syncthreads();
    if (tid < 3) s_tot5[tid] = ((s_w5[tid][0] + s_w5[tid][1]) + s_w5[tid][2]) + s_w5[tid][3];
    __syncthreads();
    ra = s_tot5[0];
    rb = s_tot5[1];
    rc = s_tot5[2];
    __syncthreads();
}

// Block reductions for the tail of the update workgroups, where global STORES are in flight: __syncthreads() makes a wave wait for
// its outstanding stores (vmcnt(0)) before it joins the barrier -- microseconds here (measured: 4.4 us at the last barrier of the
// commit-only launch) for an exchange that only goes through LDS.  These wait for the LDS counter alone.
__device__ __forceinline__ void lsq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool IS_MAX>
__device__ __forceinline__ double lsq_block_reduce_lds(double v, double *sh /* 16 doubles */) {
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    lsq_lds_barrier();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = sh[0];
#pragma unroll
        for (int i = 1; i < LSQ_BIG_NT / 64; ++i) r = IS_MAX ? fmax(r, sh[i]) : r + sh[i];
    }
    lsq_lds_barrier();
    return r;
}

template <int = 0>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_lsmr_fused(SellDev S, int wrows, int m, int nxpad, LsmrFused a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_BIG_NT / 64];
    __shared__ LsmrState ns;
    __shared__ double s_vs, s_cu;
    __shared__ int s_done;
    double *xl = smem;            // nxpad doubles
    double *yw = smem + nxpad;    // LSQ_SELL_ROWS_MAX doubles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n;
    const bool upd = (int)blockIdx.x < a.ub;   // update workgroups come first in the dispatch order: never queued behind a product

    if (upd) {
        // ---- what K3 did (lsmr.jl:119-156, 205-231; iterative_lsmr.jl:92,195-196) ----
        static_assert(sizeof(LsmrState) % 8 == 0 && sizeof(LsmrState) / 8 <= LSQ_BIG_NT, "state copy: one 8-byte word per thread");
        // These few workgroups run beside 250 streaming ones: a load takes microseconds, so every round of loads is issued as ONE
        // batch.  Round 1: the state, the partials (inside ordered_sum256x3) and all of x, hbar, h for ||x||.
        constexpr int NQ = (LSQ_LDS_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
        double xi[NQ], hbi[NQ], hi[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = min(tid + q * LSQ_BIG_NT, n - 1);
            xi[q] = a.x_in[j];
            hbi[q] = a.hbar_in[j];
            hi[q] = a.h_in[j];
        }
        if (tid < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)&ns)[tid] = ((const unsigned long long *)a.st_in)[tid];
        // (did a product workgroup of the previous launch give up on its record?  every update workgroup looks, all decide alike)
        const bool prev_failed = a.tag_prev != 0u &&
                                 (unsigned)__hip_atomic_load(&a.ho->w[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.tag_prev;
        double beta2, betax2, alpha2;
        ordered_sum512x3(a.pu_in, a.npu_in, a.px_in, a.npx_in, a.pv, a.npv, beta2, betax2, alpha2);   // (its barriers also publish ns)
        if (ns.done) {    // a launch queued behind a finished solve: hand the state on (kernels behind it read st_out), release the readers
            if (blockIdx.x == 0) {
                if (tid < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)a.st_out)[tid] = ((const unsigned long long *)&ns)[tid];
                if (tid == 0) {
                    lsmr_handoff_write(a.ho, a.tag, 1.0, 0.0, 1);
                    publish_relaxed(a.mail, &ns);      // (again: a failure mark of the previous launch's readers reaches the host this way)
                }
            }
            return;
        }
        const bool was_first = ns.first != 0;
        if (tid == 0) lsmr_scalars(ns, beta2, betax2, alpha2, a.dg != nullptr, a.px_in != nullptr);
        __syncthreads();
        if (!was_first) {
            // ||x_k||^2 over ALL of x in every update workgroup alike (thread-strided, wave tree, 16 waves in order): no
            // cross-workgroup reduction, the same bits everywhere
            const double c1 = ns.c1, c2 = ns.c2;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double hb = hbi[q] * c1 + hi[q];
                const double xj = xi[q] + c2 * hb;
                acc += (tid + q * LSQ_BIG_NT < n) ? xj * xj : 0.0;
            }
            const double total = block_sum<LSQ_BIG_NT>(acc, sh);
            if (tid == 0) lsmr_decide(ns, total);
        } else if (tid == 0) {
            ns.first = 0;
            if (!(ns.normAr != 0.0)) { ns.done = 1; ns.notdone = 0; }      // lsmr.jl:115: exit if b = 0 or A'b = 0
        }
        if (tid == 0 && prev_failed) { ns.istop = 99; ns.done = 1; ns.notdone = 0; }   // u~ of the previous launch is incomplete
        __syncthreads();
        const bool done_now = ns.done != 0;
        const double vs = ns.vscale, cu = ns.cu, c1 = ns.c1, c2 = ns.c2, c3 = ns.c3;
        if (blockIdx.x == 0) {
            if (tid == 0 && !a.test_no_record) lsmr_handoff_write(a.ho, a.tag, vs, cu, done_now ? 1 : 0);   // first: somebody may be waiting for it
            if (tid < (int)(sizeof(LsmrState) / 8)) ((unsigned long long *)a.st_out)[tid] = ((const unsigned long long *)&ns)[tid];
            if (tid == 0) {
                if (!was_first) {   // hints for the host's prediction of the stop iteration, in front of the progress word
                    __hip_atomic_store((double *)&a.mail->test1, ns.normr / ns.normb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store((double *)&a.mail->test2, ns.normAr / (ns.normA * ns.normr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                publish_relaxed(a.mail, &ns);
            }
        }
        // Round 2: this workgroup's own elements (every ub-th group of 1024), again one batch of loads
        constexpr int NU = (NQ + 2) / 3;      // ub >= 3
        double e_vt[NU], e_P[NU], e_dg[NU], e_ux[NU], e_h[NU], e_hb[NU], e_x[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int j = min((int)(blockIdx.x + k * a.ub) * LSQ_BIG_NT + tid, n - 1);
            e_vt[k] = a.vt[j];
            e_P[k] = a.P ? a.P[j] : 1.0;
            e_dg[k] = a.dg ? a.dg[j] : 0.0;
            e_ux[k] = a.dg ? a.ux[j] : 0.0;
            e_h[k] = a.h_in[j];
            e_hb[k] = a.hbar_in[j];
            e_x[k] = a.x_in[j];
        }
        double aux = 0.0;
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int j = (int)(blockIdx.x + k * a.ub) * LSQ_BIG_NT + tid;
            if (j >= n) continue;
            const double Pj = e_P[k];
            const double vj = e_vt[k] * vs;                 // lsmr.jl:78,124 rmul!(v, inv(alpha))
            a.v[j] = vj;
            if (was_first) {                                // :89-90, iterative_lsmr.jl:183,242
                a.h_out[j] = vj;
                a.hbar_out[j] = 0.0;
                a.x_out[j] = 0.0;
                a.xout[j] = 0.0;
            } else {
                const double hj = e_h[k];
                const double hb = e_hb[k] * c1 + hj;        // :152-153
                a.hbar_out[j] = hb;
                const double xj = e_x[k] + c2 * hb;         // :154
                a.x_out[j] = xj;
                a.h_out[j] = hj * c3 + vj;                  // :155-156
                a.xout[j] = a.P ? xj * Pj : xj;             // the caller's x always holds P.*x of the newest iterate
            }
            if (a.dg && !done_now) {    // damped rows of the NEXT u (iterative_lsmr.jl:92): u~x <- d.*t - cu*u~x
                const double tj = a.P ? vj * Pj : vj;       // iterative_lsmr.jl:31 ldiv!(tmp, P, a)
                const double un = tj * e_dg[k] - cu * e_ux[k];
                a.ux[j] = un;
                aux += un * un;
            }
        }
        if (done_now) return;        // (no next iteration: nobody reads sum(u~x^2))
        const double bux = lsq_block_reduce_lds<false>(aux, sh);     // (block_sum's association: wave tree, 16 waves in order)
        if (tid == 0) {
            a.px_out[blockIdx.x] = bux;
            if (blockIdx.x == 0) *a.npx_out = a.ub;
        }
        return;
    }

    // ---- product workgroups: u~ <- (J w)/alpha - cu u~ with w = (v~ .* P) .* s  (lsmr.jl:118) ----
    constexpr int XR = (LSQ_LDS_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
    {
        double xr[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) xr[q] = a.w[min(tid + q * LSQ_BIG_NT, n - 1)];
        const int dflag = a.st_in->done;
#pragma unroll
        for (int q = 0; q < XR; ++q)
            if (tid + q * LSQ_BIG_NT < n) xl[tid + q * LSQ_BIG_NT] = xr[q];
        if (dflag) return;      // launches queued behind a finished solve stop here
    }
    constexpr int Q = LSQ_SELL_ROWS_MAX / LSQ_BIG_NT;
    const int pb = (int)blockIdx.x - a.ub, npb = (int)gridDim.x - a.ub;
    double racc = 0.0;
    bool have_scalars = false;
    if (a.cautious) {
        // CAUTIOUS launch: the host expects (or cannot exclude) that the finished iteration was the last.  The product workgroups
        // look at workgroup 0's record first -- ~8 us into the launch -- and stream only if the solve goes on: a launch that commits
        // the stop then costs what the update workgroups cost, and one that does not starts its product 8 us late instead of paying a
        // wasted product, a skipped tail and a host round trip.  (Same bounded wait, same failure path as below.)
        if (tid == 0) {
            LsmrHandoffWords rec = lsmr_handoff_request(a.ho);
            long long spins = 0;
            while (!lsmr_handoff_valid(rec, a.tag)) {
                if (++spins > (a.test_no_record ? 4096 : LSQ_FUSED_SPIN_LIMIT)) break;
                __builtin_amdgcn_s_sleep(8);
                rec = lsmr_handoff_request(a.ho);
            }
            const bool ok = spins <= (a.test_no_record ? 4096 : LSQ_FUSED_SPIN_LIMIT);
            s_vs = __longlong_as_double((long long)((rec.w[0] & 0xffffffffull) | (rec.w[1] << 32)));
            s_cu = __longlong_as_double((long long)((rec.w[2] & 0xffffffffull) | (rec.w[3] << 32)));
            s_done = ok ? (int)(rec.w[4] & 0xffffffffull) : 2;
            if (!ok) lsmr_handoff_fail(a.ho, a.tag);
        }
        __syncthreads();
        if (s_done) return;
        have_scalars = true;
    }
    for (int w = pb; w < S.nblocks; w += npb) {
        const int base = w * wrows, rows = min(wrows, m - base);
        const int s0 = w * S.spw, s1 = s0 + S.spw;
        double pre[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) pre[q] = a.uold[base + min(tid + q * LSQ_BIG_NT, rows - 1)];
        // (sell_wave_slices with one addition: wave 0 requests workgroup 0's record BEFORE its last slice, so that the record's
        //  latency -- microseconds on a saturated memory system -- passes during that slice instead of holding up the epilogue)
        LsmrHandoffWords rec;
#pragma unroll
        for (int k = 0; k < 5; ++k) rec.w[k] = 0ull;
        {
            constexpr int NW = LSQ_BIG_NT / 64;
            int sidx = s0 + wv;
            SellSliceRef A = sell_slice_ref(S, sidx, s1, lane);
            __syncthreads();      // w staged / the previous window's epilogue is done with yw
            unsigned long long poll = 0ull;     // the decision word as it was when this wave's previous slice began
            for (; sidx < s1; sidx += NW) {
                // the finished iteration was the last: stop streaming (the launch that commits the stop costs ~12 us instead of a
                // whole product; the word was requested a slice ago, so looking at it waits for nothing)
                if (!have_scalars && (unsigned)(poll >> 32) == a.tag && (poll & 1ull)) break;
                if (!have_scalars) poll = __hip_atomic_load(&a.ho->w[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const SellSliceRef r = A;
                A = sell_slice_ref(S, sidx + NW, s1, lane);
                if (!have_scalars && tid == 0 && sidx + NW >= s1) rec = lsmr_handoff_request(a.ho);
                const size_t oa = (size_t)r.sm.x + lane * 2;
                const unsigned pos = r.inf & LSQ_SELL_POS_MASK;
                double sum = 0.0, sq = 0.0;
                sell_lane_sum<false>(S.val + oa, S.idx16 + oa, r.sm.y, (int)(r.inf >> LSQ_SELL_POS_BITS), xl, sum, sq, lane);
                if (pos != LSQ_SELL_POS_MASK) yw[pos] = sum;
            }
        }
        if (!have_scalars && tid == 0) {
            // (published ~15 us before the early request in the usual case; otherwise ask again, bounded)
            long long spins = 0;
            while (!lsmr_handoff_valid(rec, a.tag)) {
                if (++spins > (a.test_no_record ? 4096 : LSQ_FUSED_SPIN_LIMIT)) break;
                __builtin_amdgcn_s_sleep(8);
                rec = lsmr_handoff_request(a.ho);
            }
            const bool ok = spins <= (a.test_no_record ? 4096 : LSQ_FUSED_SPIN_LIMIT);
            s_vs = __longlong_as_double((long long)((rec.w[0] & 0xffffffffull) | (rec.w[1] << 32)));
            s_cu = __longlong_as_double((long long)((rec.w[2] & 0xffffffffull) | (rec.w[3] << 32)));
            s_done = ok ? (int)(rec.w[4] & 0xffffffffull) : 2;
            if (!ok) lsmr_handoff_fail(a.ho, a.tag);      // never silent: the next launch ends the solve with istop = 99
        }
        have_scalars = true;
        __syncthreads();
        if (s_done) return;                 // the finished iteration was the last (the update workgroups finish x), or the hand-off failed
        const double vs = s_vs, cu = s_cu;
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
