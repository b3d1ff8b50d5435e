// Sliced layouts ("SELL-64, sorted inside the output window") for the two big sparse products.
//
// Why: with the gathered vector in LDS the products are pure streams of (value, 16-bit index), and
// what kept the segment kernels (k_seg_stream_lds / k_bcsc_lds) at ~63 % of the HBM peak was not the
// stream but the segmented reduction around it -- products through LDS, two workgroup barriers per
// tile, per-segment ptr loads.  Here every lane owns one output (a row of J for J*x, one column
// inside one row-window for J'*y) and adds its products in a register, in index order (= the
// reference's summation order, utils.jl / SparseArrays mul!): no product staging, no barrier inside
// the stream, no ptr array.
//
// Layout, built once per pattern (lsq_sparse.hip: build_sell):
//   * outputs are cut into blocks ("windows") that one 1024-thread workgroup owns:
//       J*x : <= 4096 consecutive rows            (gather vector x: all n entries in LDS; n > 12160: one column window of
//                                                   x at a time, k_sell_rows_wide, block = (row block, window))
//       J'*y: <= 2560 consecutive columns of one <= 16384-row gather window (y[window] in LDS); blocks are placed XCD-aware
//             (lsq_xcd_block: the column blocks of a window behind one L2)
//   * inside a block the outputs are sorted by entry count and grouped 64 at a time (a slice,
//     lane = output); a slice stores max-count L entries per lane, interleaved
//     in pairs:  slot(j, lane) = off + ((j/2)*64 + lane)*2 + (j%2)   -> a lane reads one 16-byte
//     value pair and one 4-byte index pair per step, a wave reads 1 KiB + 256 B contiguous;
//     an ODD slice (round 5: L is no longer rounded up to even) ends with one unpaired entry per lane in a
//     compact group of its own:  slot(L-1, lane) = off + (L/2)*128 + lane  (512 B + 128 B per wave).
//     Sorting makes the padding small (~7 % on the Poisson(10) rows of the C4 workload).
//   * every block is padded to the same number of slices (empty ones at the end), so block b owns slices
//     [b * spw, (b + 1) * spw) and no per-block table has to be read at the head of a launch;
//   * info[slice*64 + lane] = position of the output inside the block | true entry count << 13;
//     slices are stored in "snake" order so that the 16 waves (slice s0+wave, +16, ...) get equal work.
//   * results are dropped into an LDS window at their ORIGINAL position, then one coalesced pass
//     applies the caller's epilogue (J*x) or writes the per-window column partials (J'*y), so the
//     vectors keep the user's order everywhere.
#pragma once

constexpr int LSQ_SELL_POS_BITS = 13;
constexpr unsigned LSQ_SELL_POS_MASK = (1u << LSQ_SELL_POS_BITS) - 1u;   // 0x1fff = "no output" (padding lane)
constexpr int LSQ_SELL_ROWS_MAX = 4096;    // J*x: output rows per block
// J'*y: a block = (gather window of <= 16384 rows of y in LDS: 128 KiB) x (<= 2560 output columns: 20 KiB).  Round 3: twice the
// rows and half the columns of rounds 1-2 (8192 x 5120) -- same number of blocks, but a column's entries inside a window
// double (C4: 7.8 -> 15.6), so the padding of the pair-interleaved slices halves (6.4 % -> 3 %), there are half as many
// per-window partials to write and to combine (C4: 128 -> 64 per column) and half as many lane descriptors to read.
constexpr int LSQ_SELL_CW_AUTO = 6;       // ... chosen by default up to this many (n <= 72960), see lsq_csc_create
constexpr int LSQ_SELL_CW_MAX = 32;       // J*x with n > LSQ_LDS_X_MAX: at most this many column windows of x (k_sell_rows_wide)
constexpr int LSQ_SELL_GROWS_MAX = 16384;
constexpr int LSQ_SELL_CCOLS_MAX = 2560;

struct SellDev {
    const int2 *smeta;            // per slice {entry offset, padded entry count}
    const unsigned *info;         // per (slice, lane)
    const unsigned short *idx16;  // gather index per stored entry
    const double *val;
    int nblocks;
    int spw;                      // slices per block: block b owns slices [b * spw, (b + 1) * spw)
};

// One batch of U value/index pairs of a lane: ALL loads first (each lane has U 20-byte requests in flight), then the sums.
// CLAMP: the slice has fewer than U pairs left; the surplus loads re-read the last pair and their products are dropped by the
// `len` selection, like every padding entry.  (Measured alternatives on C4, J*v launch: skipping the surplus loads behind
// wave-uniform branches 24.4 us against 23.7 us -- the branches split the batch; a fixed 8-wide first batch 24.5 us -- the
// surplus gathers and adds are not free either.)
template <int U>
struct SellBatch {
    double2 a[U];
    unsigned c[U];
};
template <int U, bool CLAMP>
__device__ __forceinline__ void sell_batch_load(SellBatch<U> &B, const double *__restrict__ vp, const unsigned short *__restrict__ ip,
                                                int p0, int np) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = CLAMP ? min(p0 + u, np - 1) : p0 + u;
        B.a[u] = *reinterpret_cast<const double2 *>(vp + (size_t)q * 128);
        B.c[u] = *reinterpret_cast<const unsigned *>(ip + (size_t)q * 128);
    }
}
template <int U, bool CLAMP, bool SQ>
__device__ __forceinline__ void sell_batch_sum(SellBatch<U> &B, int p0, int np, int len, const double *xl, double &sum, double &sq) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // every operand is fetched and every sum formed unconditionally; padding is dropped by SELECTION (a
        // conditional add lets the compiler sink the value fetch into an exec-masked block behind a full
        // s_waitcnt vmcnt(0) inside this loop).  The additions happen in entry order.
        const double x0 = xl[B.c[u] & 0xffffu], x1 = xl[B.c[u] >> 16];
        double p0v = B.a[u].x * x0, p1v = B.a[u].y * x1;
        asm volatile("" : "+v"(p0v), "+v"(p1v));   // (the products exist here, whatever the selections below)
        const int j = 2 * (p0 + u);
        // (clamped surplus pairs are never selected: an odd slice has len up to 2 np + 1, so the pair index decides, not len alone)
        const bool pin = !CLAMP || p0 + u < np;
        const bool in0 = pin && j < len, in1 = pin && j + 1 < len;
        const double t0 = sum + p0v;
        sum = in0 ? t0 : sum;
        const double t1 = sum + p1v;
        sum = in1 ? t1 : sum;
        if constexpr (SQ) {
            const double q0 = sq + B.a[u].x * B.a[u].x;
            sq = in0 ? q0 : sq;
            const double q1 = sq + B.a[u].y * B.a[u].y;
            sq = in1 ? q1 : sq;
        }
    }
}

// all products of one lane's output inside one slice (L entries per lane, even and wave-uniform; `len` of them real), added in
// index order.  A slice of the C4 workload holds 4-10 pairs per lane: one batch, one memory round trip (the round-1 loop paid one
// per 4 pairs plus one per leftover pair, in sequence).
// The unpaired last entry of an odd slice (compact group behind the pairs).  vp / ip point at the lane's first pair (off + 2 lane):
// the group's element for this lane sits at off + (L/2)*128 + lane.  Requested with the batches, added last (index order).
struct SellTail {
    double a;
    unsigned short c;
};
__device__ __forceinline__ SellTail sell_tail_load(const double *__restrict__ vp, const unsigned short *__restrict__ ip, int L, int lane) {
    SellTail t;
    const long long o = (long long)(L >> 1) * 128 - lane;
    t.a = vp[o];
    t.c = ip[o];
    return t;
}
template <bool SQ>
__device__ __forceinline__ void sell_tail_sum(const SellTail &t, int L, int len, const double *xl, double &sum, double &sq) {
    double p = t.a * xl[t.c];
    asm volatile("" : "+v"(p));
    const bool in = L - 1 < len;
    const double s1 = sum + p;
    sum = in ? s1 : sum;
    if constexpr (SQ) {
        const double q1 = sq + t.a * t.a;
        sq = in ? q1 : sq;
    }
}
template <bool SQ, bool TAIL = true>
__device__ __forceinline__ void sell_lane_sum_from(const double *__restrict__ vp, const unsigned short *__restrict__ ip, int p0, int L,
                                                   int len, const double *xl, double &sum, double &sq, int lane) {
    const int np = L >> 1;
    SellTail tail;
    tail.a = 0.0;
    tail.c = 0;
    if (TAIL && (L & 1)) tail = sell_tail_load(vp, ip, L, lane);      // (wave-uniform; in flight with the batches)
    for (; p0 + 8 <= np; p0 += 8) {
        SellBatch<8> B;
        sell_batch_load<8, false>(B, vp, ip, p0, np);
        sell_batch_sum<8, false, SQ>(B, p0, np, len, xl, sum, sq);
    }
    const int rem = np - p0;
    if (rem > 4) {
        SellBatch<8> B;
        sell_batch_load<8, true>(B, vp, ip, p0, np);
        sell_batch_sum<8, true, SQ>(B, p0, np, len, xl, sum, sq);
    } else if (rem > 2) {
        SellBatch<4> B;
        sell_batch_load<4, true>(B, vp, ip, p0, np);
        sell_batch_sum<4, true, SQ>(B, p0, np, len, xl, sum, sq);
    } else if (rem > 0) {
        SellBatch<2> B;
        sell_batch_load<2, true>(B, vp, ip, p0, np);
        sell_batch_sum<2, true, SQ>(B, p0, np, len, xl, sum, sq);
    }
    if (TAIL && (L & 1)) sell_tail_sum<SQ>(tail, L, len, xl, sum, sq);
}

template <bool SQ>
__device__ __forceinline__ void sell_lane_sum(const double *__restrict__ vp, const unsigned short *__restrict__ ip, int L, int len,
                                              const double *xl, double &sum, double &sq, int lane) {
    sell_lane_sum_from<SQ>(vp, ip, 0, L, len, xl, sum, sq, lane);
}

// The slices of one wave (s0 + wave, + 16, ...), with the next slice's descriptor requested before the current slice's stream.
//   `out(pos, sum, sq)` stores a lane's result.
// Contains the workgroup barrier that separates staging the gather vector from the first gather: the first descriptor is
// requested before it.  (Requesting TWO slices per wave up front made the J*v launch slower, 29.5 us against 23.7 us: the
// stream is not short of requests in flight, a CU's miss queue is already full with 16 waves x 5-8 KB.)
struct SellSliceRef {
    int2 sm;        // {entry offset, padded entry count}
    unsigned inf;   // position | true count << 13
};
__device__ __forceinline__ SellSliceRef sell_slice_ref(const SellDev &S, int s, int s1, int lane) {
    SellSliceRef r;
    r.sm = make_int2(0, 0);
    r.inf = LSQ_SELL_POS_MASK;
    if (s < s1) {
        r.sm = S.smeta[__builtin_amdgcn_readfirstlane(s)];
        r.inf = S.info[(size_t)s * 64 + lane];
    }
    return r;
}
template <bool SQ, class Out>
__device__ __forceinline__ void sell_wave_slices(const SellDev &S, int s0, int s1, int wv, int lane, const double *xl, Out out) {
    constexpr int NW = LSQ_BIG_NT / 64;
    int s = s0 + wv;
    SellSliceRef A = sell_slice_ref(S, s, s1, lane);
    __syncthreads();
    for (; s < s1; s += NW) {
        const SellSliceRef a = A;
        A = sell_slice_ref(S, s + NW, s1, lane);
        const size_t oa = (size_t)a.sm.x + lane * 2;
        const unsigned pos = a.inf & LSQ_SELL_POS_MASK;
        double sum = 0.0, sq = 0.0;
        sell_lane_sum<SQ>(S.val + oa, S.idx16 + oa, a.sm.y, (int)(a.inf >> LSQ_SELL_POS_BITS), xl, sum, sq, lane);
        if (pos != LSQ_SELL_POS_MASK) out(pos, sum, sq);
    }
}

// ---- J*x: dot of every row with x, then the epilogue on (row, dot) in row order ----------------
// xscale (or null): the matrix is the stored values times diag(xscale) -- a COLUMN-SCALED Jacobian J = V diag(s) whose scaled
// entries are never materialised (lsq_mat::d_colscale): the gather vector is staged as s .* x, the stream is V's.
template <class Epi>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_sell_rows(SellDev S, int wrows, int m, const double *__restrict__ x,
                                                          const double *__restrict__ xscale, int nx, int nxpad, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_BIG_NT / 64];
    double *xl = smem;            // nxpad doubles
    double *yw = smem + nxpad;    // LSQ_SELL_ROWS_MAX doubles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the gather vector is fetched FIRST, together with the `done` flag of a finished solve and the epilogue's scalars:
    // one memory latency at the head of the kernel instead of three in a row
    constexpr int XR = (LSQ_LDS_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
    double xr[XR];
#pragma unroll
    for (int q = 0; q < XR; ++q) xr[q] = x[min(tid + q * LSQ_BIG_NT, nx - 1)];
    if (xscale) {   // (wave-uniform; the factors travel with the gather vector: still one latency at the head)
        double sr[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) sr[q] = xscale[min(tid + q * LSQ_BIG_NT, nx - 1)];
#pragma unroll
        for (int q = 0; q < XR; ++q) xr[q] *= sr[q];
    }
    const int dflag = epi.done ? *epi.done : 0;

    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    if constexpr (EpiHasPrepare<Epi>::value) epi.prepare();
#pragma unroll
    for (int q = 0; q < XR; ++q)
        if (tid + q * LSQ_BIG_NT < nx) xl[tid + q * LSQ_BIG_NT] = xr[q];
    if (dflag) return;   // launches queued behind a finished solve stop here
    constexpr int Q = LSQ_SELL_ROWS_MAX / LSQ_BIG_NT;
    double racc = 0.0;
    for (int w = blockIdx.x; w < S.nblocks; w += gridDim.x) {
        const int base = w * wrows, rows = min(wrows, m - base);
        const int s0 = w * S.spw, s1 = s0 + S.spw;
        double pre[Q];
        if constexpr (EpiHasPre<Epi>::value) {   // epilogue inputs of this window: in flight during the stream
#pragma unroll
            for (int q = 0; q < Q; ++q) pre[q] = epi.pre(base + min(tid + q * LSQ_BIG_NT, rows - 1));
        }
        // (the barrier inside: x staged / the previous window's epilogue is done with yw)
        sell_wave_slices<false>(S, s0, s1, wv, lane, xl, [&](unsigned pos, double sum, double) { yw[pos] = sum; });
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = tid + q * LSQ_BIG_NT;
            if (i < rows) {
                if constexpr (EpiHasPre<Epi>::value) epi.seg_pre(base + i, yw[i], pre[q], racc);
                else epi.seg(base + i, yw[i], racc);
            }
        }
    }
    // side work (e.g. damped rows) is laid out for LSQ_NT-thread blocks
    for (int e = blockIdx.x; e < epi.extra_blocks; e += gridDim.x)
        if (tid < LSQ_NT) epi.extra(e, racc);
    finish_block_nt<LSQ_BIG_NT>(epi, racc, sh);
}

// ---- TWO products with the same stored values in ONE pass: rows of V*(s .* xa) and of V*xb --------------------------
// The tail of an LM iteration on a model r(x) = V phi(x) - b with the column-scaled Jacobian J = V diag(s) streams V twice:
// the predicted residual |J dx - f|^2 (levenberg_marquardt.jl:114-117) and the trial residual f!(x_trial) = V phi(x_trial) - b
// with its sum of squares (:107, :111) -- 2 x 107 MB at C4 for 16 MB of vectors.  Here every (value, index) pair is loaded
// once and multiplied into both gather vectors.  Both vectors must be resident in LDS (2 x 8n bytes: n <= 10200 with the
// 160 KB of a CU), which leaves no room for the output window of k_sell_rows: a lane applies the epilogue to its row itself
// (f[i], b[i] requested at the head of the slice, the trial residual stored straight from the lane -- 8-byte accesses inside
// the block's 32 KB window, merged in L2).  A row's two sums are formed left to right by one lane, as everywhere: the bits
// of J dx and of V phi(x_trial) are those of the two separate launches; the two sums of squares are added per lane in slice
// order, then wave tree / 16 waves / blocks in index order (deterministic; another association than the row-order pass of
// the separate launches, i.e. equal to it to round-off).
template <int U, bool CLAMP>
__device__ __forceinline__ void sell_batch_sum2(SellBatch<U> &B, int p0, int np, int len, const double *xa, const double *xb,
                                                double &sa, double &sb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i0 = B.c[u] & 0xffffu, i1 = B.c[u] >> 16;
        double a0 = B.a[u].x * xa[i0], a1 = B.a[u].y * xa[i1], b0 = B.a[u].x * xb[i0], b1 = B.a[u].y * xb[i1];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));   // (the products exist here: see sell_batch_sum)
        const int j = 2 * (p0 + u);
        const bool pin = !CLAMP || p0 + u < np;
        const bool in0 = pin && j < len, in1 = pin && j + 1 < len;
        const double ta0 = sa + a0;
        sa = in0 ? ta0 : sa;
        const double ta1 = sa + a1;
        sa = in1 ? ta1 : sa;
        const double tb0 = sb + b0;
        sb = in0 ? tb0 : sb;
        const double tb1 = sb + b1;
        sb = in1 ? tb1 : sb;
    }
}
__device__ __forceinline__ void sell_lane_sum2(const double *__restrict__ vp, const unsigned short *__restrict__ ip, int L, int len,
                                               const double *xa, const double *xb, double &sa, double &sb, int lane) {
    const int np = L >> 1;
    int p0 = 0;
    SellTail tail;
    tail.a = 0.0;
    tail.c = 0;
    if (L & 1) tail = sell_tail_load(vp, ip, L, lane);      // (the unpaired last entry of an odd slice)
    for (; p0 + 8 <= np; p0 += 8) {
        SellBatch<8> B;
        sell_batch_load<8, false>(B, vp, ip, p0, np);
        sell_batch_sum2<8, false>(B, p0, np, len, xa, xb, sa, sb);
    }
    const int rem = np - p0;
    if (rem > 4) {
        SellBatch<8> B;
        sell_batch_load<8, true>(B, vp, ip, p0, np);
        sell_batch_sum2<8, true>(B, p0, np, len, xa, xb, sa, sb);
    } else if (rem > 2) {
        SellBatch<4> B;
        sell_batch_load<4, true>(B, vp, ip, p0, np);
        sell_batch_sum2<4, true>(B, p0, np, len, xa, xb, sa, sb);
    } else if (rem > 0) {
        SellBatch<2> B;
        sell_batch_load<2, true>(B, vp, ip, p0, np);
        sell_batch_sum2<2, true>(B, p0, np, len, xa, xb, sa, sb);
    }
    if (L & 1) {
        double pa = tail.a * xa[tail.c], pb = tail.a * xb[tail.c];
        asm volatile("" : "+v"(pa), "+v"(pb));
        const bool in = L - 1 < len;
        const double ta = sa + pa, tb = sb + pb;
        sa = in ? ta : sa;
        sb = in ? tb : sb;
    }
}
constexpr int LSQ_PAIR_X_MAX = 10200;      // doubles per gather vector: 2 x 10200 x 8 B + the kernel's static LDS (~140 B) <= 160 KB
struct SellPairEpi {
    const int *done;            // skip flag of a launch queued behind an undecided LSMR solve (LsmrTail), or null
    // the two m-vectors the epilogue reads come in SLICE ORDER (element s * 64 + lane = the row that lane of slice s owns;
    // k_sell_perm_rows): a lane's 8-byte access to its row's own position is one cache line per lane -- 64 line requests per
    // wave instruction, as many as the whole value + index stream of the slice; measured 42.7 us with f, b and the output all
    // in row order, 35.6 us with f and b in slice order, 30.8 us with the output in slice order too (profiles/r04)
    const double *fa;           // f (current residual), slice order:  ra = (V (s .* xa))_i - fa_i,  sum ra^2 -> *slot_a
    const double *fb;           // b (model constant), slice order:    rb = (V xb)_i - fb_i,  sum rb^2 -> *slot_b
    double *out_b;              // rb in row order (what the gather of J'f and the caller read)
    double *out_b_perm;         // rb in slice order (the next iteration's fa if the step is accepted)
    double *part_a, *part_b;    // block partials (gridDim.x each)
    unsigned *counter;          // ticket slot of grid_reduce
    double *slot_a, *slot_b;
    LsqSlotPublish pub;         // the iteration's scalars -> host once both sums are final
    // LM's acceptance test taken on the device as well (levenberg_marquardt.jl:118-122, the host's own expression on the same
    // doubles): *gate = 0 if the step will be accepted, 1 if not -- the skip word of the NEXT iteration's gradient pass, which
    // is queued behind this kernel before the host has seen the scalars (null: no such pass is queued)
    int *gate;
    double ssr, min_quality;
    double *slot_gate;          // the decision as a scalar (1.0 accepted / 0.0 not) among the published ones: the host adopts the
                                // queued pass only if the device took the decision the host takes
};
// dst[s * 64 + lane] = src[row owned by lane of slice s] (0 for lanes without a row): an m-vector in slice order
template <int = 0>
__global__ void __launch_bounds__(LSQ_NT) k_sell_perm_rows(SellDev S, int wrows, int nslices, const double *__restrict__ src,
                                                           double *__restrict__ dst) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < (long long)nslices * 64; i += (long long)gridDim.x * LSQ_NT) {
        const unsigned inf = S.info[i];
        const unsigned pos = inf & LSQ_SELL_POS_MASK;
        const int w = (int)(i >> 6) / S.spw;
        dst[i] = pos != LSQ_SELL_POS_MASK ? src[(size_t)w * wrows + pos] : 0.0;
    }
}
template <int = 0>     // (a template so that the header can be included by several translation units)
__global__ void __launch_bounds__(LSQ_BIG_NT) k_sell_rows_pair(SellDev S, int wrows, int m, const double *__restrict__ xa,
                                                               const double *__restrict__ sa_scale, const double *__restrict__ xb,
                                                               int nx, int nxpad, SellPairEpi e) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_BIG_NT / 64];
    double *la = smem, *lb = smem + nxpad;
    constexpr int NW = LSQ_BIG_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int XR = (LSQ_PAIR_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
    {   // both gather vectors and the factors: one memory latency at the head
        double ra[XR], rs[XR], rb[XR];
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            const int i = min(tid + q * LSQ_BIG_NT, nx - 1);
            ra[q] = xa[i];
            rs[q] = sa_scale ? sa_scale[i] : 1.0;
            rb[q] = xb[i];
        }
        const int dflag = e.done ? *e.done : 0;
#pragma unroll
        for (int q = 0; q < XR; ++q)
            if (tid + q * LSQ_BIG_NT < nx) {
                la[tid + q * LSQ_BIG_NT] = ra[q] * rs[q];
                lb[tid + q * LSQ_BIG_NT] = rb[q];
            }
        if (dflag) return;
    }
    // The trial residual goes out twice: in slice order (coalesced, straight from the lane) and in ROW order.  A lane's 8-byte store
    // to "its" row is one cache line per lane (round 4: 2.06x the write traffic this kernel owes, 38.3 us instead of 30.8 with
    // everything in slice order).  Round 5: a lane keeps the residuals of its (at most G) slices in registers; when the block's
    // stream is over the gather vectors are dead, so their LDS becomes the output window that did not fit beside them -- results
    // dropped at their row positions, one coalesced pass to memory.  (A workgroup that owns several blocks re-stages the gather
    // vectors from the cache for the next one; C4: one block per workgroup.)
    constexpr int G = LSQ_SELL_ROWS_MAX / 64 / NW;      // slices per wave and block
    double acc_a = 0.0, acc_b = 0.0;
    for (int w = blockIdx.x; w < S.nblocks; w += gridDim.x) {
        const int base = w * wrows, rows = min(wrows, m - base);
        const int s0 = w * S.spw, s1 = s0 + S.spw;
        if (w != (int)blockIdx.x) {           // the output pass of the previous block used the window: stage the gather vectors again
            __syncthreads();
            for (int i = tid; i < nx; i += LSQ_BIG_NT) {
                la[i] = xa[i] * (sa_scale ? sa_scale[i] : 1.0);
                lb[i] = xb[i];
            }
        }
        int s = s0 + wv;
        SellSliceRef A = sell_slice_ref(S, s, s1, lane);
        __syncthreads();                      // the gather vectors are staged -- nothing else is shared
        double keep[G];
        unsigned kpos[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            kpos[g] = LSQ_SELL_POS_MASK;
            keep[g] = 0.0;
            if (s < s1) {                     // (wave-uniform)
                const SellSliceRef a = A;
                A = sell_slice_ref(S, s + NW, s1, lane);
                const unsigned pos = a.inf & LSQ_SELL_POS_MASK;
                const bool valid = pos != LSQ_SELL_POS_MASK;
                const size_t pidx = (size_t)s * 64 + lane;
                const double fa = e.fa[pidx], fb = e.fb[pidx];        // (coalesced; in flight during the stream)
                const size_t oa = (size_t)a.sm.x + lane * 2;
                double sum_a = 0.0, sum_b = 0.0;
                sell_lane_sum2(S.val + oa, S.idx16 + oa, a.sm.y, (int)(a.inf >> LSQ_SELL_POS_BITS), la, lb, sum_a, sum_b, lane);
                if (valid) {
                    const double r_a = sum_a - fa, r_b = sum_b - fb;
                    e.out_b_perm[pidx] = r_b;
                    keep[g] = r_b;
                    kpos[g] = pos;
                    acc_a += r_a * r_a;
                    acc_b += r_b * r_b;
                }
                s += NW;
            }
        }
        __syncthreads();                      // every gather of this block is done: the vectors' LDS is free
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (kpos[g] != LSQ_SELL_POS_MASK) smem[kpos[g]] = keep[g];
        __syncthreads();
        for (int i = tid; i < rows; i += LSQ_BIG_NT) e.out_b[base + i] = smem[i];
    }
    const double ba = block_sum<LSQ_BIG_NT>(acc_a, sh);
    const double bb = block_sum<LSQ_BIG_NT>(acc_b, sh);
    if (tid == 0) __hip_atomic_store(&e.part_b[blockIdx.x], bb, RLX_AGENT);   // (drained with part_a's store inside grid_reduce)
    double total_a = 0.0;
    const bool last = grid_reduce<LSQ_BIG_NT>(ba, e.part_a, e.counter, gridDim.x, sh, [&](double t) { total_a = t; });
    if (!last) return;
    double pb = 0.0;
    for (int i = tid; i < (int)gridDim.x; i += LSQ_BIG_NT) pb += __hip_atomic_load(&e.part_b[i], RLX_AGENT);
    const double total_b = block_sum<LSQ_BIG_NT>(pb, sh);
    if (tid == 0) {
        *e.slot_a = total_a;
        *e.slot_b = total_b;
        if (e.gate) {
            const double pred_red = fabs(e.ssr - total_a);
            const double rho = pred_red > 0 ? (e.ssr - total_b) / pred_red : 0.0;
            const bool acc = rho > e.min_quality;
            *e.gate = acc ? 0 : 1;
            *e.slot_gate = acc ? 1.0 : 0.0;
        }
        if (e.pub.count > 0) {
            for (int i = 0; i < e.pub.count; ++i) {
                const double *src = e.pub.src + i;
                const double v = src == e.slot_a ? total_a : (src == e.slot_b ? total_b : *src);
                __hip_atomic_store(e.pub.dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(e.pub.seq_word, e.pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- J*x for n > LSQ_LDS_X_MAX: x passes through LDS one column window at a time --------------------
// Block b = row block * ncw + window holds the rows' entries inside that window (rows without entries there have no
// lane).  The row block's workgroup walks the windows in ascending order; a lane CONTINUES the row's running sum from the
// output window in LDS (sum = yw[pos]; sum += products in column order; yw[pos] = sum), so every row is still one
// left-to-right sum over its entries -- the reference's order, bit for bit, as in k_sell_rows.
// A (row block, window) holds few entries per row (n = 50000, 10 per row: two), so its slices are short and a wave that
// took them one after the other would spend its time in memory round trips with 1-2 KB in flight.  Instead a wave requests
// the first two pairs of ALL its slices of the window (<= 4: 4096 rows = 64 slices over 16 waves) together with the window
// of x, before the barrier that hands the window over; their descriptors were requested during the previous window.  Only
// rows with more than four entries inside one window pay further round trips.
// 512 threads (8 waves, one workgroup per CU: 256 VGPRs per lane) so that the heads of a wave's 8 slices, the window of x and
// the next window's descriptors are all live in registers at once; with 1024 threads (128 VGPRs) the same code spilled.
constexpr int LSQ_WIDE_NT = 512;
// Width cap of a column window: 12544 doubles (98 KB) beside the 32 KB output window.  A little more than LSQ_LDS_X_MAX so
// that n <= 25088 needs TWO windows: measured at 10^6 x 25000 (nnz 10^7) 39.9 / 35.8 / 32.8 us with 5 / 4 / 3 windows.
constexpr int LSQ_SELL_WIDE_X_MAX = 12544;
constexpr int LSQ_SELL_WIDE_G = LSQ_SELL_ROWS_MAX / 64 / (LSQ_WIDE_NT / 64);   // slices per wave and (row block, window)

template <class Epi>
__global__ void __launch_bounds__(LSQ_WIDE_NT) k_sell_rows_wide(SellDev S, int wrows, int m, int ncw, int cwidth,
                                                                const double *__restrict__ x, int nx, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_WIDE_NT / 64];
    constexpr int NW = LSQ_WIDE_NT / 64, G = LSQ_SELL_WIDE_G;
    double *xl = smem;              // cwidth doubles (cwidth even)
    double *yw = smem + cwidth;     // LSQ_SELL_ROWS_MAX doubles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrb = S.nblocks / ncw;
    constexpr int XR = (LSQ_SELL_WIDE_X_MAX + LSQ_WIDE_NT - 1) / LSQ_WIDE_NT;
    constexpr int Q = LSQ_SELL_ROWS_MAX / LSQ_WIDE_NT;
    SellSliceRef A[G];
    double xr[XR];
    auto refs_at = [&](int w, int cw) {   // descriptors of this wave's slices of (row block w, window cw)
        const int sb = (w * ncw + cw) * S.spw;
#pragma unroll
        for (int g = 0; g < G; ++g) A[g] = sell_slice_ref(S, sb + wv + g * NW, sb + S.spw, lane);
    };
    auto fetch_x = [&](int cw) {
        const int c0 = cw * cwidth;
#pragma unroll
        for (int q = 0; q < XR; ++q) xr[q] = x[min(c0 + tid + q * LSQ_WIDE_NT, nx - 1)];
    };
    if ((int)blockIdx.x < nrb) refs_at(blockIdx.x, 0);
    const int dflag = epi.done ? *epi.done : 0;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    if constexpr (EpiHasPrepare<Epi>::value) epi.prepare();
    if (dflag) return;   // launches queued behind a finished solve stop here
    double racc = 0.0;
    for (int w = blockIdx.x; w < nrb; w += gridDim.x) {
        const int base = w * wrows, rows = min(wrows, m - base);
        for (int cw = 0; cw < ncw; ++cw) {
            // the window of x and the head of every slice of this window: requested before anything is waited for
            // (measured and dropped: the NEXT window of x requested behind the LDS write, to travel during the sums -- its 48
            //  registers live across the sums made the kernel spill 75, 45 us instead of 33 at n = 25000)
            fetch_x(cw);
            SellSliceRef a[G];
            SellBatch<2> B[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                a[g] = A[g];
                const size_t oa = (size_t)a[g].sm.x + lane * 2;   // (an empty slice: a valid address, nothing selected)
                sell_batch_load<2, true>(B[g], S.val + oa, S.idx16 + oa, 0, max(a[g].sm.y >> 1, 1));
            }
            const bool more = cw + 1 < ncw, next_block = w + (int)gridDim.x < nrb;
            if (more) refs_at(w, cw + 1);
            else if (next_block) refs_at(w + gridDim.x, 0);
            __syncthreads();   // the previous window's gathers / the previous block's epilogue are done with xl, yw
#pragma unroll
            for (int q = 0; q < XR; ++q)
                if (tid + q * LSQ_WIDE_NT < cwidth) xl[tid + q * LSQ_WIDE_NT] = xr[q];
            if (cw == 0) {
#pragma unroll
                for (int q = 0; q < Q; ++q) yw[tid + q * LSQ_WIDE_NT] = 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // (the column-windowed layout keeps EVEN slices -- build_sell's allow_odd = false: this kernel holds the heads of eight
                //  slices, a window of x and the next descriptors in 256 registers, and every form of the unpaired-entry path spilled)
                const int np = a[g].sm.y >> 1, len = (int)(a[g].inf >> LSQ_SELL_POS_BITS);
                const unsigned pos = a[g].inf & LSQ_SELL_POS_MASK;
                if (np == 0) continue;   // (wave-uniform)
                double sum = pos != LSQ_SELL_POS_MASK ? yw[pos] : 0.0, sq = 0.0;
                sell_batch_sum<2, true, false>(B[g], 0, np, len, xl, sum, sq);
                if (pos != LSQ_SELL_POS_MASK) yw[pos] = sum;
            }
            // rows with more than four entries inside the window: the rest of their slices, one slice after the other
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (a[g].sm.y <= 4) continue;   // (wave-uniform)
                const unsigned pos = a[g].inf & LSQ_SELL_POS_MASK;
                const size_t oa = (size_t)a[g].sm.x + lane * 2;
                double sum = pos != LSQ_SELL_POS_MASK ? yw[pos] : 0.0, sq = 0.0;
                sell_lane_sum_from<false, false>(S.val + oa, S.idx16 + oa, 2, a[g].sm.y, (int)(a[g].inf >> LSQ_SELL_POS_BITS), xl, sum, sq, lane);
                if (pos != LSQ_SELL_POS_MASK) yw[pos] = sum;
            }
        }
        double pre[Q];
        if constexpr (EpiHasPre<Epi>::value) {
#pragma unroll
            for (int q = 0; q < Q; ++q) pre[q] = epi.pre(base + min(tid + q * LSQ_WIDE_NT, rows - 1));
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = tid + q * LSQ_WIDE_NT;
            if (i < rows) {
                if constexpr (EpiHasPre<Epi>::value) epi.seg_pre(base + i, yw[i], pre[q], racc);
                else epi.seg(base + i, yw[i], racc);
            }
        }
    }
    for (int e = blockIdx.x; e < epi.extra_blocks; e += gridDim.x)
        if (tid < LSQ_NT) epi.extra(e, racc);
    finish_block_nt<LSQ_WIDE_NT>(epi, racc, sh);
}

// out = a .* b (the gather vector of a column-scaled wide handle: s .* x, formed once instead of once per row block)
template <int = 0>
__global__ void __launch_bounds__(LSQ_NT) k_sell_vmul(int n, const double *__restrict__ a, const double *__restrict__ b,
                                                      double *__restrict__ out) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) out[i] = a[i] * b[i];
}

// ---- J'*y: per (gather window, column) partial sums; k_combine adds the windows ----------------
// block b = gw * ncb + cb; part layout [gw][n] (SQ: [gw][2n] = dots | squares)
// window of y -> LDS, all loads of a thread issued before the first use
__device__ __forceinline__ void sell_cols_stage_y(int b, int ncb, int grows, int m, const double *__restrict__ y, double *yl, int tid) {
    constexpr int YR = LSQ_SELL_GROWS_MAX / LSQ_BIG_NT;
    const int gw = b / ncb, gbase = gw * grows, rows = min(grows, m - gbase);
    double yr[YR];
#pragma unroll
    for (int j = 0; j < YR; ++j) yr[j] = y[gbase + min(tid + j * LSQ_BIG_NT, rows - 1)];
#pragma unroll
    for (int j = 0; j < YR; ++j)
        if (tid + j * LSQ_BIG_NT < grows) yl[tid + j * LSQ_BIG_NT] = (tid + j * LSQ_BIG_NT < rows) ? yr[j] : 0.0;
}
// the workgroup's blocks: a CONTIGUOUS run of block indices (blocks of one gather window are consecutive, so a workgroup that
// holds several blocks -- wide n: 1280 blocks at n = 50000 -- re-stages its 128 KB window of y only when the window changes,
// not once per block); the y window of the first one is already on its way to LDS (sell_cols_stage_y)
// XCD-aware placement (lsq_xcd_block): the blocks of one gather window (its column blocks) read the same 128 KB of y, so they
// should sit behind ONE L2 -- with runs dealt round-robin the four column blocks of a C4 gather window landed on four XCDs and
// every window of y was fetched four times (32 MB instead of 8 per launch).
__device__ __forceinline__ int sell_cols_wg() { return lsq_xcd_block((int)blockIdx.x, (int)gridDim.x); }
__device__ __forceinline__ int sell_cols_first(int nblocks) {
    const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    return sell_cols_wg() * per;
}
template <bool SQ>
__device__ __forceinline__ void sell_cols_pass(const SellDev &S, int ncb, int ccols, int grows, int m, int n,
                                               const double *__restrict__ y, double *__restrict__ part, double *smem) {
    double *yl = smem;                          // LSQ_SELL_GROWS_MAX doubles
    double *ow = smem + LSQ_SELL_GROWS_MAX;     // LSQ_SELL_CCOLS_MAX doubles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (S.nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = sell_cols_wg() * per, b1 = min(S.nblocks, b0 + per);
    for (int b = b0; b < b1; ++b) {
        const int gw = b / ncb, cb = b - gw * ncb;
        const int cbase = cb * ccols, cols = min(ccols, n - cbase);
        if (b != b0) {
            __syncthreads();   // the previous block's output pass is done with ow (and its gathers with yl)
            if (gw != (b - 1) / ncb) sell_cols_stage_y(b, ncb, grows, m, y, yl, tid);
        }
        const int s0 = b * S.spw, s1 = s0 + S.spw;
        double *dst = part + (size_t)gw * (SQ ? 2 : 1) * n + cbase;
        sell_wave_slices<SQ>(S, s0, s1, wv, lane, yl, [&](unsigned pos, double sum, double sq) {   // (lane = one column)
            ow[pos] = sum;
            // (the squares -- the gradient + colsumabs2 pass of multiplied-out matrices, once per g! -- go straight to
            //  memory: a second LDS staging buffer would not fit next to the 128 KiB window)
            if constexpr (SQ) dst[n + pos] = sq;
        });
        __syncthreads();
        for (int i = tid; i < cols; i += LSQ_BIG_NT) dst[i] = ow[i];
    }
}

template <bool SQ>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_sell_cols(SellDev S, int ncb, int ccols, int grows, int m, int n,
                                                          const double *__restrict__ y, double *__restrict__ part,
                                                          const int *done) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the first window is fetched together with the `done` flag of a finished solve (one latency, not two in a row)
    const int dflag = done ? *done : 0;
    if (sell_cols_first(S.nblocks) < S.nblocks) sell_cols_stage_y(sell_cols_first(S.nblocks), ncb, grows, m, y, smem, threadIdx.x);
    if (dflag) return;
    sell_cols_pass<SQ>(S, ncb, ccols, grows, m, n, y, part, smem);
}
