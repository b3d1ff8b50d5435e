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
//       J*x : <= 4096 consecutive rows            (gather vector x: all n entries in LDS)
//       J'*y: <= 5120 consecutive columns of one <= 8192-row gather window (y[window] in LDS)
//   * inside a block the outputs are sorted by entry count and grouped 64 at a time (a slice,
//     lane = output); a slice stores max-count (rounded up to even) entries per lane, interleaved
//     in pairs:  slot(j, lane) = off + ((j/2)*64 + lane)*2 + (j%2)   -> a lane reads one 16-byte
//     value pair and one 4-byte index pair per step, a wave reads 1 KiB + 256 B contiguous.
//     Sorting makes the padding small (~7 % on the Poisson(10) rows of the C4 workload).
//   * info[slice*64 + lane] = position of the output inside the block | true entry count << 13;
//     slices are stored in "snake" order so that the 16 waves (slice s0+wave, +16, ...) get equal work.
//   * results are dropped into an LDS window at their ORIGINAL position, then one coalesced pass
//     applies the caller's epilogue (J*x) or writes the per-window column partials (J'*y), so the
//     vectors keep the user's order everywhere.
#pragma once

constexpr int LSQ_SELL_POS_BITS = 13;
constexpr unsigned LSQ_SELL_POS_MASK = (1u << LSQ_SELL_POS_BITS) - 1u;   // 0x1fff = "no output" (padding lane)
constexpr int LSQ_SELL_ROWS_MAX = 4096;    // J*x: output rows per block
constexpr int LSQ_SELL_GROWS_MAX = 8192;   // J'*y: rows of the gather window (64 KiB of y in LDS)
constexpr int LSQ_SELL_CCOLS_MAX = 5120;   // J'*y: output columns per block (40 KiB, twice with squares)

struct SellDev {
    const int *wslice;            // nblocks+1 slice ranges
    const int2 *smeta;            // per slice {entry offset, padded entry count}
    const unsigned *info;         // per (slice, lane)
    const unsigned short *idx16;  // gather index per stored entry
    const double *val;
    int nblocks;
};

// all products of one lane's output inside one slice, added in index order
// SCALE: the stored values are `vp[..] * sc` (a column-scaled model Jacobian J = A diag(s) whose column copy has not been
// materialised yet): every value pair is scaled as it is loaded and written to `dp` at the same offset, so that this pass IS
// the materialisation (the padding entries are zeros in the source and stay zeros).
template <bool SQ, bool SCALE = false>
__device__ __forceinline__ void sell_lane_sum(const double *__restrict__ vp, const unsigned short *__restrict__ ip, int L,
                                              int len, const double *xl, double &sum, double &sq, double sc = 1.0,
                                              double *__restrict__ dp = nullptr) {
    int j = 0;
    for (; j + 8 <= L; j += 8) {   // four 20-byte groups in flight per lane
        double2 a[4];
        unsigned c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const double2 *>(vp + (size_t)(j / 2 + u) * 128);
            c[u] = *reinterpret_cast<const unsigned *>(ip + (size_t)(j / 2 + u) * 128);
        }
        if constexpr (SCALE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u].x *= sc;
                a[u].y *= sc;
                *reinterpret_cast<double2 *>(dp + (size_t)(j / 2 + u) * 128) = a[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // every operand is fetched and every sum formed unconditionally; padding is dropped by SELECTION (a
            // conditional add lets the compiler sink the value fetch into an exec-masked block behind a full
            // s_waitcnt vmcnt(0) inside this loop).  Same additions in the same order as before.
            double p0 = a[u].x * xl[c[u] & 0xffffu], p1 = a[u].y * xl[c[u] >> 16];
            asm volatile("" : "+v"(p0), "+v"(p1));   // (the products exist here, whatever the selections below)
            const bool in0 = j + 2 * u < len, in1 = j + 2 * u + 1 < len;
            const double t0 = sum + p0;
            sum = in0 ? t0 : sum;
            const double t1 = sum + p1;
            sum = in1 ? t1 : sum;
            if constexpr (SQ) {
                const double q0 = sq + a[u].x * a[u].x;
                sq = in0 ? q0 : sq;
                const double q1 = sq + a[u].y * a[u].y;
                sq = in1 ? q1 : sq;
            }
        }
    }
    for (; j < L; j += 2) {
        double2 a = *reinterpret_cast<const double2 *>(vp + (size_t)(j / 2) * 128);
        const unsigned c = *reinterpret_cast<const unsigned *>(ip + (size_t)(j / 2) * 128);
        if constexpr (SCALE) {
            a.x *= sc;
            a.y *= sc;
            *reinterpret_cast<double2 *>(dp + (size_t)(j / 2) * 128) = a;
        }
        double p0 = a.x * xl[c & 0xffffu], p1 = a.y * xl[c >> 16];
        asm volatile("" : "+v"(p0), "+v"(p1));
        const bool in0 = j < len, in1 = j + 1 < len;
        const double t0 = sum + p0;
        sum = in0 ? t0 : sum;
        const double t1 = sum + p1;
        sum = in1 ? t1 : sum;
        if constexpr (SQ) {
            const double q0 = sq + a.x * a.x;
            sq = in0 ? q0 : sq;
            const double q1 = sq + a.y * a.y;
            sq = in1 ? q1 : sq;
        }
    }
}

// ---- J*x: dot of every row with x, then the epilogue on (row, dot) in row order ----------------
template <class Epi>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_sell_rows(SellDev S, int wrows, int m, const double *__restrict__ x,
                                                          int nx, int nxpad, Epi epi) {
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
        double pre[Q];
        if constexpr (EpiHasPre<Epi>::value) {   // epilogue inputs of this window: in flight during the stream
#pragma unroll
            for (int q = 0; q < Q; ++q) pre[q] = epi.pre(base + min(tid + q * LSQ_BIG_NT, rows - 1));
        }
        __syncthreads();   // x staged / the previous window's epilogue is done with yw
        const int s0 = S.wslice[w], s1 = S.wslice[w + 1];
        for (int s = s0 + wv; s < s1; s += LSQ_BIG_NT / 64) {
            const int2 sm = S.smeta[__builtin_amdgcn_readfirstlane(s)];
            const unsigned inf = S.info[(size_t)s * 64 + lane];
            double sum = 0.0, sq = 0.0;
            sell_lane_sum<false>(S.val + (size_t)sm.x + lane * 2, S.idx16 + (size_t)sm.x + lane * 2, sm.y,
                                 (int)(inf >> LSQ_SELL_POS_BITS), xl, sum, sq);
            const unsigned pos = inf & LSQ_SELL_POS_MASK;
            if (pos != LSQ_SELL_POS_MASK) yw[pos] = sum;
        }
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

// ---- J'*y: per (gather window, column) partial sums; k_combine adds the windows ----------------
// block b = gw * ncb + cb; part layout [gw][n] (SQ: [gw][2n] = dots | squares)
// SCALE (see sell_lane_sum): S.val is the UNSCALED source, `scale[col]` the column factors, `dval` the layout's own value
// array, which this pass fills.
template <bool SQ, bool SCALE = false>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_sell_cols(SellDev S, int ncb, int ccols, int grows, int m, int n,
                                                          const double *__restrict__ y, double *__restrict__ part,
                                                          const int *done, const double *__restrict__ scale = nullptr,
                                                          double *__restrict__ dval = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *yl = smem;                          // LSQ_SELL_GROWS_MAX doubles
    double *ow = smem + LSQ_SELL_GROWS_MAX;     // LSQ_SELL_CCOLS_MAX doubles (+ the same again for SQ)
    double *ow2 = ow + LSQ_SELL_CCOLS_MAX;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int YR = LSQ_SELL_GROWS_MAX / LSQ_BIG_NT;
    // window of y -> LDS, all loads of a thread issued before the first use
    auto stage_y = [&](int b) {
        const int gw = b / ncb, gbase = gw * grows, rows = min(grows, m - gbase);
        double yr[YR];
#pragma unroll
        for (int j = 0; j < YR; ++j) yr[j] = y[gbase + min(tid + j * LSQ_BIG_NT, rows - 1)];
#pragma unroll
        for (int j = 0; j < YR; ++j)
            if (tid + j * LSQ_BIG_NT < grows) yl[tid + j * LSQ_BIG_NT] = (tid + j * LSQ_BIG_NT < rows) ? yr[j] : 0.0;
    };
    // the first window is fetched together with the `done` flag of a finished solve (one latency, not two in a row)
    const int dflag = done ? *done : 0;
    if ((int)blockIdx.x < S.nblocks) stage_y(blockIdx.x);
    if (dflag) return;
    for (int b = blockIdx.x; b < S.nblocks; b += gridDim.x) {
        const int gw = b / ncb, cb = b - gw * ncb;
        const int cbase = cb * ccols, cols = min(ccols, n - cbase);
        if (b != (int)blockIdx.x) {
            __syncthreads();   // the previous block's output pass is done with ow / yl
            stage_y(b);
        }
        __syncthreads();
        const int s0 = S.wslice[b], s1 = S.wslice[b + 1];
        for (int s = s0 + wv; s < s1; s += LSQ_BIG_NT / 64) {
            const int2 sm = S.smeta[__builtin_amdgcn_readfirstlane(s)];
            const unsigned inf = S.info[(size_t)s * 64 + lane];
            const unsigned pos = inf & LSQ_SELL_POS_MASK;
            double sum = 0.0, sq = 0.0;
            if constexpr (SCALE) {
                const double sc = pos != LSQ_SELL_POS_MASK ? scale[cbase + (int)pos] : 0.0;    // (lane = one column of the block)
                sell_lane_sum<SQ, true>(S.val + (size_t)sm.x + lane * 2, S.idx16 + (size_t)sm.x + lane * 2, sm.y,
                                        (int)(inf >> LSQ_SELL_POS_BITS), yl, sum, sq, sc, dval + (size_t)sm.x + lane * 2);
            } else {
                sell_lane_sum<SQ>(S.val + (size_t)sm.x + lane * 2, S.idx16 + (size_t)sm.x + lane * 2, sm.y,
                                  (int)(inf >> LSQ_SELL_POS_BITS), yl, sum, sq);
            }
            if (pos != LSQ_SELL_POS_MASK) {
                ow[pos] = sum;
                if constexpr (SQ) ow2[pos] = sq;
            }
        }
        __syncthreads();
        double *dst = part + (size_t)gw * (SQ ? 2 : 1) * n + cbase;
        for (int i = tid; i < cols; i += LSQ_BIG_NT) {
            dst[i] = ow[i];
            if constexpr (SQ) dst[n + i] = ow2[i];
        }
    }
}
