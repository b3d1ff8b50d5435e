// Segmented sparse product engines for gfx950 (64-wide wavefronts, HBM-bound).
//
// A "segment" is a CSR row (J*x, gather from an n-vector) or a CSC column (J'*y, gather from an
// m-vector).  Both products are GATHERS: the CSC scatter of the reference's SparseArrays mul!
// (call sites levenberg_marquardt.jl:114, iterative_lsmr.jl:91) is avoided by keeping a CSR
// mirror of the fixed pattern.  Three launch plans, picked once per pattern:
//   STREAM  short segments (~10 nnz, C4 rows): a 256-thread block streams a tile of <= 2045
//           consecutive nnz with 16-byte loads (2x double2 + int4 per lane per step), parks the
//           products in LDS, then one thread per segment sums its slice in index order.
//   WAVE    medium segments: one 64-lane wavefront per segment, double2/int2 loads, DPP-free
//           shuffle reduction.
//   BLOCK   long segments (C4 columns, 1000 nnz): one 256-thread block per segment.
// Every plan calls an epilogue functor once per segment with the finished dot product, so the
// LSMR/LM fusions (axpy with the previous vector, damping rows, preconditioner scaling, sum of
// squares for the next norm) cost no extra pass over an m- or n-vector.
// Reductions are two-stage and index-ordered (grid_reduce) => run-to-run deterministic.
#pragma once
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "lsq_common.h"

constexpr int LSQ_TILE_WINDOW = 2048;                // LDS doubles per stream tile
constexpr int LSQ_TILE_NNZ = LSQ_TILE_WINDOW - 3;    // aligned-down window may start 3 early
constexpr int LSQ_TILE_SEGS = LSQ_NT;                // one thread per segment in a tile

struct SegsDev {
    const int *ptr;
    const int *idx;
    const double *val;
    const int *tiles;
    const unsigned short *idx16;
    const int *order;  // optional work-item permutation
    int nseg;
    int ntiles;
    int nnz;
};

// Epilogue concept:
//   static constexpr bool REDUCE;           // grid-reduce a per-segment contribution?
//   const int *done;                        // optional early-exit flag (device), may be null
//   int extra_blocks;                       // blocks appended to the grid for side work
//   __device__ void seg(int s, double dot, double &racc) const;
//   __device__ void extra(int blk, double &racc) const;
//   double *partials; unsigned *counter;    // if REDUCE
//   __device__ void finalize(double total) const;   // thread 0 of the last block

template <bool SQ>
__device__ __forceinline__ double prod1(double v, const double *__restrict__ x, int c) {
    return SQ ? v * v : v * x[c];
}

// Deferred ("consumer-side") finalisation: an epilogue with `using defer = void;` only publishes
// its block partial (plain store) and the number of partials; the NEXT kernel's blocks each add
// the partials in a fixed order (ordered_sum256) -- no arrival tickets, no acquire fence and no
// serial last-block tail in the producing kernel, and still run-to-run deterministic.
template <class E, class = void>
struct EpiDefers : std::false_type {};
template <class E>
struct EpiDefers<E, std::void_t<typename E::defer>> : std::true_type {};
// `block_prepare()` (opt in with `using has_block_prepare = void;`) is executed by every thread
// of every block at kernel start (it may use __syncthreads): the place for consumer-side sums.
template <class E, class = void>
struct EpiHasBlockPrepare : std::false_type {};
template <class E>
struct EpiHasBlockPrepare<E, std::void_t<typename E::has_block_prepare>> : std::true_type {};

// Sum of partials[0..count) identical in every block of every kernel: thread t < 256 adds
// partials[t], partials[t+256], ... in order, waves 0-3 are combined in order, then broadcast.
__device__ __forceinline__ double ordered_sum256(const double *partials, int count) {
    __shared__ double s_w[4];
    __shared__ double s_tot;
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (tid < 256)
        for (int i = tid; i < count; i += 256) acc += partials[i];
    if (tid < 256) {
        acc = wave_sum(acc);
        if ((tid & 63) == 0) s_w[tid >> 6] = acc;
    }
    __syncthreads();
    if (tid == 0) s_tot = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
    __syncthreads();
    const double r = s_tot;
    __syncthreads();
    return r;
}

// Up to three such sums with ONE pair of barriers (the loads of all three are in flight together);
// each result is bit-identical to ordered_sum256 of the same array.  Null arrays give 0.
__device__ __forceinline__ void ordered_sum256x3(const double *pa, const int *na, const double *pb, const int *nb,
                                                 const double *pc, const int *nc, double &ra, double &rb, double &rc) {
    __shared__ double s_w3[3][4];
    __shared__ double s_tot3[3];
    const int tid = threadIdx.x;
    if (tid < 256) {
        // the counts and the first 256 entries of every array are fetched TOGETHER (the arrays hold at least 256 entries;
        // entries past the count are dropped afterwards): one memory round trip instead of two on a latency-bound path.
        // Same sums, bit for bit: 0.0 + pa[tid] == pa[tid] for the non-negative partials summed here.
        const double a0 = pa ? pa[tid] : 0.0, b0 = pb ? pb[tid] : 0.0, c0 = pc ? pc[tid] : 0.0;
        const int ca = pa ? *na : 0, cb = pb ? *nb : 0, cc = pc ? *nc : 0;
        double a = tid < ca ? a0 : 0.0, b = tid < cb ? b0 : 0.0, c = tid < cc ? c0 : 0.0;
        for (int i = tid + 256; i < ca; i += 256) a += pa[i];
        for (int i = tid + 256; i < cb; i += 256) b += pb[i];
        for (int i = tid + 256; i < cc; i += 256) c += pc[i];
        a = wave_sum(a);
        b = wave_sum(b);
        c = wave_sum(c);
        if ((tid & 63) == 0) {
            s_w3[0][tid >> 6] = a;
            s_w3[1][tid >> 6] = b;
            s_w3[2][tid >> 6] = c;
        }
    }
    __syncthreads();
    if (tid < 3) s_tot3[tid] = ((s_w3[tid][0] + s_w3[tid][1]) + s_w3[tid][2]) + s_w3[tid][3];
    __syncthreads();
    ra = s_tot3[0];
    rb = s_tot3[1];
    rc = s_tot3[2];
    __syncthreads();
}

template <int NT, class Epi>
__device__ __forceinline__ void finish_block_nt(const Epi &epi, double racc, double *sh) {
    if constexpr (Epi::REDUCE) {
        double bv = block_sum<NT>(racc, sh);
        if constexpr (EpiDefers<Epi>::value) {
            if (threadIdx.x == 0) {
                epi.partials[blockIdx.x] = bv;
                if (blockIdx.x == 0) *epi.npartials = (int)gridDim.x;
            }
        } else {
            grid_reduce<NT>(bv, epi.partials, epi.counter, gridDim.x, sh, [&](double t) { epi.finalize(t); });
        }
    }
}
template <class Epi>
__device__ __forceinline__ void finish_block(const Epi &epi, double racc, double *sh) {
    finish_block_nt<LSQ_NT>(epi, racc, sh);
}

template <class Epi, bool SQ>
__global__ void __launch_bounds__(LSQ_NT) k_seg_stream(SegsDev S, const double *__restrict__ x, Epi epi) {
    __shared__ __attribute__((aligned(16))) double prod[LSQ_TILE_WINDOW];
    __shared__ double sh[LSQ_NT / 64];
    if (epi.done && *epi.done) return;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    const int tid = threadIdx.x;
    const int nwork = S.ntiles + epi.extra_blocks;
    double racc = 0.0;
    for (int bb = blockIdx.x; bb < nwork; bb += gridDim.x) {
        if (bb >= S.ntiles) {
            epi.extra(bb - S.ntiles, racc);
            continue;
        }
        const int b = S.order ? S.order[bb] : bb;
        const int s0 = S.tiles[b], s1 = S.tiles[b + 1];
        const int k0 = S.ptr[s0], k1 = S.ptr[s1];
        if (k1 - k0 <= LSQ_TILE_NNZ) {
            const int ka = k0 & ~3;  // 32-byte aligned for val, 16-byte for idx
#pragma unroll
            for (int c = 0; c < LSQ_TILE_WINDOW / (4 * LSQ_NT); ++c) {
                const int k = ka + c * 4 * LSQ_NT + 4 * tid;
                if (k < k1) {
                    // arrays are padded by 4 entries, neighbours' indices are valid gather
                    // targets, and out-of-segment products land in LDS slots nobody reads.
                    const double2 v0 = *reinterpret_cast<const double2 *>(S.val + k);
                    const double2 v1 = *reinterpret_cast<const double2 *>(S.val + k + 2);
                    int4 ci = make_int4(0, 0, 0, 0);
                    if (!SQ) ci = *reinterpret_cast<const int4 *>(S.idx + k);
                    double2 p0, p1;
                    p0.x = prod1<SQ>(v0.x, x, ci.x);
                    p0.y = prod1<SQ>(v0.y, x, ci.y);
                    p1.x = prod1<SQ>(v1.x, x, ci.z);
                    p1.y = prod1<SQ>(v1.y, x, ci.w);
                    double2 *dst = reinterpret_cast<double2 *>(prod + (k - ka));
                    dst[0] = p0;
                    dst[1] = p1;
                }
            }
            __syncthreads();
            const int s = s0 + tid;
            if (s < s1) {
                const int a = S.ptr[s] - ka, e = S.ptr[s + 1] - ka;
                double sum = 0.0;
                for (int j = a; j < e; ++j) sum += prod[j];
                epi.seg(s, sum, racc);
            }
            __syncthreads();  // prod is rewritten by the next work item
        } else {
            // a single segment longer than a tile: the whole block strides over it
            double sum = 0.0;
            for (int k = k0 + tid; k < k1; k += LSQ_NT) sum += prod1<SQ>(S.val[k], x, SQ ? 0 : S.idx[k]);
            sum = block_sum<LSQ_NT>(sum, sh);
            if (tid == 0) epi.seg(s0, sum, racc);
        }
    }
    finish_block(epi, racc, sh);
}

// STREAM plan with the gathered vector staged in LDS ("column tiles in LDS"): when x is small
// enough (n*8 <= 96 KiB, e.g. the 80 KB n-vector of C4) every gather x[idx[k]] otherwise costs a
// 128-byte L1 line fill from L2 for 8 useful bytes, and that fill traffic -- not HBM -- bounds the
// kernel.  One persistent 1024-thread workgroup per CU copies x into LDS once (coalesced), then
// streams big tiles of <= 8189 nnz: 16-byte loads of val/idx, products against the LDS copy into
// an LDS product buffer, one thread per segment sums its slice in index order.  The next tile's
// val/idx are prefetched into registers before the current tile is reduced.
constexpr int LSQ_BIG_NT = 1024;
constexpr int LSQ_BIG_WINDOW = 8 * LSQ_BIG_NT;          // product doubles in LDS
constexpr int LSQ_BIG_NNZ = LSQ_BIG_WINDOW - 3;
constexpr int LSQ_BIG_SEGS = LSQ_BIG_NT;
constexpr int LSQ_LDS_X_MAX = 12160;                     // doubles of x staged (95 KiB; 160 KiB LDS total)

// four gather indices of entries k..k+3 (k a multiple of 4): 16-byte int4 or 8-byte ushort4
template <bool IDX16>
__device__ __forceinline__ int4 load_idx4(const SegsDev &S, int k) {
    if constexpr (IDX16) {
        const uint2 p = *reinterpret_cast<const uint2 *>(S.idx16 + k);
        return make_int4((int)(p.x & 0xffffu), (int)(p.x >> 16), (int)(p.y & 0xffffu), (int)(p.y >> 16));
    } else {
        return *reinterpret_cast<const int4 *>(S.idx + k);
    }
}

struct BigTileRegs {  // one big tile's worth of val/idx per thread (8 nnz) + its epilogue inputs
    double2 v0[2], v1[2];
    int4 ci[2];
    int4 meta;            // {s0, s1, k0, k1}: tile extent (wave-uniform)
    int pa, pe;           // this thread's segment slice [pa, pe) in absolute nnz positions
    double pre;           // value the epilogue wants from memory for this thread's segment
};

// optional epilogue hook: `double pre(int s)` is loaded together with the tile (so the global
// read of e.g. the previous u[s] is not a serialized latency in the reduce phase) and handed to
// `seg_pre(s, dot, pre, racc)`.  An epilogue opts in with a member typedef `using has_pre = void;`.
// `prepare()` (opt in with `using has_prepare = void;`) runs once per thread at kernel start, to
// cache device-resident scalars the epilogue needs: a scalar read inside the tile loop would be
// the newest entry of the in-order vmcnt queue and force a full drain of the prefetches.
template <class E, class = void>
struct EpiHasPrepare : std::false_type {};
template <class E>
struct EpiHasPrepare<E, std::void_t<typename E::has_prepare>> : std::true_type {};
template <class E, class = void>
struct EpiHasPre : std::false_type {};
template <class E>
struct EpiHasPre<E, std::void_t<typename E::has_pre>> : std::true_type {};

// `meta[t]` = {first segment, end segment, first nnz, end nnz} of big tile t (one 16-byte load).
// The loads of a tile need its extent, and vmcnt retires in order: fetching the extent right
// before the tile would make every prefetch wait for all older loads.  So extents are fetched two
// tiles ahead of the data, data two tiles ahead of use, and every step issues the same number of
// (clamped, never predicated) loads so that the compiler waits with counted vmcnt.
template <class Epi, bool IDX16>
__global__ void __launch_bounds__(LSQ_BIG_NT) k_seg_stream_lds(SegsDev S, const int4 *__restrict__ meta, int nbig,
                                                               const double *__restrict__ x, int nx, int nxpad,
                                                               Epi epi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[LSQ_BIG_NT / 64];
    double *xl = smem;
    double *prod = smem + nxpad;
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    constexpr bool HAS_PRE = EpiHasPre<Epi>::value;
    const int kmax = (S.nnz + 3) & ~3;   // arrays are padded by 8 entries
    auto load_meta = [&](int tb) {
        int4 mt = meta[tb < nbig ? tb : nbig - 1];
        if (tb >= nbig) mt.y = mt.x;  // dummy tile: no segments
        return mt;
    };
    auto load_data = [&](BigTileRegs &r, int4 mt) {
        r.meta = mt;
        const int ka = mt.z & ~3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = min(ka + c * 4 * LSQ_BIG_NT + 4 * tid, kmax);
            r.v0[c] = *reinterpret_cast<const double2 *>(S.val + k);
            r.v1[c] = *reinterpret_cast<const double2 *>(S.val + k + 2);
            r.ci[c] = load_idx4<IDX16>(S, k);
        }
        const int s = min(mt.x + tid, S.nseg - 1);
        r.pa = S.ptr[s];
        r.pe = S.ptr[s + 1];
        if constexpr (HAS_PRE) r.pre = epi.pre(s);
    };
    double racc = 0.0;
    // products of the tile held in r -> LDS; re-arm r with the tile described by `nm`; fetch the
    // extent of the tile after that into `nm`; reduce; epilogue
    auto step = [&](BigTileRegs &r, int4 &nm, int tb_after_next) {
        // (the plan builder only enables this kernel when every segment fits a big tile, so there
        //  is exactly one code path here and the outstanding-load counts are static)
        const int s0 = r.meta.x, s1 = r.meta.y, k0 = r.meta.z;
        const int ka = k0 & ~3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // unconditional: slots past the tile hold products of (valid, clamped) neighbour
            // entries that nobody reads -- a branch here would cost a full vmcnt(0) drain
            double2 p0, p1;
            p0.x = r.v0[c].x * xl[r.ci[c].x];
            p0.y = r.v0[c].y * xl[r.ci[c].y];
            p1.x = r.v1[c].x * xl[r.ci[c].z];
            p1.y = r.v1[c].y * xl[r.ci[c].w];
            double2 *dst = reinterpret_cast<double2 *>(prod + c * 4 * LSQ_BIG_NT + 4 * tid);
            dst[0] = p0;
            dst[1] = p1;
        }
        const int a = r.pa - ka, e = r.pe - ka;
        const double pre = r.pre;
        const int s = s0 + tid;
        load_data(r, nm);
        nm = load_meta(tb_after_next);
        __syncthreads();
        if (s < s1) {
            double sum = 0.0;
            for (int j = a; j < e; ++j) sum += prod[j];
            if constexpr (HAS_PRE) epi.seg_pre(s, sum, pre, racc);
            else epi.seg(s, sum, racc);
        }
        __syncthreads();
    };
    if (epi.done && *epi.done) return;   // launches queued behind a finished solve stop here
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    if constexpr (EpiHasPrepare<Epi>::value) epi.prepare();
    BigTileRegs ra, rb;
    const int b0 = blockIdx.x;
    int4 ma = load_meta(b0), mb = load_meta(b0 + G);
    load_data(ra, ma);       // HBM loads are in flight while x is copied into LDS
    load_data(rb, mb);
    ma = load_meta(b0 + 2 * G);
    mb = load_meta(b0 + 3 * G);
    {   // copy x into LDS: all (<= 12) loads of a thread are issued before the first is used
        constexpr int XR = (LSQ_LDS_X_MAX + LSQ_BIG_NT - 1) / LSQ_BIG_NT;
        double xr[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j) xr[j] = x[min(tid + j * LSQ_BIG_NT, nx - 1)];
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (tid + j * LSQ_BIG_NT < nx) xl[tid + j * LSQ_BIG_NT] = xr[j];
    }
    __syncthreads();
    int b = b0;
    for (; b + G < nbig; b += 2 * G) {
        step(ra, ma, b + 4 * G);
        step(rb, mb, b + 5 * G);
    }
    if (b < nbig) step(ra, ma, b + 4 * G);
    // side work (e.g. the damped rows of LSMR) is laid out for LSQ_NT-thread blocks
    for (int e = blockIdx.x; e < epi.extra_blocks; e += G)
        if (tid < LSQ_NT) epi.extra(e, racc);
    finish_block_nt<LSQ_BIG_NT>(epi, racc, sh);
}

// J'*y with the gathered m-vector staged window by window in LDS.  A gather y[row] from global
// memory costs a 128-byte L1 line fill for 8 useful bytes, and for an 8 MB y that fill traffic
// (not HBM) bounds the product.  Rows are cut into windows of `rw` rows (<= 4096: 32 KiB of y);
// one persistent 1024-thread workgroup owns a window at a time: it copies y[window] into LDS with
// coalesced loads, then streams the window's entries (column-sorted, big tiles of <= 8189 nnz and
// <= 2048 (window, column) segments) exactly like k_seg_stream_lds: 16-byte val/idx loads two
// tiles ahead, products against the LDS copy, per-segment sums in index order.  Segment sums go
// to part[w*n + j]; k_combine adds the windows of a column in index order (deterministic).
constexpr int LSQ_WIN_ROWS_MAX = 4096;
constexpr int LSQ_WIN_SEGS = 2 * LSQ_BIG_NT;
struct WinTileRegs {
    double2 v0[2], v1[2];
    int4 ci[2];
    int4 meta;
    int pa[2], pe[2];
};

// SQ: the same pass also forms the per-window column sums of squares (colsumabs2, utils.jl:146-151)
// from the values it already holds in registers; the partials are then laid out [w][2n]
// (dots | squares) and one k_combine over 2n "columns" finishes both.
template <bool IDX16, bool SQ>
__global__ void __launch_bounds__(LSQ_BIG_NT)
k_bcsc_lds(SegsDev S, const int4 *__restrict__ meta, const int *__restrict__ wtile, int nwin, int rw, int m, int n,
           const double *__restrict__ y, double *__restrict__ part, const int *done) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (done && *done) return;
    double *yl = smem;                       // rw doubles
    double *prod = smem + LSQ_WIN_ROWS_MAX;  // LSQ_BIG_WINDOW doubles
    double *prod2 = prod + LSQ_BIG_WINDOW;   // SQ only
    const int tid = threadIdx.x;
    const int kmax = (S.nnz + 3) & ~3;
    const int nbig = wtile[nwin];
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int t0 = wtile[w], t1 = wtile[w + 1];
        const int base = w * rw;
        auto load_meta = [&](int tb) {
            int4 mt = meta[tb < t1 ? tb : (nbig > 0 ? nbig - 1 : 0)];
            if (tb >= t1) mt.y = mt.x;
            return mt;
        };
        auto load_data = [&](WinTileRegs &r, int4 mt) {
            r.meta = mt;
            const int ka = mt.z & ~3;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = min(ka + c * 4 * LSQ_BIG_NT + 4 * tid, kmax);
                r.v0[c] = *reinterpret_cast<const double2 *>(S.val + k);
                r.v1[c] = *reinterpret_cast<const double2 *>(S.val + k + 2);
                r.ci[c] = load_idx4<IDX16>(S, k);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int s = min(mt.x + tid + q * LSQ_BIG_NT, S.nseg - 1);
                r.pa[q] = S.ptr[s];
                r.pe[q] = S.ptr[s + 1];
            }
        };
        auto step = [&](WinTileRegs &r, int4 &nm, int tb_after_next) {
            const int s0 = r.meta.x, s1 = r.meta.y, ka = r.meta.z & ~3;
            const int hi = rw - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // 16-bit indices are offsets inside the window; 32-bit ones are global rows, clamped
                // so that tile-slack entries of neighbouring windows stay inside the LDS copy
                double2 p0, p1;
                if constexpr (IDX16) {
                    p0.x = r.v0[c].x * yl[r.ci[c].x];
                    p0.y = r.v0[c].y * yl[r.ci[c].y];
                    p1.x = r.v1[c].x * yl[r.ci[c].z];
                    p1.y = r.v1[c].y * yl[r.ci[c].w];
                } else {
                    p0.x = r.v0[c].x * yl[min(max(r.ci[c].x - base, 0), hi)];
                    p0.y = r.v0[c].y * yl[min(max(r.ci[c].y - base, 0), hi)];
                    p1.x = r.v1[c].x * yl[min(max(r.ci[c].z - base, 0), hi)];
                    p1.y = r.v1[c].y * yl[min(max(r.ci[c].w - base, 0), hi)];
                }
                double2 *dst = reinterpret_cast<double2 *>(prod + c * 4 * LSQ_BIG_NT + 4 * tid);
                dst[0] = p0;
                dst[1] = p1;
                if constexpr (SQ) {
                    double2 *dq = reinterpret_cast<double2 *>(prod2 + c * 4 * LSQ_BIG_NT + 4 * tid);
                    dq[0] = make_double2(r.v0[c].x * r.v0[c].x, r.v0[c].y * r.v0[c].y);
                    dq[1] = make_double2(r.v1[c].x * r.v1[c].x, r.v1[c].y * r.v1[c].y);
                }
            }
            int a[2], e[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[q] = r.pa[q] - ka;
                e[q] = r.pe[q] - ka;
            }
            load_data(r, nm);
            nm = load_meta(tb_after_next);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int s = s0 + tid + q * LSQ_BIG_NT;
                if (s < s1) {
                    double sum = 0.0;
                    if constexpr (SQ) {
                        double sq = 0.0;
                        for (int j = a[q]; j < e[q]; ++j) {
                            sum += prod[j];
                            sq += prod2[j];
                        }
                        const size_t o = (size_t)w * 2 * n + (s - w * n);
                        part[o] = sum;
                        part[o + n] = sq;
                    } else {
                        for (int j = a[q]; j < e[q]; ++j) sum += prod[j];
                        part[s] = sum;
                    }
                }
            }
            __syncthreads();
        };
        WinTileRegs ra, rb;
        int4 ma = load_meta(t0), mb = load_meta(t0 + 1);
        load_data(ra, ma);
        load_data(rb, mb);
        ma = load_meta(t0 + 2);
        mb = load_meta(t0 + 3);
        {   // window of y -> LDS (all loads issued before the first use)
            constexpr int YR = LSQ_WIN_ROWS_MAX / LSQ_BIG_NT;
            const int rows = min(rw, m - base);
            double yr[YR];
#pragma unroll
            for (int j = 0; j < YR; ++j) yr[j] = y[base + min(tid + j * LSQ_BIG_NT, rows - 1)];
#pragma unroll
            for (int j = 0; j < YR; ++j)
                if (tid + j * LSQ_BIG_NT < rw) yl[tid + j * LSQ_BIG_NT] = (tid + j * LSQ_BIG_NT < rows) ? yr[j] : 0.0;
        }
        __syncthreads();
        int b = t0;
        for (; b + 1 < t1; b += 2) {
            step(ra, ma, b + 4);
            step(rb, mb, b + 5);
        }
        if (b < t1) step(ra, ma, b + 4);
        __syncthreads();  // the next window overwrites yl
    }
}

// pair-wise masked partial dot over [k0,k1) for a group of G lanes (lane index g in [0,G))
template <bool SQ, int G>
__device__ __forceinline__ double seg_partial(const SegsDev &S, const double *__restrict__ x, int k0,
                                              int k1, int g) {
    const int ka = k0 & ~1;
    double s0 = 0.0, s1 = 0.0;
    int k = ka + 2 * g;
    // two independent 16-byte streams in flight per lane
    for (; k + 2 * G < k1; k += 4 * G) {
        const double2 va = *reinterpret_cast<const double2 *>(S.val + k);
        const double2 vb = *reinterpret_cast<const double2 *>(S.val + k + 2 * G);
        int2 ca = make_int2(0, 0), cb = make_int2(0, 0);
        if (!SQ) {
            ca = *reinterpret_cast<const int2 *>(S.idx + k);
            cb = *reinterpret_cast<const int2 *>(S.idx + k + 2 * G);
        }
        double a0 = (k >= k0) ? prod1<SQ>(va.x, x, ca.x) : 0.0;
        double a1 = prod1<SQ>(va.y, x, ca.y);  // k+1 < k1 holds: k + 2G < k1
        double b0 = prod1<SQ>(vb.x, x, cb.x);
        double b1 = (k + 2 * G + 1 < k1) ? prod1<SQ>(vb.y, x, cb.y) : 0.0;
        s0 += a0 + a1;
        s1 += b0 + b1;
    }
    if (k < k1) {
        const double2 va = *reinterpret_cast<const double2 *>(S.val + k);
        int2 ca = make_int2(0, 0);
        if (!SQ) ca = *reinterpret_cast<const int2 *>(S.idx + k);
        double a0 = (k >= k0) ? prod1<SQ>(va.x, x, ca.x) : 0.0;
        double a1 = (k + 1 < k1) ? prod1<SQ>(va.y, x, ca.y) : 0.0;
        s0 += a0 + a1;
    }
    return s0 + s1;
}

template <class Epi, bool SQ>
__global__ void __launch_bounds__(LSQ_NT) k_seg_wave(SegsDev S, const double *__restrict__ x, Epi epi,
                                                      int nsegblocks) {
    __shared__ double sh[LSQ_NT / 64];
    if (epi.done && *epi.done) return;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nwork = nsegblocks + epi.extra_blocks;
    double racc = 0.0;
    for (int bb = blockIdx.x; bb < nwork; bb += gridDim.x) {
        if (bb >= nsegblocks) {
            epi.extra(bb - nsegblocks, racc);
            continue;
        }
        const int b = S.order ? S.order[bb] : bb;
        const int s = b * (LSQ_NT / 64) + w;
        if (s < S.nseg) {
            const int k0 = S.ptr[s], k1 = S.ptr[s + 1];
            double sum = wave_sum(seg_partial<SQ, 64>(S, x, k0, k1, lane));
            if (lane == 0) epi.seg(s, sum, racc);
        }
    }
    finish_block(epi, racc, sh);
}

template <class Epi, bool SQ>
__global__ void __launch_bounds__(LSQ_NT) k_seg_block(SegsDev S, const double *__restrict__ x, Epi epi) {
    __shared__ double sh[LSQ_NT / 64];
    if (epi.done && *epi.done) return;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    const int nwork = S.nseg + epi.extra_blocks;
    double racc = 0.0;
    for (int b = blockIdx.x; b < nwork; b += gridDim.x) {
        if (b >= S.nseg) {
            epi.extra(b - S.nseg, racc);
            continue;
        }
        const int k0 = S.ptr[b], k1 = S.ptr[b + 1];
        double sum = block_sum<LSQ_NT>(seg_partial<SQ, LSQ_NT>(S, x, k0, k1, threadIdx.x), sh);
        if (threadIdx.x == 0) epi.seg(b, sum, racc);
    }
    finish_block(epi, racc, sh);
}

static inline SegsDev segs_dev(const LsqSegs &s) {
    return SegsDev{s.d_ptr, s.d_idx, s.d_val, s.d_tiles, s.d_idx16, s.d_order, s.nseg, s.ntiles, (int)s.nnz};
}

// Launch the plan chosen for `segs`.  Work items = segment blocks + epi.extra_blocks; the grid is
// capped (blocks loop over work items) so the partial-sum buffer always suffices.
constexpr int LSQ_MAX_GRID = 16384;
template <bool SQ, class Epi>
static inline int launch_segs(lsq_ctx *ctx, const LsqSegs &segs, const double *x, const Epi &epi) {
    SegsDev S = segs_dev(segs);
    static_assert(LSQ_MAX_GRID <= LSQ_MAX_PARTIALS, "partials buffer too small");
    // reducing kernels keep the grid at <= 2048 blocks (8 per CU): fewer arrival tickets
    constexpr int maxg = Epi::REDUCE ? 2048 : LSQ_MAX_GRID;
    auto cap = [](long long w) { return (int)(w > maxg ? maxg : w); };
    switch (segs.plan) {
    case LSQ_PLAN_STREAM: {
        if (!SQ && segs.nbig > 0 && segs.nx <= LSQ_LDS_X_MAX) {
            // LDS-staged gather vector: one persistent 1024-thread workgroup per CU
            const int nxpad = (segs.nx + 1) & ~1;
            const size_t lds = (size_t)(nxpad + LSQ_BIG_WINDOW) * sizeof(double);
            const bool i16 = segs.d_idx16 != nullptr;
            auto kern = i16 ? k_seg_stream_lds<Epi, true> : k_seg_stream_lds<Epi, false>;
            LSQ_TRY(lsq_set_lds(ctx, (const void *)kern, (LSQ_LDS_X_MAX + LSQ_BIG_WINDOW) * sizeof(double)));
            int grid = std::max(1, std::min(segs.nbig, ctx->num_cus));
            hipEvent_t e0, e1;
            if (lsq_prof_take(ctx, &e0, &e1))
                LSQ_LAUNCH_TIMED(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, ctx->stream, e0, e1, 0, S,
                                      (const int4 *)segs.d_big, segs.nbig, x, segs.nx, nxpad, epi);
            else
                LSQ_LAUNCH(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, ctx->stream, S, (const int4 *)segs.d_big,
                                   segs.nbig, x, segs.nx, nxpad, epi);
            break;
        }
        int grid = cap((long long)segs.ntiles + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_seg_stream<Epi, SQ>), dim3(grid), dim3(LSQ_NT), 0, ctx->stream, S, x, epi);
        break;
    }
    case LSQ_PLAN_WAVE: {
        int nb = lsq_div_up(segs.nseg, LSQ_NT / 64);
        int grid = cap((long long)nb + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_seg_wave<Epi, SQ>), dim3(grid), dim3(LSQ_NT), 0, ctx->stream, S, x, epi, nb);
        break;
    }
    default: {
        int grid = cap((long long)segs.nseg + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_seg_block<Epi, SQ>), dim3(grid), dim3(LSQ_NT), 0, ctx->stream, S, x, epi);
        break;
    }
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// dense column-major products with the same epilogue interface (dense Jacobian + LSMR, and the
// GEMVs of the optimizer loops: levenberg_marquardt.jl:102,114; dogleg.jl:99,109,171)
// ---------------------------------------------------------------------------------------------
// y = J x: one thread per row; lanes read consecutive rows of a column => coalesced.  Columns are
// split in 4 interleaved accumulators to keep 4 loads in flight.
template <class Epi>
__global__ void __launch_bounds__(LSQ_NT) k_dense_n(const double *__restrict__ A, int m, int n,
                                                     const double *__restrict__ x, Epi epi, int nrowblocks) {
    __shared__ double sh[LSQ_NT / 64];
    if (epi.done && *epi.done) return;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    const int nwork = nrowblocks + epi.extra_blocks;
    double racc = 0.0;
    for (int b = blockIdx.x; b < nwork; b += gridDim.x) {
        if (b >= nrowblocks) {
            epi.extra(b - nrowblocks, racc);
            continue;
        }
        const int i = b * LSQ_NT + threadIdx.x;
        if (i < m) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const double *p = A + i;
            int j = 0;
            for (; j + 3 < n; j += 4) {
                a0 += p[(size_t)j * m] * x[j];
                a1 += p[(size_t)(j + 1) * m] * x[j + 1];
                a2 += p[(size_t)(j + 2) * m] * x[j + 2];
                a3 += p[(size_t)(j + 3) * m] * x[j + 3];
            }
            for (; j < n; ++j) a0 += p[(size_t)j * m] * x[j];
            epi.seg(i, (a0 + a1) + (a2 + a3), racc);
        }
    }
    finish_block(epi, racc, sh);
}

// The same for a matrix whose row blocks alone do not fill the device (m / 256 workgroups: 64 at C3, 16 at C2 -- the
// one-thread-per-row kernel above then streams at 0.4-1 TB/s): a block takes (row block rb, column chunk cw) and writes
// its partial dot products to part[cw * m + i]; k_combine adds the chunks in index order and runs the caller's epilogue.
template <int = 0>
__global__ void __launch_bounds__(LSQ_NT) k_dense_n_win(const double *__restrict__ A, int m, int n, const double *__restrict__ x,
                                                         int ccols, int nrb, double *__restrict__ part, const int *done) {
    if (done && *done) return;
    const int rb = blockIdx.x % nrb, cw = blockIdx.x / nrb;
    const int i = rb * LSQ_NT + threadIdx.x;
    const int j0 = cw * ccols, j1 = min(n, j0 + ccols);
    if (i >= m) return;
    const double *p = A + i;
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.0;
    int j = j0;
    for (; j + 7 < j1; j += 8) {      // eight loads in flight per thread
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(j + u) * m];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += t[u] * x[j + u];
    }
    for (; j < j1; ++j) a[0] += p[(size_t)j * m] * x[j];
    part[(size_t)cw * m + i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
// column chunks for J*x of a dense m x n matrix (0: one thread per row fills the device by itself)
static inline int lsq_dense_n_chunks(const lsq_ctx *c, int m, int n) {
    const int nrb = (m + LSQ_NT - 1) / LSQ_NT;
    if (n < 64 || nrb >= 4 * c->num_cus) return 0;
    int nch = std::min((n + 31) / 32, (8 * c->num_cus + nrb - 1) / nrb);
    return nch > 1 ? nch : 0;
}

// x = J' y (SQ: column sums of squares): one block per column, contiguous reads.
template <class Epi, bool SQ>
__global__ void __launch_bounds__(LSQ_NT) k_dense_t(const double *__restrict__ A, int m, int n,
                                                     const double *__restrict__ y, Epi epi) {
    __shared__ double sh[LSQ_NT / 64];
    if (epi.done && *epi.done) return;
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    const int nwork = n + epi.extra_blocks;
    double racc = 0.0;
    for (int b = blockIdx.x; b < nwork; b += gridDim.x) {
        if (b >= n) {
            epi.extra(b - n, racc);
            continue;
        }
        const double *col = A + (size_t)b * m;
        double a0 = 0.0, a1 = 0.0;
        int i = threadIdx.x;
        for (; i + LSQ_NT < m; i += 2 * LSQ_NT) {
            double c0 = col[i], c1 = col[i + LSQ_NT];
            a0 += SQ ? c0 * c0 : c0 * y[i];
            a1 += SQ ? c1 * c1 : c1 * y[i + LSQ_NT];
        }
        if (i < m) {
            double c0 = col[i];
            a0 += SQ ? c0 * c0 : c0 * y[i];
        }
        double sum = block_sum<LSQ_NT>(a0 + a1, sh);
        if (threadIdx.x == 0) epi.seg(b, sum, racc);
    }
    finish_block(epi, racc, sh);
}

// The same for a dense matrix with FEW columns (fewer than the device has CUs): one block per column would leave most
// of the chip idle, so a block takes (row window w, column j) and writes its partial to part[w * n + j]; k_combine
// adds the windows in index order and runs the caller's epilogue.
template <bool SQ>
__global__ void __launch_bounds__(LSQ_NT) k_dense_t_win(const double *__restrict__ A, int m, int n, const double *__restrict__ y,
                                                         int wrows, double *__restrict__ part, const int *done) {
    __shared__ double sh[LSQ_NT / 64];
    if (done && *done) return;
    const int j = blockIdx.x % n, w = blockIdx.x / n;
    const int r0 = w * wrows, rows = min(wrows, m - r0);
    const double *col = A + (size_t)j * m + r0;
    const double *yy = SQ ? nullptr : y + r0;
    double a0 = 0.0, a1 = 0.0;
    int i = threadIdx.x;
    for (; i + LSQ_NT < rows; i += 2 * LSQ_NT) {
        double c0 = col[i], c1 = col[i + LSQ_NT];
        a0 += SQ ? c0 * c0 : c0 * yy[i];
        a1 += SQ ? c1 * c1 : c1 * yy[i + LSQ_NT];
    }
    if (i < rows) {
        double c0 = col[i];
        a0 += SQ ? c0 * c0 : c0 * yy[i];
    }
    double sum = block_sum<LSQ_NT>(a0 + a1, sh);
    if (threadIdx.x == 0) part[(size_t)w * n + j] = sum;
}
// window count for a dense m x n matrix (0: one block per column is fine)

// Per-window column sums -> d_bpart[w*n + j] (first pass of the window-blocked J'*y)
struct EpiPart {
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *part;
    double *partials;
    unsigned *counter;
    __device__ void seg(int s, double dot, double &) const { part[s] = dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

// second pass: column j = sum over windows, then the caller's epilogue.  A block owns 32 columns;
// 8 thread groups each add every 8th window in index order (8 loads in flight per thread), and
// the 8 group sums are added in index order -- a fixed association, hence deterministic.
constexpr int LSQ_CMB_COLS = 32;
constexpr int LSQ_CMB_GROUPS = LSQ_NT / LSQ_CMB_COLS;
// cscale (or null): every combined dot is multiplied by cscale[j] before the epilogue sees it -- J'y = s .* (V'y) for a
// column-scaled Jacobian J = V diag(s) (lsq_mat::d_colscale).
// has_col_prefetch: the epilogue's own per-column operands (EpiV: P, sqrt(damp), u~x, v) are requested with the first round of
// partials instead of after the sums -- this kernel is two memory round trips long, not three (round 6)
template <class E, class = void>
struct EpiHasColPrefetch : std::false_type {};
template <class E>
struct EpiHasColPrefetch<E, std::void_t<typename E::has_col_prefetch>> : std::true_type {};
template <class Epi>
__global__ void __launch_bounds__(LSQ_NT) k_combine(const double *__restrict__ part, int n, int nwin, Epi epi,
                                                     int ncolblocks, const double *__restrict__ cscale = nullptr) {
    __shared__ double sh[LSQ_NT / 64];
    __shared__ double grp[LSQ_CMB_GROUPS][LSQ_CMB_COLS + 1];
    const int nwork = ncolblocks + epi.extra_blocks;
    const int cidx = threadIdx.x % LSQ_CMB_COLS, g = threadIdx.x / LSQ_CMB_COLS;
    // latency-bound (10 MB over 313 workgroups): the `done` flag of a finished solve and the first eight window partials of
    // this thread are fetched BEFORE the epilogue's prologue (which waits for the previous kernel's partial sums)
    const int dflag = epi.done ? *epi.done : 0;
    double t0[8];
    const bool pre0 = (int)blockIdx.x < ncolblocks && (int)blockIdx.x * LSQ_CMB_COLS + cidx < n && g + 7 * LSQ_CMB_GROUPS < nwin;
    if (pre0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t0[q] = part[(size_t)(g + q * LSQ_CMB_GROUPS) * n + blockIdx.x * LSQ_CMB_COLS + cidx];
    }
    if constexpr (EpiHasColPrefetch<Epi>::value) {
        if (g == 0 && (int)blockIdx.x < ncolblocks && (int)blockIdx.x * LSQ_CMB_COLS + cidx < n)
            epi.col_prefetch((int)blockIdx.x * LSQ_CMB_COLS + cidx);
    }
    if constexpr (EpiHasBlockPrepare<Epi>::value) epi.block_prepare();
    if (dflag) return;   // launches queued behind a finished solve stop here
    double racc = 0.0;
    for (int b = blockIdx.x; b < nwork; b += gridDim.x) {
        if (b >= ncolblocks) {
            epi.extra(b - ncolblocks, racc);
            continue;
        }
        const int j = b * LSQ_CMB_COLS + cidx;
        double acc = 0.0;
        const double cs0 = (cscale && g == 0 && j < n) ? cscale[j] : 1.0;   // (in flight with the partials)
        if (j < n) {
            int w = g;
            if (pre0 && b == (int)blockIdx.x) {       // (same additions in the same order)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t0[q];
                w += 8 * LSQ_CMB_GROUPS;
            }
            for (; w + 7 * LSQ_CMB_GROUPS < nwin; w += 8 * LSQ_CMB_GROUPS) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = part[(size_t)(w + q * LSQ_CMB_GROUPS) * n + j];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            for (; w < nwin; w += LSQ_CMB_GROUPS) acc += part[(size_t)w * n + j];
        }
        grp[g][cidx] = acc;
        __syncthreads();
        if (g == 0 && j < n) {
            double dot = 0.0;
#pragma unroll
            for (int q = 0; q < LSQ_CMB_GROUPS; ++q) dot += grp[q][cidx];
            if (cscale) dot *= cs0;
            epi.seg(j, dot, racc);
        }
        __syncthreads();
    }
    finish_block(epi, racc, sh);
}

#include "lsq_sell.h"

static inline SellDev sell_dev(const LsqSell &s, const double *val = nullptr) {
    return SellDev{s.d_smeta, s.d_info, s.d_idx16, val ? val : s.d_val, s.nblocks, s.spw};
}

// (stored values) * diag(xscale) * x over the sliced rows; xscale = J->d_colscale gives J*x of a column-scaled Jacobian,
// nullptr the product with the stored values V themselves (what a model r = V phi(x) - b needs)
// `val`: another value array in the same layout (a model's A beside a multiplied-out J) instead of J's own
template <class Epi>
static inline int launch_sell_rows(lsq_mat *J, const double *xscale, const double *x, const Epi &epi, const double *val = nullptr) {
    lsq_ctx *c = J->ctx;
    const LsqSell &S = J->srows;
    if (S.ncw > 1) {   // n > LSQ_LDS_X_MAX: one column window of x in LDS at a time
        const size_t lds = (size_t)(S.cwidth + LSQ_SELL_ROWS_MAX) * sizeof(double);
        auto kern = k_sell_rows_wide<Epi>;
        LSQ_TRY(lsq_set_lds(c, (const void *)kern, (LSQ_SELL_WIDE_X_MAX + 2 + LSQ_SELL_ROWS_MAX) * sizeof(double)));
        if (xscale) {   // column-scaled handle: the gather vector s .* x once, not once per row block and window
            LSQ_LAUNCH(k_sell_vmul<0>, dim3(std::min(lsq_div_up(J->n, LSQ_NT), c->num_cus * 4)), dim3(LSQ_NT), 0, c->stream,
                               J->n, x, xscale, S.d_sx);
            x = S.d_sx;
        }
        const int grid = std::max(1, std::min(S.nblocks / S.ncw, c->num_cus));
        hipEvent_t e0, e1;
        if (lsq_prof_take(c, &e0, &e1))
            LSQ_LAUNCH_TIMED(kern, dim3(grid), dim3(LSQ_WIDE_NT), lds, c->stream, e0, e1, 0, sell_dev(S, val), S.wrows,
                                  J->m, S.ncw, S.cwidth, x, J->n, epi);
        else
            LSQ_LAUNCH(kern, dim3(grid), dim3(LSQ_WIDE_NT), lds, c->stream, sell_dev(S, val), S.wrows, J->m, S.ncw,
                               S.cwidth, x, J->n, epi);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    const int nxpad = (J->n + 1) & ~1;
    const size_t lds = (size_t)(nxpad + LSQ_SELL_ROWS_MAX) * sizeof(double);
    auto kern = k_sell_rows<Epi>;
    LSQ_TRY(lsq_set_lds(c, (const void *)kern, (LSQ_LDS_X_MAX + LSQ_SELL_ROWS_MAX) * sizeof(double)));
    const int grid = std::max(1, std::min(S.nblocks, c->num_cus));
    hipEvent_t e0, e1;
    if (lsq_prof_take(c, &e0, &e1))
        LSQ_LAUNCH_TIMED(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, e0, e1, 0, sell_dev(S, val), S.wrows,
                              J->m, x, xscale, J->n, nxpad, epi);
    else
        LSQ_LAUNCH(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, sell_dev(S, val), S.wrows, J->m, x, xscale,
                           J->n, nxpad, epi);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// first pass of J'*y over the sliced columns: per gather-window partials into J->scols.d_part
template <bool SQ>
static inline int launch_sell_cols(lsq_mat *J, const double *y, const int *done) {
    lsq_ctx *c = J->ctx;
    const LsqSell &S = J->scols;
    const size_t lds = (size_t)(LSQ_SELL_GROWS_MAX + LSQ_SELL_CCOLS_MAX) * sizeof(double);
    auto kern = k_sell_cols<SQ>;
    LSQ_TRY(lsq_set_lds(c, (const void *)kern, lds));
    const int grid = std::max(1, std::min(S.nblocks, c->num_cus));
    hipEvent_t e0, e1;
    if (lsq_prof_take(c, &e0, &e1))
        LSQ_LAUNCH_TIMED(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, e0, e1, 0, sell_dev(S), S.ncb, S.ccols,
                              S.grows, J->m, J->n, y, S.d_part, done);
    else
        LSQ_LAUNCH(kern, dim3(grid), dim3(LSQ_BIG_NT), lds, c->stream, sell_dev(S), S.ncb, S.ccols, S.grows,
                           J->m, J->n, y, S.d_part, done);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// One entry point for "dot every row (trans=0) / column (trans=1) of J with x, then epilogue".
template <class Epi>
static inline int launch_product(lsq_mat *J, int trans, const double *x, const Epi &epi) {
    lsq_ctx *c = J->ctx;
    constexpr int maxg = Epi::REDUCE ? 2048 : LSQ_MAX_GRID;
    auto cap = [](long long w) { return (int)(w > maxg ? maxg : w); };
    if (J->kind == LSQ_MAT_OP) {
        // matrix-free operator: the host callback produces the plain product, then the epilogue runs over it
        // (k_combine with a single "window" is exactly "apply the epilogue to a vector of dot products")
        LSQ_HIP(hipStreamSynchronize(c->stream));
        if (J->op_mul(trans, x, J->d_optmp, J->op_user) != 0) {
            lsq_set_error("operator mul callback reported failure");
            return LSQ_ECALLBACK;
        }
        const int len = trans ? J->n : J->m;
        int nb = lsq_div_up(len, LSQ_CMB_COLS);
        int grid = cap((long long)nb + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_combine<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_optmp, len, 1, epi, nb);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    if (J->kind == LSQ_MAT_CSC) {
        if (!trans) {
            LSQ_TRY(lsq_ensure_csr(J));
            if (J->srows.active) return launch_sell_rows(J, J->d_colscale, x, epi);
            return launch_segs<false>(c, J->csr, x, epi);
        }
        if (J->scols.active) {
            LSQ_TRY(lsq_ensure_csr(J));
            LSQ_TRY(launch_sell_cols<false>(J, x, epi.done));
            int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
            int grid = cap((long long)nb + epi.extra_blocks);
            LSQ_LAUNCH((k_combine<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->scols.d_part, J->n,
                               J->scols.ngw, epi, nb, J->d_colscale);
            LSQ_HIP(hipGetLastError());
            return LSQ_OK;
        }
        if (J->nwin > 1) {
            LSQ_TRY(lsq_ensure_csr(J));
            if (J->bcsc.plan == LSQ_PLAN_LDSWIN) {
                const size_t lds = (size_t)(LSQ_WIN_ROWS_MAX + LSQ_BIG_WINDOW) * sizeof(double);
                const bool i16 = J->bcsc.d_idx16 != nullptr;
                auto kern = i16 ? k_bcsc_lds<true, false> : k_bcsc_lds<false, false>;
                LSQ_TRY(lsq_set_lds(c, (const void *)kern, lds));
                int g2 = std::max(1, std::min(J->bcsc.nwin, c->num_cus));
                hipEvent_t e0, e1;
                if (lsq_prof_take(c, &e0, &e1))
                    LSQ_LAUNCH_TIMED(kern, dim3(g2), dim3(LSQ_BIG_NT), lds, c->stream, e0, e1, 0, segs_dev(J->bcsc),
                                          (const int4 *)J->bcsc.d_big, J->bcsc.d_wtile, J->bcsc.nwin, J->bcsc.rw, J->m,
                                          J->n, x, J->d_bpart, epi.done);
                else
                    LSQ_LAUNCH(kern, dim3(g2), dim3(LSQ_BIG_NT), lds, c->stream, segs_dev(J->bcsc),
                                       (const int4 *)J->bcsc.d_big, J->bcsc.d_wtile, J->bcsc.nwin, J->bcsc.rw, J->m,
                                       J->n, x, J->d_bpart, epi.done);
            } else {
                EpiPart ep{epi.done, 0, J->d_bpart, nullptr, nullptr};
                LSQ_TRY(launch_segs<false>(c, J->bcsc, x, ep));
            }
            int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
            int grid = cap((long long)nb + epi.extra_blocks);
            LSQ_LAUNCH((k_combine<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_bpart, J->n,
                               J->nwin, epi, nb);
            LSQ_HIP(hipGetLastError());
            return LSQ_OK;
        }
        LSQ_TRY(lsq_ensure_csc(J));
        return launch_segs<false>(c, J->csc, x, epi);
    }
    if (!trans && lsq_dense_n_chunks(c, J->m, J->n)) {
        const int nch = lsq_dense_n_chunks(c, J->m, J->n), nrb = lsq_div_up(J->m, LSQ_NT);
        const int ccols = ((J->n + nch - 1) / nch + 7) / 8 * 8, nchunks = (J->n + ccols - 1) / ccols;
        LSQ_TRY(lsq_dense_part_elems(J, (size_t)nchunks * J->m));
        LSQ_LAUNCH(k_dense_n_win<0>, dim3(nrb * nchunks), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x, ccols, nrb,
                           J->d_dpart, epi.done);
        int nb = lsq_div_up(J->m, LSQ_CMB_COLS);
        int grid = cap((long long)nb + epi.extra_blocks);
        LSQ_LAUNCH((k_combine<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dpart, J->m, nchunks, epi, nb);
    } else if (!trans) {
        int nb = lsq_div_up(J->m, LSQ_NT);
        int grid = cap((long long)nb + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_dense_n<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m,
                               J->n, x, epi, nb);
    } else if (const int nwin = lsq_dense_t_windows(c, J->m, J->n)) {
        LSQ_TRY(lsq_dense_part(J, nwin));
        const int wrows = ((J->m + nwin - 1) / nwin + 3) / 4 * 4;
        LSQ_LAUNCH((k_dense_t_win<false>), dim3(nwin * J->n), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, x,
                           wrows, J->d_dpart, epi.done);
        int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
        int grid = cap((long long)nb + epi.extra_blocks);
        LSQ_LAUNCH((k_combine<Epi>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dpart, J->n, nwin, epi, nb);
    } else {
        int grid = cap((long long)J->n + epi.extra_blocks);
        if (grid > 0)
            LSQ_LAUNCH((k_dense_t<Epi, false>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense,
                               J->m, J->n, x, epi);
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
