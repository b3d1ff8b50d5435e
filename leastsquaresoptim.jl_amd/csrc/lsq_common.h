// Internal declarations shared by the HIP translation units of liblsqhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "../../include/lsqhip.h"

void lsq_set_error(const char *fmt, ...);

#define LSQ_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            lsq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                       \
            return LSQ_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define LSQ_TRY(call)                 \
    do {                              \
        int s__ = (call);             \
        if (s__ != LSQ_OK) return s__; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Debug modes (lsq_debug_set in include/lsqhip.h; environment LSQ_DEBUG_LAUNCH_JITTER=<us>, LSQ_DEBUG_SERIAL=<0|1|2>,
// read when the first context is created).  Every kernel launch of the library goes through LSQ_LAUNCH:
//   jitter  a random host stall (uniform in [0, us], before one launch in four; one in 64 is 20x longer) in front of the
//           launch -- what a busy host does to the queue once per thousand launches, on every solve.  Any hand-off between
//           kernels that holds only "because the next launch follows at once" fails under it.
//   serial  1: the launch is waited for before the host goes on (nothing overlaps: no side-stream concurrency, no
//           look-ahead, no speculation behind unfinished kernels) -- same kernels, same arithmetic, so the results
//           must be bit-identical to the normal mode;  2: in addition the paths built on in-kernel exchanges between
//           workgroups (QR slab exchange, pipelined triangular solves, one-launch Cholesky) are off -- other kernels,
//           compared to a tolerance.
// ---------------------------------------------------------------------------------------------
extern int lsq_dbg_jitter_us;
extern int lsq_dbg_serial;
void lsq_dbg_init();
void lsq_dbg_stall();
#define LSQ_LAUNCH(kern, grid, block, lds, stream, ...)                         \
    do {                                                                        \
        if (lsq_dbg_jitter_us > 0) lsq_dbg_stall();                             \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);        \
        if (lsq_dbg_serial) (void)hipStreamSynchronize(stream);                 \
    } while (0)
#define LSQ_LAUNCH_TIMED(kern, grid, block, lds, stream, e0, e1, ...)           \
    do {                                                                        \
        if (lsq_dbg_jitter_us > 0) lsq_dbg_stall();                             \
        hipExtLaunchKernelGGL(kern, grid, block, lds, stream, e0, e1, __VA_ARGS__); \
        if (lsq_dbg_serial) (void)hipStreamSynchronize(stream);                 \
    } while (0)

// roctx ranges at the C-ABI entries: a trace taken with `rocprofv3 --marker-trace --kernel-trace` can be read per
// ldiv! / optimize! / product.  The marker library is bound at run time (librocprofiler-sdk-roctx.so, else libroctx64.so) and
// only when LSQ_ROCTX=1 is set: without it a range is one predictable branch.
struct LsqRange {
    bool on;
    explicit LsqRange(const char *name);
    ~LsqRange();
};
#define LSQ_RANGE(name) LsqRange lsq_range__(name)

constexpr int LSQ_NT = 256;             // threads per block for streaming kernels (4 waves)
constexpr int LSQ_MAX_PARTIALS = 1 << 16;
constexpr int LSQ_NSLOTS = 64;          // device scalar slots / reduction counter slots

// Host-visible mailbox (pinned, coherent): the device publishes inner-loop progress here so the
// host never calls hipStreamSynchronize inside LSMR.
struct LsqMailbox {
    volatile int iter;
    volatile int istop;
    volatile int done;
    volatile int seq;
    // hints beside the progress word (written in front of it): the two stopping quantities that end LM's inner solves,
    // test1 = |r|/|b| and test2 = |A'r|/(|A||r|) (lsmr.jl:207-208) of the iteration the word announces.  The host only
    // PREDICTS with them (where to queue the caller's tail, lsq_lsmr_solve); a stale pair costs a wrong guess, nothing else.
    volatile double test1, test2;
};

struct lsq_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    double *d_slots;      // LSQ_NSLOTS device scalars (results of reductions)
    double *h_slots;      // pinned + mapped host mirror: [0..NSLOTS) values, [NSLOTS] = sequence word
    double *d_hslots;     // device address of h_slots
    unsigned long long slot_seq = 0;
    // two high-priority helper streams shared by every solver of the context (lsq_ctx_helper_stream; the dense QR's k_cqr_top
    // and its look-ahead passes).  Per-SOLVER streams were a trap: HIP multiplexes streams onto four hardware queues, so with
    // a second QR solver alive the fifth stream shared a queue with the main one and every stage-1 kernel waited behind
    // k_cqr_top (Dogleg+QR 7.7 -> 17 ms per outer iteration, measured in round 5).
    hipStream_t helper_stream[2] = {nullptr, nullptr};
    double *d_partials;   // LSQ_MAX_PARTIALS block partials
    unsigned *d_counters; // LSQ_NSLOTS arrival counters (zero between kernels)
    LsqMailbox *h_mail;   // pinned + mapped
    LsqMailbox *d_mail;   // device address of h_mail
    int num_cus;
    unsigned mail_epoch;  // bumps per inner solve; tags mailbox words
    void *workspace = nullptr;  // cached LsqWorkspace of lsq_optimize (lsq_optimize.hip)
    // one-shot host work to run while the device is busy (the sharded loops' scalar exchange): the LSMR
    // driver calls it once its look-ahead window is full, i.e. with >= 2 iterations of device work queued
    int (*idle_hook)(void *) = nullptr;
    void *idle_user = nullptr;
    int idle_status = 0;
    // optional HIP-event instrumentation (lsq_prof_begin/end)
    int prof_max = 0;
    int prof_pending = -1;               // kernel id whose NEXT launch should carry dispatch timestamps
    int prof_kernels = 3;                // bit k: instrument kernel k (a timed launch costs ~9 us of gaps)
    int prof_stride = 1, prof_tick = 0;  // time every prof_stride-th armed launch
    std::vector<hipEvent_t> prof_ev[2];  // start/stop pairs per kernel id
    std::vector<hipEvent_t> prof_pool;   // events created by lsq_prof_begin, so that a timed launch creates nothing
    // kernels whose dynamic-LDS limit has been raised on THIS context's device (function attributes may be per device)
    std::unordered_map<const void *, size_t> lds_cfg;
    // give-ups of the fast paths that rely on co-resident workgroups, summed over every solver of this context
    // (LsqFallback in lsq_solver.h; index: 0 one-launch Cholesky, 1 pipelined triangular solves, 2 QR slab exchange /
    // pipelined certified solve, 3 CholeskyQR2 panel breakdowns)
    int fallback_giveups[4] = {0, 0, 0, 0};
    // LSMR solves whose caller's next kernels were enqueued behind a guessed last iteration (LsmrTail): {guesses, wrong ones}
    long long tail_spec[2] = {0, 0};
    hipStream_t occupy_stream = nullptr;   // lsq_bench_occupy
    // uploads that overlap compute (lsq_mat_set_values_async): created on first use
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
};

// raise a kernel's dynamic-LDS limit once per context (= per device)
static inline int lsq_set_lds(lsq_ctx *c, const void *kern, size_t bytes) {
    auto it = c->lds_cfg.find(kern);
    if (it != c->lds_cfg.end() && it->second >= bytes) return LSQ_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return LSQ_EHIP;
    }
    c->lds_cfg[kern] = bytes;
    return LSQ_OK;
}

// record a start (phase 0) / stop (phase 1) event for kernel `kid` if instrumentation is on
// phase 0 arms the instrumentation for kernel `kid`: a launch site that supports it (the LDS-staged
// kernels) then launches with hipExtLaunchKernelGGL(start, stop), whose events carry the DISPATCH's
// own begin/end timestamps (what rocprofv3 reports) -- no marker packets between kernels.  Launch
// sites that do not support it are bracketed with ordinary event records instead (phase 1).
static inline void lsq_prof_mark(lsq_ctx *c, int kid, int phase) {
    if (c->prof_max <= 0) return;
    auto &v = c->prof_ev[kid];
    if (phase == 0) {
        if ((int)v.size() >= 2 * c->prof_max) return;
        if ((c->prof_tick++ % c->prof_stride) != 0) return;
        c->prof_pending = kid;
        return;
    }
    if (c->prof_pending != kid) return;   // consumed by an ext launch (or never armed)
    c->prof_pending = -1;                 // not consumed: nothing was timed for this launch
}
// used by launch sites: returns true and fills start/stop when the launch should be timed
static inline bool lsq_prof_take(lsq_ctx *c, hipEvent_t *start, hipEvent_t *stop) {
    if (c->prof_max <= 0 || c->prof_pending < 0) return false;
    if (c->prof_pool.size() >= 2) {          // (hipEventCreate costs microseconds of host time: not inside the solve)
        *start = c->prof_pool.back(); c->prof_pool.pop_back();
        *stop = c->prof_pool.back(); c->prof_pool.pop_back();
    } else {
        if (hipEventCreate(start) != hipSuccess) return false;
        if (hipEventCreate(stop) != hipSuccess) { hipEventDestroy(*start); return false; }
    }
    auto &v = c->prof_ev[c->prof_pending];
    v.push_back(*start);
    v.push_back(*stop);
    c->prof_pending = -1;
    return true;
}

// ---------------------------------------------------------------------------------------------
// sparse / dense matrix handle
// ---------------------------------------------------------------------------------------------
enum { LSQ_MAT_DENSE = 0, LSQ_MAT_CSC = 1, LSQ_MAT_OP = 2 };
enum { LSQ_PLAN_STREAM = 0, LSQ_PLAN_WAVE = 1, LSQ_PLAN_BLOCK = 2, LSQ_PLAN_LDSWIN = 3 };

// One direction of a sparse product: segments (rows for J*x via the CSR mirror, columns for
// J'*y via CSC) with the launch plan chosen once per pattern from the segment-length profile.
struct LsqSegs {
    int nseg = 0;          // number of segments (m for CSR, n for CSC)
    long long nnz = 0;
    int *d_ptr = nullptr;  // nseg+1
    int *d_idx = nullptr;  // nnz (gather index)
    unsigned short *d_idx16 = nullptr;  // same indices in 16 bits when they fit (LDS-staged kernels: 10 B/nnz)
    unsigned short *d_col16 = nullptr;  // window-blocked CSC: column of every entry (device-side g! scaling)
    double *d_val = nullptr;
    int plan = LSQ_PLAN_STREAM;
    int ntiles = 0;        // stream plan: number of tiles
    int *d_tiles = nullptr; // ntiles+1 segment boundaries of the tiles
    int *d_order = nullptr; // optional permutation of the work items (XCD-aware placement)
    int nx = 0;            // length of the gathered vector (n for CSR rows, m for CSC columns)
    int rw = 0;            // window-blocked CSC: rows per window
    int nwin = 0;          // window-blocked CSC: number of windows (segments = nwin * n)
    int *d_wtile = nullptr; // LDS-window plan: nwin+1 big-tile ranges of the windows
    int nbig = 0;          // stream plan, LDS-staged variant: number of big tiles
    int *d_big = nullptr;  // nbig x int4 {s0, s1, k0, k1} (<= 8189 nnz, <= 1024 segments each)
};

// Sliced layout of one product direction (lsq_sell.h): replaces the CSR mirror (J*x) and the
// window-blocked CSC (J'*y) on big patterns.
struct LsqSell {
    bool active = false;
    int nblocks = 0, nslices = 0;
    int spw = 0;                   // slices per block (every block is padded to the same count: block b starts at slice b * spw)
    long long nstore = 0;          // stored entries incl. padding
    int wrows = 0;                 // J*x: output rows per block
    int ncw = 1, cwidth = 0;       // J*x, n > LSQ_LDS_X_MAX: column windows per row block (block b = row block * ncw + window),
                                   // columns per window; the gather index of an entry is its column minus the window's first
    int ncb = 0, ccols = 0;        // J'*y: column blocks per gather window, columns per block
    int ngw = 0, grows = 0;        // J'*y: gather windows, rows per gather window
    int2 *d_smeta = nullptr;       // nslices x {entry offset, padded count}
    unsigned *d_info = nullptr;    // nslices*64: output position in the block | entry count << 13
    unsigned short *d_idx16 = nullptr;  // gather index per stored entry
    unsigned short *d_col16 = nullptr;  // J'*y: column of every stored entry (device-side g! scaling)
    double *d_val = nullptr;
    int *d_map = nullptr;          // stored entry -> CSC position (-1: padding)
    double *d_part = nullptr;      // J'*y: [gather window][2n] partials (dots | squares)
    double *d_sx = nullptr;        // J*x with column windows: s .* x of a column-scaled handle (n doubles)
};

inline unsigned long long lsq_next_mat_uid() {
    static std::atomic<unsigned long long> next{1};
    return next.fetch_add(1, std::memory_order_relaxed);
}
struct lsq_mat {
    lsq_ctx *ctx;
    int kind;
    int m, n;
    long long nnz;
    // dense
    double *d_dense = nullptr;
    // sparse
    LsqSegs csc;           // columns; d_val is the user-visible nzval (CSC order)
    LsqSegs csr;           // rows; d_val refreshed from csc.d_val through d_map
    int *d_map = nullptr;  // csr position -> csc position
    bool csr_fresh = false;
    bool csc_fresh = true;  // false: a device g! wrote the mirrors only (see lsq_ensure_csc)
    bool upload_pending = false;   // lsq_mat_set_values_async: the host buffer is still being read
    // COLUMN-SCALED Jacobian J = V diag(s) (lsq_mat_set_colscale): V = the stored values, s = n factors in a device buffer
    // of the caller.  On the sliced layouts nothing is ever multiplied out: J*x gathers s .* x, J'y scales the combined dots,
    // colsumabs2(J) = s.^2 .* colsumabs2(V) with colsumabs2(V) cached.  Elsewhere (small / dense / segment-kernel matrices)
    // the handle keeps V aside (d_cs_base) and multiplies the values out whenever s changes.
    const double *d_colscale = nullptr;        // fused mode only: what the product kernels read
    const double *d_cs_user = nullptr;         // the caller's s (both modes)
    double *d_cs_base = nullptr;               // materialising mode: V in CSC order / dense column-major
    double *d_colsum_base = nullptr;           // fused mode: colsumabs2(V)
    unsigned long long base_version = 0;       // bumps whenever V changes
    unsigned long long colsum_base_version = ~0ull;
    // Row-window-blocked CSC for J'*y when the gathered m-vector outgrows an XCD's L2 (4 MiB):
    // rows are cut into `nwin` windows; segment (w, j) holds column j's entries with rows in
    // window w, so all gathers of a window hit a <= 1 MiB slice of y that stays L2-resident on
    // the XCD the window is scheduled on.  Per-window column sums land in d_bpart (nwin x n)
    // and are combined in index order by k_combine.
    LsqSegs bcsc;
    int nwin = 0;
    int *d_bmap = nullptr;  // bcsc position -> csc position
    double *d_bpart = nullptr;
    double *d_dpart = nullptr;   // dense matrices with few columns: per-window partials of J'y (nwin x n)
    size_t dpart_cap = 0;
    LsqSell srows, scols;   // when active they carry the values instead of csr / bcsc (whose d_val is freed)
    // matrix-free operator (LSQ_MAT_OP): host callbacks on device pointers + one scratch vector of max(m, n)
    int (*op_mul)(int, const double *, double *, void *) = nullptr;
    int (*op_colsum)(double *, void *) = nullptr;
    void *op_user = nullptr;
    double *d_optmp = nullptr;
    unsigned long long version = 0;  // bumps whenever values change
    // process-unique id of this handle: caches keyed on (handle, version) must not confuse a handle with one that was
    // destroyed and re-created at the same address
    unsigned long long uid = lsq_next_mat_uid();
    // cached colsumabs2 (utils.jl:139-151 is called twice per LM iteration by the reference)
    double *d_colsum = nullptr;
    unsigned long long colsum_version = ~0ull;
};

// hipMemset runs on the NULL stream and is asynchronous to the host, while the library's stream is
// non-blocking (not ordered against the NULL stream): a creation-time memset could therefore land
// AFTER the first upload on the library stream and wipe it.  Creation paths are not hot: wait.
#define LSQ_ZERO(ptr, val, bytes)                  \
    do {                                           \
        LSQ_HIP(hipMemset((ptr), (val), (bytes))); \
        LSQ_HIP(hipStreamSynchronize(nullptr));    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Wave reductions on the DPP network (row rotations inside the 16-lane rows, then the four row results through
// v_readlane) instead of six rounds through the LDS crossbar (ds_bpermute): every lane ends up with the result, in
// a fixed association -- ((r0 + r1) + (r2 + r3)) over rows, rotation pairs inside a row -- so all lanes, all waves
// and all kernels agree bit for bit on the same inputs.
template <int CTRL>
__device__ __forceinline__ double lsq_dpp_mov_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lsq_readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += lsq_dpp_mov_f64<0x128>(v);   // row_ror:8
    v += lsq_dpp_mov_f64<0x124>(v);   // row_ror:4
    v += lsq_dpp_mov_f64<0x122>(v);   // row_ror:2
    v += lsq_dpp_mov_f64<0x121>(v);   // row_ror:1
    return (lsq_readlane_f64(v, 0) + lsq_readlane_f64(v, 16)) + (lsq_readlane_f64(v, 32) + lsq_readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, lsq_dpp_mov_f64<0x128>(v));
    v = fmax(v, lsq_dpp_mov_f64<0x124>(v));
    v = fmax(v, lsq_dpp_mov_f64<0x122>(v));
    v = fmax(v, lsq_dpp_mov_f64<0x121>(v));
    return fmax(fmax(lsq_readlane_f64(v, 0), lsq_readlane_f64(v, 16)), fmax(lsq_readlane_f64(v, 32), lsq_readlane_f64(v, 48)));
}

// XCD-aware placement.  Workgroup b of a launch runs on XCD b mod 8 and every XCD has its own L2, so workgroups that read the
// same operand should have indices 8 apart, not adjacent.  lsq_xcd_block turns the hardware index into a LOGICAL index such that
// each XCD owns one contiguous range of logical indices (XCD x: its slot-th workgroup, slot = b / 8, takes logical index
// first(x) + slot): kernels whose neighbouring logical blocks share data (the column blocks of a gather window of J'u, the column
// tiles of one split-K slice of V'B) then find it in their own L2 instead of fetching it once per XCD.
__device__ __forceinline__ int lsq_xcd_block(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

// Sum over the block in a fixed order (deterministic); result valid in thread 0.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *sh /* NT/64 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += sh[i];
    }
    __syncthreads();
    return r;
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double *sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = sh[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) r = fmax(r, sh[i]);
    }
    __syncthreads();
    return r;
}

// Two-stage grid reduction without a second launch: every block publishes its partial with an
// agent-scope (write-through) store, drains it, then takes a ticket; the block that draws the
// last ticket re-reads all partials with agent-scope loads IN INDEX ORDER (run-to-run
// deterministic) and calls fin(total) from thread 0.  Tickets are hierarchical -- one counter per
// XCD-sized group of blocks (own cache line each) and a top counter taken by each group's last
// arriver -- because a single device-scope counter serialises at ~12 ns per arrival (a 2048-block
// grid would spend 25 us in the fan-in alone).  All counters are left at zero.
// Returns true in every thread of the last block (after fin has run).
constexpr int LSQ_RGROUPS = 8;          // ticket groups (blocks b % 8: one per XCD)
constexpr int LSQ_RFLAT = 64;           // up to this many workgroups: one counter instead of the two levels
constexpr int LSQ_CTR_STRIDE = 32;      // unsigneds between counters (128-byte lines)
constexpr int LSQ_CTR_SLOT = (LSQ_RGROUPS + 1) * LSQ_CTR_STRIDE;  // unsigneds per reduction slot
template <int NT, bool IS_MAX = false, class Fin>
__device__ __forceinline__ bool grid_reduce(double block_val, double *partials, unsigned *counter,
                                            int nblocks, double *sh, Fin fin) {
    __shared__ int s_last;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partials[b], block_val, RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int g = b % LSQ_RGROUPS;
        const int ngroups = nblocks < LSQ_RGROUPS ? nblocks : LSQ_RGROUPS;
        const unsigned gsize = (unsigned)((nblocks - g + LSQ_RGROUPS - 1) / LSQ_RGROUPS);
        int last = 0;
        if (nblocks <= LSQ_RFLAT) {
            // few workgroups (the n-length kernels of C4: 40): ONE counter -- two dependent atomics in a row cost a second
            // memory round trip on a latency-bound kernel, and 40 arrivals at ~12 ns each serialise for less than that
            if (__hip_atomic_fetch_add(counter, 1u, RLX_AGENT) == (unsigned)(nblocks - 1)) {
                __hip_atomic_store(counter, 0u, RLX_AGENT);
                last = 1;
            }
        } else if (__hip_atomic_fetch_add(counter + (1 + g) * LSQ_CTR_STRIDE, 1u, RLX_AGENT) == gsize - 1) {
            __hip_atomic_store(counter + (1 + g) * LSQ_CTR_STRIDE, 0u, RLX_AGENT);
            unsigned t2 = __hip_atomic_fetch_add(counter, 1u, RLX_AGENT);
            if (t2 == (unsigned)(ngroups - 1)) {
                __hip_atomic_store(counter, 0u, RLX_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return false;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += NT) {
        double p = __hip_atomic_load(&partials[i], RLX_AGENT);
        acc = IS_MAX ? fmax(acc, p) : acc + p;
    }
    double tot = IS_MAX ? block_max<NT>(acc, sh) : block_sum<NT>(acc, sh);
    if (threadIdx.x == 0) fin(tot);
    return true;
}

// device scalars -> host through the pinned, mapped slot mirror (lsq_vec.hip): either the
// one-thread k_publish_slots launch (lsq_read_slots) or the finalize step of the kernel that
// produces the last scalar (LsqSlotPublish handed to its epilogue), then lsq_wait_slots.
struct LsqSlotPublish {
    const double *src = nullptr;
    int count = 0;
    double *dst = nullptr;
    unsigned long long *seq_word = nullptr;
    unsigned long long seq = 0;
};
LsqSlotPublish lsq_slots_ticket(lsq_ctx *c, int first, int count);
int lsq_wait_slots(lsq_ctx *c, int first, int count, unsigned long long seq, double *h_out);
// check_isfinite(x) (utils.jl:70-75) into a device slot, no host hand-over: -1.0 or the first non-finite index
int lsq_first_nonfinite_to_slot(lsq_ctx *c, int n, const double *x, double *d_slot);
int lsq_read_ints(lsq_ctx *c, const int *d_a, const int *d_b, const int *d_c, const int *d_d, int h_out[4]);   // null pointers read as 0
LsqSlotPublish lsq_ints_ticket(lsq_ctx *c);
int lsq_wait_ints(lsq_ctx *c, unsigned long long seq, const int *d_a, const int *d_b, const int *d_c, const int *d_d, int h_out[4]);

static inline void lsq_run_idle_hook(lsq_ctx *c) {
    if (!c->idle_hook) return;
    int (*h)(void *) = c->idle_hook;
    c->idle_hook = nullptr;
    c->idle_status = h(c->idle_user);
}

static inline unsigned *lsq_ctr(const lsq_ctx *c, int k) { return c->d_counters + (size_t)k * LSQ_CTR_SLOT; }
static inline int lsq_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// reference-order ("exact") small-problem path, lsq_exact.hip
constexpr int LSQ_EXACT_MAX_DIM = 2048;
constexpr long long LSQ_EXACT_MAX_NNZ = 1 << 18;
bool lsq_small_vec(long long n);
bool lsq_small_mat(const lsq_mat *J);
int lsq_exact_product(lsq_mat *J, int trans, const double *x, double *y);   // y = J x / J'x, alpha=1, beta=0
int lsq_exact_colsumabs2(lsq_mat *J, double *out);
int lsq_exact_mul(lsq_mat *J, int trans, double alpha, const double *x, double beta, double *y);
int lsq_seq_reduce(lsq_ctx *c, int mode, int n, const double *x, const double *y, const double *w, double *d_out);
int lsq_exact_lm_damp(lsq_ctx *c, int n, const double *colsum, double inv_delta, double *dtd);

void lsq_workspace_free(void *workspace);

// internal entry points shared between translation units
int lsq_sparse_mul(lsq_mat *J, int trans, double alpha, const double *d_x, double beta, double *d_y);
int lsq_dense_mul(lsq_mat *J, int trans, double alpha, const double *d_x, double beta, double *d_y);
// gather windows of the dense J'y (0: one block per column, k_dense_t)
static inline int lsq_dense_t_windows(const lsq_ctx *c, int m, int n) {
    if (n <= 0 || n >= c->num_cus || m < 8192) return 0;
    int nwin = (4 * c->num_cus + n - 1) / n;
    nwin = std::min(nwin, std::max(1, m / 2048));
    return nwin > 1 ? nwin : 0;
}
int lsq_dense_colsumabs2(lsq_mat *J, double *d_out);
int lsq_dense_part(lsq_mat *J, int nwin);   // makes sure J->d_dpart holds nwin x n doubles
int lsq_dense_part_elems(lsq_mat *J, size_t need);   // ... or `need` doubles
int lsq_sparse_colsumabs2(lsq_mat *J, double *d_out);
int lsq_ensure_csr(lsq_mat *J);
int lsq_ensure_csc(lsq_mat *J);
// mirrors of an arbitrary CSC-ordered value array (e.g. a model's constant matrix) in the layouts
// the products of J read: rows = CSR mirror or sliced rows, cols = window-blocked CSC or sliced cols
int lsq_mirror_rows(lsq_mat *J, const double *d_csc_vals, double *d_out);
int lsq_mirror_cols(lsq_mat *J, const double *d_csc_vals, double *d_out);
long long lsq_mirror_rows_len(const lsq_mat *J);
long long lsq_mirror_cols_len(const lsq_mat *J);
bool lsq_can_fuse_grad_colsum(const lsq_mat *J);
bool lsq_colscale_fusable(const lsq_mat *J);   // lsq_mat_set_colscale would take the fused (never multiplied out) mode
int lsq_sparse_grad_colsum(lsq_mat *J, const double *f, double *g);  // g = J'f, fills the colsum cache
int lsq_sparse_grad_colsum_spec(lsq_mat *J, const double *f, double *g, const double *s_new, const int *gate);  // see lsq_sparse.hip
void lsq_sparse_grad_colsum_adopt(lsq_mat *J);   // the speculative pass ran for the handle's present factors: its colsumabs2 is current
void lsq_sparse_colsum_forget(lsq_mat *J);       // ... or must not be trusted (it ran for factors that were never installed)
const double *lsq_cached_colsum(lsq_mat *J);  // nullptr on failure (error set)
// reads slot values to host (synchronises the stream)
int lsq_read_slots(lsq_ctx *ctx, int first, int count, double *h_out);
