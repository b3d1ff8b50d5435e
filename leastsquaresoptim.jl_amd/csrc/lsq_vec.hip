// Context, device memory and BLAS-1 kernels on device vectors (gfx950).
// These are what lsmr.jl:30-44 and the optimizer loops require from a vector type
// (norm, rmul!, axpy!, copyto!, fill!, sum(abs2,.), maximum(abs,.), clamp!, map!) plus
// wdot/wnorm (utils.jl:165-176) and maxabs_projected_gradient (utils.jl:39-55).
#include <cmath>
#include <cstring>

#include <algorithm>

#include <dlfcn.h>

#include "lsq_common.h"

static thread_local char g_err[512] = "";

// ---- roctx ranges (lsq_common.h: LSQ_RANGE) ----------------------------------------------------
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static bool roctx_bind() {
    // contexts may be driven from several host threads: bound exactly once (C++11 static initialisation is thread-safe)
    static const bool bound = [] {
        const char *e = getenv("LSQ_ROCTX");
        if (!e || atoi(e) <= 0) return false;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) return true;
        }
        return false;
    }();
    return bound;
}
LsqRange::LsqRange(const char *name) : on(roctx_bind()) { if (on) g_roctx_push(name); }
LsqRange::~LsqRange() { if (on) g_roctx_pop(); }

void lsq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lsq_last_error(void) { return g_err; }

// ---- debug modes (lsq_common.h: LSQ_LAUNCH) --------------------------------------------------
int lsq_dbg_jitter_us = 0;
int lsq_dbg_serial = 0;
static thread_local unsigned long long g_dbg_rng = 0x9e3779b97f4a7c15ull;   // (per host thread: contexts may be driven from several)
static std::atomic<unsigned long long> g_dbg_stalls{0};
void lsq_dbg_init() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char *e = getenv("LSQ_DEBUG_LAUNCH_JITTER")) lsq_dbg_jitter_us = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("LSQ_DEBUG_SERIAL")) lsq_dbg_serial = atoi(e) > 0 ? atoi(e) : 0;
}
void lsq_dbg_stall() {
    // xorshift64*: a fixed sequence per process (a failing interleaving can be replayed); one launch in four is held back
    g_dbg_rng ^= g_dbg_rng >> 12; g_dbg_rng ^= g_dbg_rng << 25; g_dbg_rng ^= g_dbg_rng >> 27;
    const unsigned long long r = g_dbg_rng * 0x2545f4914f6cdd1dull;
    if ((r & 3) != 0) return;
    unsigned long long us = (r >> 8) % (unsigned long long)(lsq_dbg_jitter_us + 1);
    if (((r >> 2) & 63) == 0) us *= 20;            // the occasional long stall (30-70 ms observed in the wild)
    g_dbg_stalls.fetch_add(1, std::memory_order_relaxed);
    timespec t0, t1;                               // busy wait: usleep's granularity (~60 us) would hide the short stalls
    clock_gettime(CLOCK_MONOTONIC, &t0);
    long long elapsed_ns;                          // signed: tv_nsec wraps when the wait crosses a second boundary
    do {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        elapsed_ns = (long long)(t1.tv_sec - t0.tv_sec) * 1000000000ll + ((long long)t1.tv_nsec - (long long)t0.tv_nsec);
    } while (elapsed_ns < (long long)us * 1000ll);
}
extern "C" int lsq_debug_set(int launch_jitter_us, int serial) {
    lsq_dbg_init();                                // (so that a later first context does not overwrite this from the environment)
    if (launch_jitter_us >= 0) lsq_dbg_jitter_us = launch_jitter_us;
    if (serial >= 0) lsq_dbg_serial = serial;
    return LSQ_OK;
}
extern "C" int lsq_debug_get(int *launch_jitter_us, int *serial, long long *stalls) {
    lsq_dbg_init();
    if (launch_jitter_us) *launch_jitter_us = lsq_dbg_jitter_us;
    if (serial) *serial = lsq_dbg_serial;
    if (stalls) *stalls = (long long)g_dbg_stalls.load(std::memory_order_relaxed);
    return LSQ_OK;
}
extern "C" int lsq_version(void) { return 100; }

extern "C" int lsq_ctx_create(int device, void *stream, lsq_ctx **out) {
    if (!out) return LSQ_EARG;
    int ndev = 0;
    LSQ_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device < 0 || device >= ndev) {
        lsq_set_error("lsq_ctx_create: device %d not available (%d HIP devices); this library "
                      "has no CPU fallback", device, ndev);
        return LSQ_EHIP;
    }
    LSQ_HIP(hipSetDevice(device));
    lsq_dbg_init();
    lsq_ctx *c = new lsq_ctx();
    c->device = device;
    c->mail_epoch = 0;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        LSQ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    LSQ_HIP(hipMalloc(&c->d_slots, LSQ_NSLOTS * sizeof(double)));
    LSQ_ZERO(c->d_slots, 0, LSQ_NSLOTS * sizeof(double));
    LSQ_HIP(hipHostMalloc((void **)&c->h_slots, (LSQ_NSLOTS + 1) * sizeof(double),
                          hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_slots, 0, (LSQ_NSLOTS + 1) * sizeof(double));
    LSQ_HIP(hipHostGetDevicePointer((void **)&c->d_hslots, c->h_slots, 0));
    LSQ_HIP(hipMalloc(&c->d_partials, LSQ_MAX_PARTIALS * sizeof(double)));
    LSQ_HIP(hipMalloc(&c->d_counters, (size_t)LSQ_NSLOTS * LSQ_CTR_SLOT * sizeof(unsigned)));
    LSQ_ZERO(c->d_counters, 0, (size_t)LSQ_NSLOTS * LSQ_CTR_SLOT * sizeof(unsigned));
    LSQ_HIP(hipHostMalloc((void **)&c->h_mail, sizeof(LsqMailbox),
                          hipHostMallocMapped | hipHostMallocCoherent));
    memset((void *)c->h_mail, 0, sizeof(LsqMailbox));
    LSQ_HIP(hipHostGetDevicePointer((void **)&c->d_mail, (void *)c->h_mail, 0));
    hipDeviceProp_t prop;
    LSQ_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    // LSQ_DEBUG_NUM_CUS=<k>: let the launch heuristics see a device of k compute units (e.g. 32 = one CPX partition of an
    // MI355X): the branches a partitioned device takes -- no slab exchange in the QR panel, launch-per-panel Cholesky when the
    // tiles do not fit, fewer workgroups everywhere -- run on an unpartitioned box (tests/test_b_gpu_kernels.py).  Only ever
    // lowers the count: every co-residency assumption made for k CUs holds on the real device.
    if (const char *e = getenv("LSQ_DEBUG_NUM_CUS")) {
        const int k = atoi(e);
        if (k >= 1 && k < c->num_cus) c->num_cus = k;
    }
    LSQ_HIP(hipDeviceSynchronize());
    *out = c;
    return LSQ_OK;
}

extern "C" int lsq_ctx_destroy(lsq_ctx *c) {
    if (!c) return LSQ_OK;
    hipStreamSynchronize(c->stream);
    lsq_workspace_free(c->workspace);
    c->workspace = nullptr;
    hipFree(c->d_slots);
    hipHostFree(c->h_slots);
    hipFree(c->d_partials);
    hipFree(c->d_counters);
    hipHostFree((void *)c->h_mail);
    for (hipEvent_t e : c->prof_pool) hipEventDestroy(e);
    if (c->copy_done) hipEventDestroy(c->copy_done);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    for (hipStream_t hs : c->helper_stream)
        if (hs) hipStreamDestroy(hs);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return LSQ_OK;
}

extern "C" int lsq_prof_begin(lsq_ctx *c, int max_samples) {
    for (int k = 0; k < 2; ++k) {
        for (hipEvent_t e : c->prof_ev[k]) hipEventDestroy(e);
        c->prof_ev[k].clear();
    }
    c->prof_max = max_samples;
    // the events of the samples to come (bounded: a pool beyond a few thousand pairs is created on demand instead)
    const size_t want = 2 * (size_t)std::min(max_samples, 4096);
    while (c->prof_pool.size() < want) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        c->prof_pool.push_back(e);
    }
    return LSQ_OK;
}

extern "C" int lsq_prof_select(lsq_ctx *c, int kernel_mask) {
    c->prof_kernels = kernel_mask & 0xff;
    c->prof_stride = (kernel_mask >> 8) > 0 ? (kernel_mask >> 8) : 1;   // bits 8..: time every k-th launch
    c->prof_tick = 0;
    return LSQ_OK;
}

extern "C" int lsq_prof_end(lsq_ctx *c, double avg_ms[2], int count[2]) {
    c->prof_max = 0;
    LSQ_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        auto &v = c->prof_ev[k];
        double tot = 0.0;
        int n = 0;
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, v[i], v[i + 1]) == hipSuccess) {
                tot += ms;
                ++n;
            }
        }
        avg_ms[k] = n ? tot / n : 0.0;
        count[k] = n;
        for (hipEvent_t e : v) hipEventDestroy(e);
        v.clear();
    }
    return LSQ_OK;
}

__global__ void k_fill(int n, double a, double *__restrict__ x);

extern "C" int lsq_prof_overhead(lsq_ctx *c, int pairs, double *h_ms) {
    if (pairs < 1) pairs = 1;
    std::vector<hipEvent_t> ev(2 * pairs);
    for (auto &e : ev) LSQ_HIP(hipEventCreate(&e));
    // a small kernel before each pair so the first marker waits for real work, like in the solve
    for (int i = 0; i < pairs; ++i) {
        LSQ_LAUNCH(k_fill, dim3(1), dim3(LSQ_NT), 0, c->stream, 1, 0.0, c->d_slots + LSQ_NSLOTS - 1);
        LSQ_HIP(hipEventRecord(ev[2 * i], c->stream));
        LSQ_HIP(hipEventRecord(ev[2 * i + 1], c->stream));
    }
    LSQ_HIP(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (int i = 0; i < pairs; ++i) {
        float ms = 0.f;
        LSQ_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    for (auto &e : ev) hipEventDestroy(e);
    *h_ms = tot / pairs;
    return LSQ_OK;
}

extern "C" int lsq_ctx_sync(lsq_ctx *c) {
    LSQ_HIP(hipStreamSynchronize(c->stream));
    return LSQ_OK;
}
extern "C" void *lsq_ctx_stream(lsq_ctx *c) { return (void *)c->stream; }

extern "C" int lsq_malloc(lsq_ctx *c, size_t bytes, void **out) {
    LSQ_HIP(hipSetDevice(c->device));
    LSQ_HIP(hipMalloc(out, bytes ? bytes : 8));
    return LSQ_OK;
}
extern "C" int lsq_free(lsq_ctx *c, void *p) {
    if (!p) return LSQ_OK;
    LSQ_HIP(hipStreamSynchronize(c->stream));
    LSQ_HIP(hipFree(p));
    return LSQ_OK;
}
extern "C" int lsq_host_alloc(lsq_ctx *c, size_t bytes, void **out) {
    LSQ_HIP(hipSetDevice(c->device));
    LSQ_HIP(hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
    return LSQ_OK;
}
extern "C" int lsq_host_free(lsq_ctx *c, void *p) {
    if (!p) return LSQ_OK;
    LSQ_HIP(hipStreamSynchronize(c->stream));
    if (c->copy_stream) LSQ_HIP(hipStreamSynchronize(c->copy_stream));
    LSQ_HIP(hipHostFree(p));
    return LSQ_OK;
}
extern "C" int lsq_h2d(lsq_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!bytes) return LSQ_OK;
    LSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    return LSQ_OK;
}
extern "C" int lsq_d2h(lsq_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!bytes) return LSQ_OK;
    LSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    return LSQ_OK;
}
// device-to-device copies run as an ordinary kernel: two back-to-back hipMemcpyAsync D2D calls were
// measured to leave a ~60 us hole in the stream between them (runtime-side staging), which cost the
// LM loop ~75 us per accepted step.
__global__ void __launch_bounds__(LSQ_NT) k_copy16(size_t n16, const double2 *__restrict__ src, double2 *__restrict__ dst) {
    for (size_t i = blockIdx.x * (size_t)LSQ_NT + threadIdx.x; i < n16; i += (size_t)gridDim.x * LSQ_NT) dst[i] = src[i];
}
__global__ void __launch_bounds__(LSQ_NT) k_copy8(size_t n8, const double *__restrict__ src, double *__restrict__ dst) {
    for (size_t i = blockIdx.x * (size_t)LSQ_NT + threadIdx.x; i < n8; i += (size_t)gridDim.x * LSQ_NT) dst[i] = src[i];
}

extern "C" int lsq_d2d(lsq_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!bytes || dst == src) return LSQ_OK;
    const bool a16 = (((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0;
    const bool a8 = (((uintptr_t)dst | (uintptr_t)src | bytes) & 7) == 0;
    if (a16) {
        size_t n = bytes / 16;
        size_t g = (n + LSQ_NT - 1) / LSQ_NT, cap = (size_t)c->num_cus * 8;
        LSQ_LAUNCH(k_copy16, dim3((unsigned)(g > cap ? cap : g)), dim3(LSQ_NT), 0, c->stream, n,
                           (const double2 *)src, (double2 *)dst);
    } else if (a8) {
        size_t n = bytes / 8;
        size_t g = (n + LSQ_NT - 1) / LSQ_NT, cap = (size_t)c->num_cus * 8;
        LSQ_LAUNCH(k_copy8, dim3((unsigned)(g > cap ? cap : g)), dim3(LSQ_NT), 0, c->stream, n,
                           (const double *)src, (double *)dst);
    } else {
        LSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// Device scalars -> host without hipStreamSynchronize (whose interrupt-driven wake-up costs ~50 us
// per outer iteration): a one-thread kernel stores the values into pinned, mapped host memory and
// then a sequence word (system-scope release); the host spins on the sequence word.
__global__ void k_publish_slots(const double *__restrict__ src, int count, double *dst, unsigned long long *seq_word,
                                unsigned long long seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < count; ++i)
            __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

LsqSlotPublish lsq_slots_ticket(lsq_ctx *c, int first, int count) {
    LsqSlotPublish p;
    p.src = c->d_slots + first;
    p.count = count;
    p.dst = c->d_hslots + first;
    p.seq_word = (unsigned long long *)(c->d_hslots + LSQ_NSLOTS);
    p.seq = ++c->slot_seq;
    return p;
}

int lsq_wait_slots(lsq_ctx *c, int first, int count, unsigned long long seq, double *h_out) {
    volatile unsigned long long *hw = (volatile unsigned long long *)(c->h_slots + LSQ_NSLOTS);
    unsigned long long spins = 0;
    while (*hw != seq) {
        if ((++spins & 0xfffffu) != 0) continue;
        if (getenv("LSQ_DEBUG_WAITS")) fprintf(stderr, "lsq_wait_slots: 2^20 spins waiting for seq %llu, word %llu (first %d count %d)\n", seq, (unsigned long long)*hw, first, count);
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) {   // launch failure / device fault: the word will never arrive
            lsq_set_error("HIP error while waiting for the iteration's scalars: %s", hipGetErrorString(q));
            return LSQ_EHIP;
        }
        if (*hw != seq) {
            // stream drained but the word is not visible (should not happen with coherent host memory)
            LSQ_HIP(hipMemcpy(c->h_slots + first, c->d_slots + first, count * sizeof(double), hipMemcpyDeviceToHost));
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < count; ++i) h_out[i] = ((volatile double *)c->h_slots)[first + i];
    return LSQ_OK;
}

// up to four device ints (status words of the dense solvers) -> host, through the same pinned mirror and the same spin: a
// hipMemcpyAsync + hipStreamSynchronize wake-up costs ~50 us of host latency, this one launch + poll a few
__global__ void k_publish_ints(const int *a, const int *b, const int *c2, const int *d, double *dst, unsigned long long *seq_word,
                               unsigned long long seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int *src[4] = {a, b, c2, d};
        for (int i = 0; i < 4; ++i)
            __hip_atomic_store(dst + i, src[i] ? (double)*src[i] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int lsq_read_ints(lsq_ctx *c, const int *d_a, const int *d_b, const int *d_c, const int *d_d, int h_out[4]) {
    LsqSlotPublish p = lsq_ints_ticket(c);
    LSQ_LAUNCH(k_publish_ints, dim3(1), dim3(64), 0, c->stream, d_a, d_b, d_c, d_d, p.dst, p.seq_word, p.seq);
    LSQ_HIP(hipGetLastError());
    return lsq_wait_ints(c, p.seq, d_a, d_b, d_c, d_d, h_out);
}
// the ticket of a status hand-over a kernel of the caller performs itself (k_tri_bsolve's last workgroup: dst[0..3] as
// doubles, then the sequence word with release semantics at system scope), and the host side of it
LsqSlotPublish lsq_ints_ticket(lsq_ctx *c) { return lsq_slots_ticket(c, LSQ_NSLOTS - 8, 4); }   // (slots 56..59: no reduction uses them)
int lsq_wait_ints(lsq_ctx *c, unsigned long long seq, const int *d_a, const int *d_b, const int *d_c, const int *d_d, int h_out[4]) {
    constexpr int FIRST = LSQ_NSLOTS - 8;
    LsqSlotPublish p;
    p.seq = seq;
    // (the fallback copy of lsq_wait_slots reads d_slots, which this kernel does not fill: only reached if the pinned word
    //  never becomes visible although the stream drained)
    double v[4];
    volatile unsigned long long *hw = (volatile unsigned long long *)(c->h_slots + LSQ_NSLOTS);
    unsigned long long spins = 0;
    while (*hw != p.seq) {
        if ((++spins & 0xfffffu) != 0) continue;
        if (getenv("LSQ_DEBUG_WAITS")) fprintf(stderr, "lsq_wait_ints: 2^20 spins waiting for seq %llu, word %llu\n", p.seq, (unsigned long long)*hw);
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) {
            lsq_set_error("HIP error while waiting for a solver status word: %s", hipGetErrorString(q));
            return LSQ_EHIP;
        }
        if (*hw != p.seq) {   // drained, word not visible: read the ints the slow way
            const int *src[4] = {d_a, d_b, d_c, d_d};
            for (int i = 0; i < 4; ++i) {
                h_out[i] = 0;
                if (src[i]) LSQ_HIP(hipMemcpy(&h_out[i], src[i], sizeof(int), hipMemcpyDeviceToHost));
            }
            return LSQ_OK;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < 4; ++i) { v[i] = ((volatile double *)c->h_slots)[FIRST + i]; h_out[i] = (int)v[i]; }
    return LSQ_OK;
}

int lsq_read_slots(lsq_ctx *c, int first, int count, double *h_out) {
    LsqSlotPublish p = lsq_slots_ticket(c, first, count);
    LSQ_LAUNCH(k_publish_slots, dim3(1), dim3(64), 0, c->stream, p.src, p.count, p.dst, p.seq_word, p.seq);
    LSQ_HIP(hipGetLastError());
    return lsq_wait_slots(c, first, count, p.seq, h_out);
}

// ---------------------------------------------------------------------------------------------
// elementwise kernels: grid-stride, 2 doubles (16 B) per lane per step where alignment allows
// ---------------------------------------------------------------------------------------------
static inline int ew_grid(const lsq_ctx *c, long long n) {
    long long g = (n + LSQ_NT - 1) / LSQ_NT;
    long long cap = (long long)c->num_cus * 8;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

__global__ void __launch_bounds__(LSQ_NT) k_axpy(int n, double a, const double *__restrict__ x,
                                                  double *__restrict__ y) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT)
        y[i] += a * x[i];
}
__global__ void __launch_bounds__(LSQ_NT) k_scal(int n, double a, double *__restrict__ x) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT)
        x[i] *= a;
}
__global__ void __launch_bounds__(LSQ_NT) k_fill(int n, double a, double *__restrict__ x) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT)
        x[i] = a;
}
__global__ void __launch_bounds__(LSQ_NT) k_clamp(int n, double lo, double hi, double *__restrict__ x) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT) {
        double v = x[i];
        x[i] = v > hi ? hi : (v < lo ? lo : v);  // Base.clamp
    }
}
__global__ void __launch_bounds__(LSQ_NT) k_ediv(int n, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ o) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT)
        o[i] = x[i] / y[i];
}
__global__ void __launch_bounds__(LSQ_NT) k_box_clip(int n, double *__restrict__ dx,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ lo,
                                                      const double *__restrict__ hi) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT) {
        // Julia's min / max propagate NaN from EITHER argument (fmin / fmax drop it): a NaN step component must stay
        // NaN so that the trial point is non-finite, as in the reference (levenberg_marquardt.jl:89-98)
        double d = dx[i];
        if (lo) { const double a = x[i] - lo[i]; d = (d != d || a != a) ? d + a : (d < a ? d : a); }
        if (hi) { const double a = x[i] - hi[i]; d = (d != d || a != a) ? d + a : (d > a ? d : a); }
        dx[i] = d;
    }
}

// reductions: MODE 0 sum(x), 1 sum(x^2), 2 sum(w*x*y), 3 max|x|, 4 projected-gradient max
template <int MODE>
__global__ void __launch_bounds__(LSQ_NT)
k_reduce(int n, const double *__restrict__ x, const double *__restrict__ y,
         const double *__restrict__ w, const double *__restrict__ lo, const double *__restrict__ hi,
         double *partials, unsigned *counter, double *out) {
    __shared__ double sh[LSQ_NT / 64];
    double acc = 0.0;
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT) {
        double v = x[i];
        if (MODE == 0) acc += v;
        if (MODE == 1) acc += v * v;
        if (MODE == 2) acc += w ? w[i] * v * y[i] : v * y[i];   // wdot / dot
        if (MODE == 3) {  // maximum(abs, x); a NaN is reported as +inf (both fail every "<= tol")
            double a = fabs(v);
            if (isnan(a)) a = INFINITY;
            acc = fmax(acc, a);
        }
        if (MODE == 4) {  // utils.jl:44-53: v = g[i], y = x
            double gi = v;
            if (lo && y[i] <= lo[i] && gi > 0.0) gi = 0.0;
            else if (hi && y[i] >= hi[i] && gi < 0.0) gi = 0.0;
            double a = fabs(gi);
            if (a > acc) acc = a;
        }
    }
    constexpr bool IS_MAX = (MODE >= 3);
    double bv = IS_MAX ? block_max<LSQ_NT>(acc, sh) : block_sum<LSQ_NT>(acc, sh);
    grid_reduce<LSQ_NT, IS_MAX>(bv, partials, counter, gridDim.x, sh, [=](double t) { *out = t; });
}

// check_isfinite (utils.jl:70-75): smallest non-finite index, or -1.  A hit at index i is coded
// as 1e15-(i+1) > 0 so that a max-reduction returns the smallest index; 0 codes "none".
__global__ void __launch_bounds__(LSQ_NT)
k_first_nonfinite(int n, const double *__restrict__ x, double *partials, unsigned *counter, double *out) {
    __shared__ double sh[LSQ_NT / 64];
    double code = 0.0;
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n;
         i += (long long)gridDim.x * LSQ_NT) {
        if (!isfinite(x[i]) && code == 0.0) code = 1e15 - (double)(i + 1);
    }
    double bv = block_max<LSQ_NT>(code, sh);
    grid_reduce<LSQ_NT, true>(bv, partials, counter, gridDim.x, sh, [=](double t) {
        *out = (t == 0.0) ? -1.0 : (1e15 - t) - 1.0;
    });
}

#define LSQ_LAUNCH_EW(kernel, n, ...)                                                     \
    do {                                                                                  \
        if ((n) > 0)                                                                      \
            LSQ_LAUNCH(kernel, dim3(ew_grid(c, (n))), dim3(LSQ_NT), 0, c->stream, \
                               __VA_ARGS__);                                              \
        LSQ_HIP(hipGetLastError());                                                       \
    } while (0)

extern "C" int lsq_axpy(lsq_ctx *c, int n, double a, const double *x, double *y) {
    LSQ_LAUNCH_EW(k_axpy, n, n, a, x, y);
    return LSQ_OK;
}
extern "C" int lsq_scal(lsq_ctx *c, int n, double a, double *x) {
    LSQ_LAUNCH_EW(k_scal, n, n, a, x);
    return LSQ_OK;
}
extern "C" int lsq_copy(lsq_ctx *c, int n, const double *x, double *y) {
    return lsq_d2d(c, y, x, (size_t)n * sizeof(double));
}
extern "C" int lsq_fill(lsq_ctx *c, int n, double a, double *x) {
    LSQ_LAUNCH_EW(k_fill, n, n, a, x);
    return LSQ_OK;
}
extern "C" int lsq_clamp(lsq_ctx *c, int n, double lo, double hi, double *x) {
    LSQ_LAUNCH_EW(k_clamp, n, n, lo, hi, x);
    return LSQ_OK;
}
__global__ void __launch_bounds__(LSQ_NT) k_emul(int n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ o) {
    for (long long i = blockIdx.x * (long long)LSQ_NT + threadIdx.x; i < n; i += (long long)gridDim.x * LSQ_NT) o[i] = x[i] * y[i];
}
extern "C" int lsq_emul(lsq_ctx *c, int n, const double *x, const double *y, double *o) {
    LSQ_LAUNCH_EW(k_emul, n, n, x, y, o);
    return LSQ_OK;
}
extern "C" int lsq_ediv(lsq_ctx *c, int n, const double *x, const double *y, double *o) {
    LSQ_LAUNCH_EW(k_ediv, n, n, x, y, o);
    return LSQ_OK;
}
extern "C" int lsq_box_clip(lsq_ctx *c, int n, double *dx, const double *x, const double *lo,
                            const double *hi) {
    if (!lo && !hi) return LSQ_OK;
    LSQ_LAUNCH_EW(k_box_clip, n, n, dx, x, lo, hi);
    return LSQ_OK;
}

template <int MODE>
static int reduce_to_host(lsq_ctx *c, int n, const double *x, const double *y, const double *w,
                          const double *lo, const double *hi, double *h_out) {
    if (n <= 0) {
        *h_out = 0.0;
        return LSQ_OK;
    }
    if (MODE <= 2 && lsq_small_vec(n)) {  // reference summation order for small vectors (lsq_exact.hip)
        LSQ_TRY(lsq_seq_reduce(c, MODE, n, x, y, w, c->d_slots + 0));
        return lsq_read_slots(c, 0, 1, h_out);
    }
    int grid = ew_grid(c, n);
    LSQ_LAUNCH(k_reduce<MODE>, dim3(grid), dim3(LSQ_NT), 0, c->stream, n, x, y, w, lo, hi,
                       c->d_partials, lsq_ctr(c, 0), c->d_slots + 0);
    LSQ_HIP(hipGetLastError());
    LSQ_TRY(lsq_read_slots(c, 0, 1, h_out));
    return LSQ_OK;
}

extern "C" int lsq_sum(lsq_ctx *c, int n, const double *x, double *h) {
    return reduce_to_host<0>(c, n, x, nullptr, nullptr, nullptr, nullptr, h);
}
extern "C" int lsq_sumsq(lsq_ctx *c, int n, const double *x, double *h) {
    return reduce_to_host<1>(c, n, x, nullptr, nullptr, nullptr, nullptr, h);
}
extern "C" int lsq_nrm2(lsq_ctx *c, int n, const double *x, double *h) {
    LSQ_TRY(reduce_to_host<1>(c, n, x, nullptr, nullptr, nullptr, nullptr, h));
    *h = sqrt(*h);
    return LSQ_OK;
}
extern "C" int lsq_dot(lsq_ctx *c, int n, const double *x, const double *y, double *h) {
    return reduce_to_host<2>(c, n, x, y, nullptr, nullptr, nullptr, h);
}
extern "C" int lsq_wdot(lsq_ctx *c, int n, const double *x, const double *y, const double *w, double *h) {
    return reduce_to_host<2>(c, n, x, y, w, nullptr, nullptr, h);
}
extern "C" int lsq_amax(lsq_ctx *c, int n, const double *x, double *h) {
    LSQ_TRY(reduce_to_host<3>(c, n, x, nullptr, nullptr, nullptr, nullptr, h));
    return LSQ_OK;
}
extern "C" int lsq_amax_projected(lsq_ctx *c, int n, const double *g, const double *x,
                                  const double *lo, const double *hi, double *h) {
    if (!lo && !hi) return lsq_amax(c, n, g, h);
    return reduce_to_host<4>(c, n, g, x, nullptr, lo, hi, h);
}
int lsq_first_nonfinite_to_slot(lsq_ctx *c, int n, const double *x, double *d_slot) {
    if (n <= 0) return lsq_fill(c, 1, -1.0, d_slot);
    LSQ_LAUNCH(k_first_nonfinite, dim3(ew_grid(c, n)), dim3(LSQ_NT), 0, c->stream, n, x, c->d_partials, lsq_ctr(c, 0), d_slot);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
extern "C" int lsq_first_nonfinite(lsq_ctx *c, int n, const double *x, int *h_index) {
    if (n <= 0) {
        *h_index = -1;
        return LSQ_OK;
    }
    int grid = ew_grid(c, n);
    LSQ_LAUNCH(k_first_nonfinite, dim3(grid), dim3(LSQ_NT), 0, c->stream, n, x,
                       c->d_partials, lsq_ctr(c, 0), c->d_slots + 0);
    LSQ_HIP(hipGetLastError());
    double v;
    LSQ_TRY(lsq_read_slots(c, 0, 1, &v));
    *h_index = (int)v;
    return LSQ_OK;
}


// ---------------------------------------------------------------------------------------------
// measurement / test helpers: a neighbour on the device
// ---------------------------------------------------------------------------------------------
// `workgroups` workgroups of 256 threads that each hold `lds_bytes` of LDS and spin for `milliseconds`: what an RCCL kernel
// or any other tenant of the device does to the fast paths that assume co-resident workgroups (include/lsqhip.h)
__global__ void __launch_bounds__(256) k_occupy(long long ticks, double *sink) {
    extern __shared__ double hog[];
    hog[threadIdx.x] = (double)threadIdx.x;
    const long long t0 = wall_clock64();
    double acc = 0.0;
    while (wall_clock64() - t0 < ticks) {
        acc += hog[(threadIdx.x * 7 + (int)acc) & 255];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == -1.0) sink[0] = acc;
}
extern "C" int lsq_bench_occupy(lsq_ctx *c, int workgroups, int lds_bytes, double milliseconds) {
    if (!c || workgroups <= 0 || lds_bytes < 2048 || lds_bytes > 160 * 1024 || !(milliseconds > 0)) return LSQ_EARG;
    if (!c->occupy_stream) LSQ_HIP(hipStreamCreateWithFlags(&c->occupy_stream, hipStreamNonBlocking));
    LSQ_TRY(lsq_set_lds(c, (const void *)k_occupy, 160 * 1024));
    int rate_khz = 100000;   // wall_clock64 ticks at 100 MHz on gfx9
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c->device);
    const long long ticks = (long long)(milliseconds * (double)rate_khz);
    LSQ_LAUNCH(k_occupy, dim3(workgroups), dim3(256), (size_t)lds_bytes, c->occupy_stream, ticks, c->d_slots + LSQ_NSLOTS - 2);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
extern "C" int lsq_bench_occupy_wait(lsq_ctx *c) {
    if (c && c->occupy_stream) LSQ_HIP(hipStreamSynchronize(c->occupy_stream));
    return LSQ_OK;
}
extern "C" int lsq_ctx_device_info(const lsq_ctx *c, int *num_cus, int *num_xcds, char *arch_name, int name_cap) {
    if (!c) return LSQ_EARG;
    if (num_cus) *num_cus = c->num_cus;
    if (num_xcds) *num_xcds = std::max(1, c->num_cus / 32);
    if (arch_name && name_cap > 0) {
        hipDeviceProp_t prop;
        LSQ_HIP(hipGetDeviceProperties(&prop, c->device));
        snprintf(arch_name, (size_t)name_cap, "%s", prop.gcnArchName);
    }
    return LSQ_OK;
}
extern "C" int lsq_ctx_fallback_stats(const lsq_ctx *c, int h_giveups[4]) {
    if (!c || !h_giveups) return LSQ_EARG;
    for (int i = 0; i < 4; ++i) h_giveups[i] = c->fallback_giveups[i];
    return LSQ_OK;
}
extern "C" int lsq_ctx_tail_stats(const lsq_ctx *c, long long h_out[2]) {
    if (!c || !h_out) return LSQ_EARG;
    h_out[0] = c->tail_spec[0];
    h_out[1] = c->tail_spec[1];
    return LSQ_OK;
}
