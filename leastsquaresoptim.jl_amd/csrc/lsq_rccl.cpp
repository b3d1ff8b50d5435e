// liblsqrccl.so: the collectives of sharded runs as direct RCCL calls (include/lsqrccl.h): the row-sharded all-reduce hook
// and the per-outer-iteration scalar exchange of independent problems.  No link-time dependency on RCCL: the functions
// are bound from the librccl.so the process already uses.  Host code only (HIP runtime API for a side stream, events and
// page-locked staging buffers; no kernels).
#include "../../include/lsqrccl.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*CommDestroyFn)(void *);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, void *, void *);
typedef const char *(*GetErrorStringFn)(int);
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;   // rccl.h: ncclFloat64 = 8, ncclSum = 0

void *g_lib = nullptr;
GetUniqueIdFn p_unique = nullptr;
CommInitRankFn p_init = nullptr;
CommDestroyFn p_destroy = nullptr;
AllReduceFn p_allreduce = nullptr;
GetErrorStringFn p_errstr = nullptr;
thread_local std::string g_err;

struct Comm {
    void *nccl = nullptr;
    long long calls = 0, doubles = 0;
};

int fail(const char *what, int rc) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s (%d)", what, p_errstr ? p_errstr(rc) : "RCCL error", rc);
    g_err = buf;
    return 1;
}

int allreduce_cb(double *d_buf, int count, void *hip_stream, void *user) {
    Comm *c = (Comm *)user;
    if (!c || !c->nccl || !p_allreduce) return 1;
    const int rc = p_allreduce(d_buf, d_buf, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->nccl, hip_stream);
    if (rc != 0) return fail("ncclAllReduce", rc);
    c->calls++;
    c->doubles += count;
    return 0;
}
}  // namespace

extern "C" const char *lsq_rccl_last_error(void) { return g_err.c_str(); }

extern "C" int lsq_rccl_load(const char *path) {
    if (g_lib) return 0;
    void *h = dlopen(path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h && !path) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        g_err = std::string("dlopen(librccl): ") + dlerror();
        return 1;
    }
    p_unique = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    p_init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    p_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    p_allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    p_errstr = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!p_unique || !p_init || !p_destroy || !p_allreduce) {
        g_err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        dlclose(h);
        return 1;
    }
    g_lib = h;
    return 0;
}

extern "C" int lsq_rccl_unique_id(unsigned char out[128]) {
    if (!g_lib && lsq_rccl_load(nullptr) != 0) return 1;
    UniqueId id;
    const int rc = p_unique(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(out, id.internal, 128);
    return 0;
}

extern "C" int lsq_rccl_comm_create(const unsigned char idb[128], int rank, int world, void **comm_out) {
    if (!comm_out || !idb) return 1;
    if (!g_lib && lsq_rccl_load(nullptr) != 0) return 1;
    UniqueId id;
    memcpy(id.internal, idb, 128);
    Comm *c = new Comm();
    const int rc = p_init(&c->nccl, world, id, rank);
    if (rc != 0) {
        delete c;
        return fail("ncclCommInitRank", rc);
    }
    *comm_out = c;
    return 0;
}

extern "C" int lsq_rccl_comm_destroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    if (c->nccl && p_destroy) p_destroy(c->nccl);
    delete c;
    return 0;
}

extern "C" void *lsq_rccl_allreduce_callback(void) { return (void *)allreduce_cb; }

extern "C" int lsq_rccl_comm_stats(void *comm, long long *calls, long long *doubles) {
    Comm *c = (Comm *)comm;
    if (!c) return 1;
    if (calls) *calls = c->calls;
    if (doubles) *doubles = c->doubles;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The one exchange of independent problems (include/lsqhip.h: lsq_options.allreduce): per OUTER iteration a SUM
// all-reduce of world + 3 doubles {sum ssr, converged count, leaving count, one gradient-norm slot per rank}, from which
// every rank recovers {sum, max, all-converged}.  The protocol (who waits, who gets the previous result, who stops issuing
// collectives) is ONE state machine; the transport under it is either RCCL on a side stream (production) or a pair of
// callbacks (the CPU tests drive it over gloo and compare it with leastsquaresoptim.jl_amd/sharding.py, its Python twin).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int NSLOT = 3;
struct Xchg {
    int rank = 0, world = 1, count = 0;
    // protocol state
    int slot = 0;            // staging slot of the NEXT exchange
    int inflight = -1;       // slot whose all-reduce an active rank left in flight (-1: none)
    bool have_last = false, aborted = false;
    double last[3] = {0, 0, 0};
    double *h_buf[2] = {nullptr, nullptr};
    // transport
    lsq_xchg_issue_fn issue = nullptr;
    lsq_xchg_finish_fn finish = nullptr;
    void *tuser = nullptr;
    // RCCL transport
    Comm *comm = nullptr;
    int device = -1;
    hipStream_t xs = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    double *d_buf[2] = {nullptr, nullptr};
    double timeout_s = 120.0;
    long long collectives = 0, synchronous = 0;
};

int rccl_issue(double *h_buf, int count, int slot, void *user) {
    Xchg *x = (Xchg *)user;
    // staging copies and the collective on a stream of the exchange's own: never behind the LM loop's queued kernels
    if (hipSetDevice(x->device) != hipSuccess) { g_err = "exchange: hipSetDevice failed"; return 1; }
    if (hipMemcpyAsync(x->d_buf[slot], h_buf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, x->xs) != hipSuccess) {
        g_err = "exchange: staging copy (host -> device) failed";
        return 1;
    }
    const int rc = p_allreduce(x->d_buf[slot], x->d_buf[slot], (size_t)count, NCCL_DOUBLE, NCCL_SUM, x->comm->nccl, x->xs);
    if (rc != 0) return fail("ncclAllReduce (scalar exchange)", rc);
    x->comm->calls++;
    x->comm->doubles += count;
    if (hipMemcpyAsync(h_buf, x->d_buf[slot], (size_t)count * sizeof(double), hipMemcpyDeviceToHost, x->xs) != hipSuccess ||
        hipEventRecord(x->ev[slot], x->xs) != hipSuccess) {
        g_err = "exchange: staging copy (device -> host) failed";
        return 1;
    }
    return 0;
}
int rccl_finish(int slot, void *user) {
    Xchg *x = (Xchg *)user;
    // polled, not hipEventSynchronize: this thread is the one that feeds the LM loop's launches, and a blocking wait's
    // wake-up costs tens of microseconds.  Bounded: a peer that died without its farewell must not hang this rank forever.
    long long spins = 0;
    std::chrono::steady_clock::time_point deadline;
    bool armed = false;
    for (;;) {
        const hipError_t st = hipEventQuery(x->ev[slot]);
        if (st == hipSuccess) return 0;
        if (st != hipErrorNotReady) { (void)hipGetLastError(); g_err = "exchange: hipEventQuery failed"; return 1; }
        if ((++spins & 0xFFFF) == 0) {
            const auto now = std::chrono::steady_clock::now();
            if (!armed) { deadline = now + std::chrono::milliseconds((long long)(x->timeout_s * 1e3)); armed = true; }
            else if (now > deadline) {
                char buf[160];
                snprintf(buf, sizeof buf, "exchange not completed after %.0f s (a peer rank is gone?)", x->timeout_s);
                g_err = buf;
                return 1;
            }
        }
    }
}

// completes the exchange of `slot`, decodes it; sets x->aborted if some rank reported that it is leaving
int finish_decode(Xchg *x, int slot, double out[3]) {
    if (x->finish(slot, x->tuser) != 0) return 1;
    const double *hv = x->h_buf[slot];
    if (hv[2] > 0.5) x->aborted = true;
    double gmax = hv[NSLOT];
    for (int r = 1; r < x->world; ++r) gmax = hv[NSLOT + r] > gmax ? hv[NSLOT + r] : gmax;
    out[0] = hv[0];
    out[1] = gmax;
    out[2] = hv[1] >= x->world - 0.5 ? 1.0 : 0.0;
    return 0;
}

// lsq_allreduce_callback: vals = {ssr, maxabs_gr, converged (1) / active (0) / leaving with an error (-1)}
int xchg_cb(double *vals, int count, void *user) {
    Xchg *x = (Xchg *)user;
    if (!x || count < 3) return 1;
    if (x->aborted) return 2;
    const int k = x->slot;
    // the exchange left in flight an iteration ago is complete by now: look at it BEFORE issuing the next one, so that a
    // rank that learns of an abort issues no further collective
    if (x->inflight >= 0) {
        if (finish_decode(x, x->inflight, x->last) != 0) return 1;
        x->have_last = true;
        x->inflight = -1;
        if (x->aborted) return 2;
    }
    const bool leaving = vals[2] < -0.5, conv = vals[2] > 0.5;
    double *hv = x->h_buf[k];
    for (int i = 0; i < x->count; ++i) hv[i] = 0.0;
    hv[0] = leaving ? 0.0 : vals[0];
    hv[1] = conv ? 1.0 : 0.0;
    hv[2] = leaving ? 1.0 : 0.0;
    hv[NSLOT + x->rank] = leaving ? 0.0 : vals[1];
    if (x->issue(hv, x->count, k, x->tuser) != 0) return 1;
    x->collectives++;
    double res[3];
    if (leaving || conv || !x->have_last) {        // frozen / leaving / first call: this iteration's values, synchronously
        if (finish_decode(x, k, res) != 0) return 1;
        x->synchronous++;
    } else {                                       // active rank: never acts on the result -- the previous exchange's values,
        memcpy(res, x->last, sizeof res);          // this one stays in flight under the iteration's device work
        x->inflight = k;
    }
    memcpy(x->last, res, sizeof res);
    x->have_last = true;
    x->slot = k ^ 1;
    if (x->aborted) return leaving ? 0 : 2;
    vals[0] = res[0]; vals[1] = res[1]; vals[2] = res[2];
    return 0;
}

Xchg *xchg_new(int rank, int world) {
    Xchg *x = new Xchg();
    x->rank = rank;
    x->world = world;
    x->count = world + NSLOT;
    if (const char *e = getenv("LSQ_EXCHANGE_TIMEOUT_S")) { const double t = atof(e); if (t > 0) x->timeout_s = t; }
    return x;
}
}  // namespace

extern "C" int lsq_rccl_xchg_create(void *comm, int rank, int world, void **out) {
    if (!comm || !out || world < 1 || rank < 0 || rank >= world) { g_err = "lsq_rccl_xchg_create: bad arguments"; return 1; }
    Xchg *x = xchg_new(rank, world);
    x->comm = (Comm *)comm;
    x->issue = rccl_issue;
    x->finish = rccl_finish;
    x->tuser = x;
    bool ok = hipGetDevice(&x->device) == hipSuccess && hipStreamCreateWithFlags(&x->xs, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k < 2; ++k) {
        ok = hipMalloc((void **)&x->d_buf[k], (size_t)x->count * sizeof(double)) == hipSuccess &&
             hipHostMalloc((void **)&x->h_buf[k], (size_t)x->count * sizeof(double), hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&x->ev[k], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        g_err = "lsq_rccl_xchg_create: HIP allocation failed";
        lsq_rccl_xchg_destroy(x);
        return 1;
    }
    *out = x;
    return 0;
}

extern "C" int lsq_rccl_xchg_create_custom(int rank, int world, lsq_xchg_issue_fn issue, lsq_xchg_finish_fn finish, void *user,
                                           void **out) {
    if (!issue || !finish || !out || world < 1 || rank < 0 || rank >= world) { g_err = "lsq_rccl_xchg_create_custom: bad arguments"; return 1; }
    Xchg *x = xchg_new(rank, world);
    x->issue = issue;
    x->finish = finish;
    x->tuser = user;
    for (int k = 0; k < 2; ++k) x->h_buf[k] = (double *)calloc((size_t)x->count, sizeof(double));
    *out = x;
    return 0;
}

extern "C" int lsq_rccl_xchg_drain(void *xchg) {
    Xchg *x = (Xchg *)xchg;
    if (!x) return 0;
    if (x->inflight >= 0) {
        const int k = x->inflight;
        x->inflight = -1;
        if (finish_decode(x, k, x->last) != 0) return 1;
        x->have_last = true;
    }
    return 0;
}

// A new run on the same handle: completes what the previous run left in flight, then forgets the previous run's
// last result and a seen abort, so that "every rank's first call waits for its exchange" holds for EVERY run (an active rank's
// first call would otherwise return the previous run's {sum ssr, max |g|, all converged = 1}).  Issues no collective; every
// rank calls it at the same point of its call sequence (between two lsq_optimize runs).
extern "C" int lsq_rccl_xchg_reset(void *xchg) {
    Xchg *x = (Xchg *)xchg;
    if (!x) return 0;
    const int rc = lsq_rccl_xchg_drain(xchg);
    x->have_last = false;
    x->last[0] = x->last[1] = x->last[2] = 0.0;
    x->aborted = false;
    return rc;
}

extern "C" int lsq_rccl_xchg_destroy(void *xchg) {
    Xchg *x = (Xchg *)xchg;
    if (!x) return 0;
    if (x->comm) {                       // RCCL transport: HIP resources (the communicator belongs to the caller)
        if (x->xs) (void)hipStreamSynchronize(x->xs);
        for (int k = 0; k < 2; ++k) {
            if (x->ev[k]) (void)hipEventDestroy(x->ev[k]);
            if (x->d_buf[k]) (void)hipFree(x->d_buf[k]);
            if (x->h_buf[k]) (void)hipHostFree(x->h_buf[k]);
        }
        if (x->xs) (void)hipStreamDestroy(x->xs);
    } else {
        for (int k = 0; k < 2; ++k) free(x->h_buf[k]);
    }
    delete x;
    return 0;
}

extern "C" void *lsq_rccl_xchg_callback(void) { return (void *)xchg_cb; }

extern "C" int lsq_rccl_xchg_stats(void *xchg, long long *collectives, long long *synchronous, int *aborted) {
    Xchg *x = (Xchg *)xchg;
    if (!x) return 1;
    if (collectives) *collectives = x->collectives;
    if (synchronous) *synchronous = x->synchronous;
    if (aborted) *aborted = x->aborted ? 1 : 0;
    return 0;
}
