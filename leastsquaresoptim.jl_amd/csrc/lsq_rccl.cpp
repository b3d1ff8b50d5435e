// liblsqrccl.so: the row-sharded all-reduce hook as a direct RCCL call (include/lsqrccl.h).  No link-time dependency on
// RCCL: the functions are bound from the librccl.so the process already uses.
#include "../../include/lsqrccl.h"

#include <dlfcn.h>

#include <cstdio>
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
