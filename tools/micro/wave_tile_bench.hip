// micro-benchmark: CSR row sums with WAVE-private tiles (no block barriers in the tile loop)
// vs. the spmv-like ceiling.  Synthetic CSR: 1e6 rows x 1e4 cols, 1e7 entries, Poisson-ish rows.
#include <hip/hip_runtime.h>
#ifndef DEPTH
#define DEPTH 2
#endif
#ifndef CH
#define CH 2
#endif
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#include <random>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int WT = 512;          // entries of LDS product space per wave
constexpr int WCAP = WT - 3;     // tile capacity (start is aligned down to a multiple of 4)

struct Regs {
    d2 v0[CH], v1[CH];
    u2 ci[CH];
    int4 meta;   // row0, nrows, k0, k1
    int pa, pe;
    double pre;
};

template <int MODE>
__global__ void __launch_bounds__(1024) k_wave_tiles(const int4 *__restrict__ meta, int ntiles, const int *__restrict__ ptr,
                                                     const unsigned short *__restrict__ idx, const double *__restrict__ val,
                                                     int nnz, int nrows_total, const double *__restrict__ x, int n,
                                                     const double *__restrict__ uold, double *__restrict__ unew, double cu,
                                                     double *partials) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[16];
    double *xl = smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nxpad = (n + 1) & ~1;
    double *prod = smem + nxpad + wv * WT;
    const int W = gridDim.x * 16;                 // waves in the grid
    const int w0 = blockIdx.x * 16 + wv;
    const int kmax = (nnz + 3) & ~3;
    // tile extents are wave-uniform: scalar loads (SGPRs, lgkmcnt) keep them out of the vmcnt queue
    auto load_meta = [&](int t) {
        const int ts = __builtin_amdgcn_readfirstlane(t < ntiles ? (t < 0 ? 0 : t) : ntiles - 1);
        return meta[ts];
    };
    auto load_data = [&](Regs &r, int4 m, int t) {
        if (t >= ntiles || t < 0) m.y = 0;   // (applied here, one step after the load: no drain of newer loads)
        // a real copy made NOW (the extent was fetched two steps ago): otherwise the compiler keeps the
        // extent in the load's own registers and rotates the pair at the loop bottom, right after the
        // newest load was issued (= s_waitcnt vmcnt(0))
        r.meta = m;
        const int ka = m.z & ~3;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = min(ka + c * 256 + 4 * lane, kmax);
            r.v0[c] = *reinterpret_cast<const d2 *>(val + k);
            r.v1[c] = *reinterpret_cast<const d2 *>(val + k + 2);
            r.ci[c] = *reinterpret_cast<const u2 *>(idx + k);
        }
        if (MODE & 4) {
            const int s = min(m.x + lane, nrows_total - 1);
            r.pa = ptr[s];
            r.pe = ptr[s + 1];
            r.pre = uold[s];
        }
    };
    double racc = 0.0;
    auto step = [&](Regs &r, int4 &nm, int t_after) {
        const int row0 = r.meta.x, nr = r.meta.y, ka = r.meta.z & ~3;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            d2 p0, p1;
            p0.x = r.v0[c].x * xl[r.ci[c].x & 0xffffu];
            p0.y = r.v0[c].y * xl[r.ci[c].x >> 16];
            p1.x = r.v1[c].x * xl[r.ci[c].y & 0xffffu];
            p1.y = r.v1[c].y * xl[r.ci[c].y >> 16];
            if (MODE & 1) {
                d2 *dst = reinterpret_cast<d2 *>(prod + c * 256 + 4 * lane);
                dst[0] = p0;
                dst[1] = p1;
            } else {
                racc += (p0.x + p0.y) + (p1.x + p1.y);
            }
        }
        int a = r.pa - ka, e = r.pe - ka;
        // everything still needed from this tile's registers is turned into derived values BEFORE
        // the registers are reloaded (a copy of a load destination would cost a vmcnt(0) rotation)
        double cpre = cu * r.pre;
        int orow = row0 + lane;
        int act = lane < nr;
        // pin the derived values HERE (an opaque asm cannot be sunk into the conditional block below,
        // and the memory clobber keeps the reloads after it): nothing of the old tile is read from r
        // below this line, so every reload can land in the register it replaces
        asm volatile("" : "+v"(a), "+v"(e), "+v"(cpre), "+v"(orow), "+v"(act)::"memory");
        load_data(r, nm, t_after - DEPTH * W);
        nm = load_meta(t_after);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if ((MODE & 2) && act) {
            double sum = 0.0;
            int j = a;
            for (; j + 3 < e; j += 4) {
                const double t0 = prod[j], t1 = prod[j + 1], t2 = prod[j + 2], t3 = prod[j + 3];
                sum += t0; sum += t1; sum += t2; sum += t3;
            }
            for (; j < e; ++j) sum += prod[j];
            const double un = sum - cpre;
            unew[orow] = un;
            racc += un * un;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // Uniform software pipeline: the registers start EMPTY (zero tiles) and every load is issued by
    // the loop body, so the loop-carried registers are exactly the load destinations (a peeled
    // prologue makes the compiler rotate them with v_mov at the loop bottom = vmcnt(0) every pass).
    Regs ra, rb, rc;
    auto clear = [&](Regs &r) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            r.v0[c] = d2{0.0, 0.0};
            r.v1[c] = d2{0.0, 0.0};
            r.ci[c] = u2{0u, 0u};
        }
        r.meta = make_int4(0, 0, 0, 0);
        r.pa = r.pe = 0;
        r.pre = 0.0;
    };
    clear(ra);
    clear(rb);
    clear(rc);
    for (int i = tid; i < n; i += 1024) xl[i] = x[i];
    __syncthreads();
#if DEPTH == 3
    int4 ma = load_meta(w0), mb = load_meta(w0 + W), mc = load_meta(w0 + 2 * W);
    for (int t = w0 - 3 * W; t < ntiles; t += 3 * W) {
        step(ra, ma, t + 6 * W);
        step(rb, mb, t + 7 * W);
        step(rc, mc, t + 8 * W);
    }
#else
    int4 ma = load_meta(w0), mb = load_meta(w0 + W);
    for (int t = w0 - 2 * W; t < ntiles; t += 2 * W) {
        step(ra, ma, t + 4 * W);
        step(rb, mb, t + 5 * W);
    }
#endif
    // block partial
    for (int o = 32; o > 0; o >>= 1) racc += __shfl_down(racc, o, 64);
    if (lane == 0) sh[wv] = racc;
    __syncthreads();
    if (tid == 0) {
        double s = 0;
        for (int i = 0; i < 16; ++i) s += sh[i];
        partials[blockIdx.x] = s;
    }
}

template <class F>
float timeit(F f, int reps = 24) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 8; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int m = 1000000, n = 10000;
    const long long nnz = 10000000;
    std::mt19937_64 rng(1);
    std::vector<int> rowcnt(m, 0);
    std::vector<int> rows(nnz);
    for (long long k = 0; k < nnz; ++k) { rows[k] = rng() % m; rowcnt[rows[k]]++; }
    std::vector<int> ptr(m + 1, 0);
    for (int i = 0; i < m; ++i) ptr[i + 1] = ptr[i] + rowcnt[i];
    std::vector<unsigned short> idx(nnz + 16);
    std::vector<double> val(nnz + 16, 0.0);
    for (long long k = 0; k < nnz; ++k) { idx[k] = rng() % n; val[k] = (double)(rng() % 1000) / 1000.0 - 0.5; }
    // wave tiles: whole rows, <= 64 rows, <= WCAP entries counted from the aligned-down start
    std::vector<int4> meta;
    int r = 0;
    while (r < m) {
        int r0 = r, k0 = ptr[r0], ka = k0 & ~3;
        while (r < m && r - r0 < 64 && ptr[r + 1] - ka <= WT) ++r;
        if (r == r0) { printf("row too long\n"); return 1; }
        meta.push_back(make_int4(r0, r - r0, k0, ptr[r]));
    }
    const int ntiles = (int)meta.size();
    printf("tiles %d (avg %.1f nnz, %.1f rows)\n", ntiles, (double)nnz / ntiles, (double)m / ntiles);
    const int NB = 3;
    int4 *dmeta; int *dptr; unsigned short *didx[NB]; double *dval[NB], *dx, *du[2], *dpart;
    CK(hipMalloc(&dmeta, ntiles * 16)); CK(hipMemcpy(dmeta, meta.data(), ntiles * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&dptr, (m + 1) * 4)); CK(hipMemcpy(dptr, ptr.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < NB; ++i) {
        CK(hipMalloc(&didx[i], (nnz + 16) * 2)); CK(hipMemcpy(didx[i], idx.data(), (nnz + 16) * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dval[i], (nnz + 16) * 8)); CK(hipMemcpy(dval[i], val.data(), (nnz + 16) * 8, hipMemcpyHostToDevice));
    }
    std::vector<double> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = (double)(rng() % 1000) / 500.0 - 1.0;
    CK(hipMalloc(&dx, n * 8)); CK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&du[i], m * 8)); CK(hipMemset(du[i], 0, m * 8)); }
    CK(hipMalloc(&dpart, 4096 * 8));
    const size_t lds = (size_t)(((n + 1) & ~1) + 16 * WT) * 8;
#define RUNMODE(MODE, label)                                                                                     \
    {                                                                                                            \
        CK(hipFuncSetAttribute((const void *)k_wave_tiles<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        for (int nb : {2}) {                                                                                     \
            int flip = 0;                                                                                        \
            float us = timeit([&] {                                                                              \
                hipLaunchKernelGGL(k_wave_tiles<MODE>, dim3(256), dim3(1024), lds, 0, dmeta, ntiles, dptr, didx[flip], dval[flip], \
                                   (int)nnz, m, dx, n, du[0], du[1], 0.5, dpart);                                \
                flip = (flip + 1) % nb;                                                                          \
            });                                                                                                  \
            printf("%-44s cycle %d: %7.2f us\n", label, nb, us);                                                \
        }                                                                                                        \
    }
    {   // regular tiles: fixed 512 entries, aligned, rows ignored (only meaningful for MODE 0)
        std::vector<int4> reg;
        for (long long k = 0; k < nnz; k += 256 * CH) reg.push_back(make_int4(0, 0, (int)k, (int)std::min<long long>(k + 256 * CH, nnz)));
        int4 *dreg;
        CK(hipMalloc(&dreg, reg.size() * 16)); CK(hipMemcpy(dreg, reg.data(), reg.size() * 16, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void *)k_wave_tiles<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int flip = 0;
        float us = timeit([&] {
            hipLaunchKernelGGL(k_wave_tiles<0>, dim3(256), dim3(1024), lds, 0, dreg, (int)reg.size(), dptr, didx[flip], dval[flip],
                               (int)nnz, m, dx, n, du[0], du[1], 0.5, dpart);
            flip ^= 1;
        });
        printf("regular aligned 512-tiles, stream only: %7.2f us (%d tiles)\n", us, (int)reg.size());
    }
    RUNMODE(0, "stream + products only");
    RUNMODE(4, "+ ptr/pre loads");
    RUNMODE(1, "+ LDS product writes");
    RUNMODE(5, "+ LDS writes + ptr/pre loads");
    RUNMODE(7, "full (row sums, store)");
    // check against host
    std::vector<double> hu(m);
    CK(hipMemcpy(hu.data(), du[1], m * 8, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int i = 0; i < m; i += 997) {
        double s = 0;
        for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += val[k] * hx[idx[k]];
        maxerr = std::max(maxerr, std::abs(s - hu[i]));
    }
    printf("max err (sampled rows) %.3e\n", maxerr);
    return 0;
}
