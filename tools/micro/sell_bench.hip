// micro-benchmark: J*v with window-sorted sliced-ELL ("SELL-64-window") and an LDS output window.
// One 1024-thread workgroup per row window (<= 4096 rows): x staged in LDS, each wave streams whole
// 64-row slices (lane = row, sequential per-row sums in registers, no LDS products, no barriers),
// results scattered into an LDS window (original row order), then one coalesced epilogue pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#include <numeric>
#include <random>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifndef G
#define G 2          // entries per load group (granularity of the padding): 1 or 2
#endif
constexpr int WROWS_MAX = 4096;

struct SliceMeta { int off; int len; };   // entry offset of the slice (multiple of 64*G), padded length (multiple of G)

// val layout: slice base + ((j/2)*64 + lane)*2 + (j%2); idx likewise (ushort)
constexpr int PG = 6;   // groups (of 2 entries) of a slice fetched one slice ahead; longer rows fetch the rest in place
struct SliceRegs {
    d2 a[PG];
    unsigned c[PG];
    unsigned info;      // local original row | (true length << 16)
    int off, L;
};

__global__ void __launch_bounds__(1024) k_sell(const int *__restrict__ wslice /* nwin+1 */, const int2 *__restrict__ smeta,
                                               const unsigned *__restrict__ rowinfo /* per permuted row: local row | len << 16 */,
                                               const unsigned short *__restrict__ idx, const double *__restrict__ val,
                                               int nwin, int wrows, int m, int nslices, const double *__restrict__ x, int n,
                                               const double *__restrict__ uold, double *__restrict__ unew, double cu,
                                               double *partials) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sh[16];
    double *xl = smem;
    const int nxpad = (n + 1) & ~1;
    double *yw = smem + nxpad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double racc = 0.0;
    bool staged = false;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int base = w * wrows, rows = min(wrows, m - base);
        const int s0 = wslice[w], s1 = wslice[w + 1];
        auto fetch = [&](SliceRegs &r, int s) {   // s may run past s1: clamped, and ignored by the consumer
            const int sc = __builtin_amdgcn_readfirstlane(min(s, nslices - 1));
            const int2 sm = smeta[sc];
            r.off = sm.x;
            r.L = s < s1 ? sm.y : 0;
            const int prow = min((sc - s0) * 64 + lane, wrows - 1);
            r.info = rowinfo[(size_t)base + prow];
            const double *vp = val + (size_t)sm.x + lane * 2;
            const unsigned short *ip = idx + (size_t)sm.x + lane * 2;
            const int gmax = max(sm.y / 2 - 1, 0);
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                const int g = min(u, gmax);
                r.a[u] = *reinterpret_cast<const d2 *>(vp + (size_t)g * 128);
                r.c[u] = *reinterpret_cast<const unsigned *>(ip + (size_t)g * 128);
            }
        };
        auto consume = [&](SliceRegs &r, int s, int snext) {
            const int L = r.L, len = (int)(r.info >> 16), li = (int)(r.info & 0xffffu);
            const int off = r.off;
            double sum = 0.0;
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                const double p0 = r.a[u].x * xl[r.c[u] & 0xffffu], p1 = r.a[u].y * xl[r.c[u] >> 16];
                if (2 * u < len) sum += p0;
                if (2 * u + 1 < len) sum += p1;
            }
            asm volatile("" : "+v"(sum));
            fetch(r, snext);   // the registers of r are free again only after the products
            if (L > 2 * PG) {               // long rows: the rest of the slice, fetched in place
                const double *vp = val + (size_t)off + lane * 2;
                const unsigned short *ip = idx + (size_t)off + lane * 2;
                for (int j = 2 * PG; j < L; j += 2) {
                    const d2 a = *reinterpret_cast<const d2 *>(vp + (size_t)(j / 2) * 128);
                    const unsigned c = *reinterpret_cast<const unsigned *>(ip + (size_t)(j / 2) * 128);
                    const double p0 = a.x * xl[c & 0xffffu], p1 = a.y * xl[c >> 16];
                    if (j < len) sum += p0;
                    if (j + 1 < len) sum += p1;
                }
            }
            if (L > 0 && (s - s0) * 64 + lane < rows) yw[li] = sum;
        };
        SliceRegs ra, rb;
        fetch(ra, s0 + wv);
        fetch(rb, s0 + wv + 16);
        // epilogue inputs for this window, fetched before the streaming starts
        double pre[WROWS_MAX / 1024];
#pragma unroll
        for (int q = 0; q < WROWS_MAX / 1024; ++q) pre[q] = uold[base + min(tid + q * 1024, rows - 1)];
        if (!staged) {
            for (int i = tid; i < n; i += 1024) xl[i] = x[i];
            staged = true;
        }
        __syncthreads();   // x staged (first window) / previous window's epilogue done with yw
        for (int s = s0 + wv; s < s1; s += 32) {
            consume(ra, s, s + 32);
            consume(rb, s + 16, s + 48);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < WROWS_MAX / 1024; ++q) {
            const int i = tid + q * 1024;
            if (i < rows) {
                const double un = yw[i] - cu * pre[q];
                unew[base + i] = un;
                racc += un * un;
            }
        }
    }
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
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 8; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int m = 1000000, n = 10000;
    const long long nnz = 10000000;
    std::mt19937_64 rng(1);
    std::vector<std::vector<std::pair<unsigned short, double>>> rows(m);
    for (long long k = 0; k < nnz; ++k) rows[rng() % m].push_back({(unsigned short)(rng() % n), (double)(rng() % 1000) / 1000.0 - 0.5});
    const int ncu = 256;
    int wrows = ((m + ncu - 1) / ncu + 63) / 64 * 64;
    if (wrows > WROWS_MAX) wrows = WROWS_MAX;
    const int nwin = (m + wrows - 1) / wrows;
    std::vector<int> wslice(nwin + 1, 0);
    std::vector<int2> smeta;
    std::vector<unsigned> rowinfo((size_t)nwin * wrows, 0);
    std::vector<double> val;
    std::vector<unsigned short> idx;
    for (int w = 0; w < nwin; ++w) {
        const int base = w * wrows, nr = std::min(wrows, m - base);
        std::vector<int> ord(nr);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return rows[base + a].size() > rows[base + b].size(); });
        wslice[w] = (int)smeta.size();
        for (int s = 0; s * 64 < nr; ++s) {
            int L = 0;
            for (int l = 0; l < 64 && s * 64 + l < nr; ++l) L = std::max<int>(L, rows[base + ord[s * 64 + l]].size());
            L = (L + G - 1) / G * G;
            const size_t off = val.size();
            val.resize(off + (size_t)L * 64, 0.0);
            idx.resize(off + (size_t)L * 64, 0);
            for (int l = 0; l < 64 && s * 64 + l < nr; ++l) {
                const auto &r = rows[base + ord[s * 64 + l]];
                rowinfo[(size_t)base + s * 64 + l] = (unsigned)ord[s * 64 + l] | ((unsigned)r.size() << 16);
                for (size_t j = 0; j < r.size(); ++j) {
                    const size_t p = off + ((j / G) * 64 + l) * G + (j % G);
                    val[p] = r[j].second;
                    idx[p] = r[j].first;
                }
            }
            smeta.push_back(make_int2((int)off, L));
        }
    }
    wslice[nwin] = (int)smeta.size();
    printf("G=%d windows %d x %d rows, slices %zu, stored entries %zu (padding %.2f%%)\n", G, nwin, wrows, smeta.size(), val.size(),
           100.0 * (val.size() - nnz) / nnz);
    val.resize(val.size() + 1024, 0.0);
    idx.resize(idx.size() + 1024, 0);
    const int NB = 2;
    int *dws; int2 *dsm; unsigned *dri; unsigned short *didx[NB]; double *dval[NB], *dx, *du[2], *dpart;
    CK(hipMalloc(&dws, (nwin + 1) * 4)); CK(hipMemcpy(dws, wslice.data(), (nwin + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dsm, smeta.size() * 8)); CK(hipMemcpy(dsm, smeta.data(), smeta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dri, rowinfo.size() * 4 + 4096)); CK(hipMemcpy(dri, rowinfo.data(), rowinfo.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < NB; ++i) {
        CK(hipMalloc(&didx[i], idx.size() * 2)); CK(hipMemcpy(didx[i], idx.data(), idx.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dval[i], val.size() * 8)); CK(hipMemcpy(dval[i], val.data(), val.size() * 8, hipMemcpyHostToDevice));
    }
    std::vector<double> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = (double)(rng() % 1000) / 500.0 - 1.0;
    CK(hipMalloc(&dx, n * 8)); CK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&du[i], (size_t)(m + 8192) * 8)); CK(hipMemset(du[i], 0, (size_t)(m + 8192) * 8)); }
    CK(hipMalloc(&dpart, 4096 * 8));
    const size_t lds = (size_t)(((n + 1) & ~1) + WROWS_MAX) * 8;
    CK(hipFuncSetAttribute((const void *)k_sell, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int flip = 0;
    float us = timeit([&] {
        hipLaunchKernelGGL(k_sell, dim3(std::min(nwin, 256)), dim3(1024), lds, 0, dws, dsm, dri, didx[flip], dval[flip], nwin, wrows, m, (int)smeta.size(), dx, n,
                           du[0], du[1], 0.5, dpart);
        flip ^= 1;
    });
    printf("SELL J*v: %7.2f us  (%.2f TB/s on the reference's 12 B/nnz + 16 B/row = 136 MB; actual %.1f MB)\n", us, 136e6 / us * 1e-6,
           (val.size() * 10.0 + m * 20.0) / 1e6);
    std::vector<double> hu(m);
    CK(hipMemcpy(hu.data(), du[1], (size_t)m * 8, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int i = 0; i < m; i += 997) {
        double s = 0;
        for (auto &e : rows[i]) s += e.second * hx[e.first];
        maxerr = std::max(maxerr, std::abs(s - hu[i]));
    }
    printf("max err (sampled rows) %.3e\n", maxerr);
    return 0;
}
