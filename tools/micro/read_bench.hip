// micro-benchmark: streaming-read ceilings on MI355X as a function of footprint and launch shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NT, int UNROLL>
__global__ void __launch_bounds__(NT) k_read(long long n2, const d2 *__restrict__ a, double *o) {
    d2 acc = {0, 0};
    const long long stride = (long long)gridDim.x * NT;
    long long i = blockIdx.x * (long long)NT + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n2; i += stride) acc += a[i];
    if (acc.x == 123.456) *o = acc.y;
}

// val (8 B) + idx16 (2 B) streams, product with an LDS-staged vector, everything summed
template <int NT, int UNROLL>
__global__ void __launch_bounds__(NT) k_spmv_like(long long nnz4, const unsigned short *__restrict__ idx,
                                                  const double *__restrict__ val, const double *__restrict__ x, int n,
                                                  double *o) {
    extern __shared__ double xl[];
    for (int i = threadIdx.x; i < n; i += NT) xl[i] = x[i];
    __syncthreads();
    double acc = 0;
    const long long stride = (long long)gridDim.x * NT;
    long long q = blockIdx.x * (long long)NT + threadIdx.x;
    for (; q + (UNROLL - 1) * stride < nnz4; q += UNROLL * stride) {
        d2 a0[UNROLL], a1[UNROLL];
        u2 c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long k = 4 * (q + u * stride);
            a0[u] = *reinterpret_cast<const d2 *>(val + k);
            a1[u] = *reinterpret_cast<const d2 *>(val + k + 2);
            c[u] = *reinterpret_cast<const u2 *>(idx + k);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc += a0[u].x * xl[c[u].x & 0xffffu] + a0[u].y * xl[c[u].x >> 16] + a1[u].x * xl[c[u].y & 0xffffu] +
                   a1[u].y * xl[c[u].y >> 16];
        }
    }
    if (acc == 123.456) *o = acc;
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
    const long long nnz = 10000000;
    const int n = 10000, NB = 8;
    double *A[NB], *x, *o;
    unsigned short *c16[NB];
    std::vector<unsigned short> h(nnz + 8);
    for (long long k = 0; k < nnz; ++k) h[k] = (unsigned short)(rand() % n);
    for (int i = 0; i < NB; ++i) {
        CK(hipMalloc(&A[i], (nnz + 8) * 8)); CK(hipMalloc(&c16[i], (nnz + 8) * 2));
        CK(hipMemcpy(c16[i], h.data(), (nnz + 8) * 2, hipMemcpyHostToDevice));
        CK(hipMemset(A[i], 0, (nnz + 8) * 8));
    }
    CK(hipMalloc(&x, n * 8)); CK(hipMemset(x, 0, n * 8)); CK(hipMalloc(&o, 64));
    for (int nb : {1, 2, 4, 8}) {
        int flip = 0;
        float us = timeit([&] { hipLaunchKernelGGL((k_read<256, 1>), dim3(2048), dim3(256), 0, 0, nnz / 2, (const d2 *)A[flip], o); flip = (flip + 1) % nb; });
        printf("read 80MB, cycle over %d arrays (%4d MB): 2048x256 u1 %7.2f us %.2f TB/s", nb, nb * 80, us, 80e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_read<256, 4>), dim3(2048), dim3(256), 0, 0, nnz / 2, (const d2 *)A[flip], o); flip = (flip + 1) % nb; });
        printf(" | u4 %7.2f us %.2f TB/s", us, 80e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_read<1024, 4>), dim3(256), dim3(1024), 0, 0, nnz / 2, (const d2 *)A[flip], o); flip = (flip + 1) % nb; });
        printf(" | 256x1024 u4 %7.2f us %.2f TB/s", us, 80e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_read<1024, 8>), dim3(256), dim3(1024), 0, 0, nnz / 2, (const d2 *)A[flip], o); flip = (flip + 1) % nb; });
        printf(" | 256x1024 u8 %7.2f us %.2f TB/s\n", us, 80e6 / us * 1e-6);
    }
    const size_t lds = n * 8;
    CK(hipFuncSetAttribute((const void *)k_spmv_like<1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 12000 * 8));
    CK(hipFuncSetAttribute((const void *)k_spmv_like<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 12000 * 8));
    CK(hipFuncSetAttribute((const void *)k_spmv_like<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 12000 * 8));
    CK(hipFuncSetAttribute((const void *)k_spmv_like<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 12000 * 8));
    for (int nb : {1, 2, 4}) {
        int flip = 0;
        float us = timeit([&] { hipLaunchKernelGGL((k_spmv_like<1024, 2>), dim3(256), dim3(1024), lds, 0, nnz / 4, c16[flip], A[flip], x, n, o); flip = (flip + 1) % nb; });
        printf("spmv-like 100MB, cycle %d (%4d MB): 256x1024 u2 %7.2f us %.2f TB/s", nb, nb * 100, us, 100e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_spmv_like<1024, 4>), dim3(256), dim3(1024), lds, 0, nnz / 4, c16[flip], A[flip], x, n, o); flip = (flip + 1) % nb; });
        printf(" | u4 %7.2f us %.2f TB/s", us, 100e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_spmv_like<512, 4>), dim3(512), dim3(512), lds, 0, nnz / 4, c16[flip], A[flip], x, n, o); flip = (flip + 1) % nb; });
        printf(" | 512x512 u4 %7.2f us %.2f TB/s", us, 100e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL((k_spmv_like<256, 4>), dim3(512), dim3(256), lds, 0, nnz / 4, c16[flip], A[flip], x, n, o); flip = (flip + 1) % nb; });
        printf(" | 512x256 u4 %7.2f us %.2f TB/s\n", us, 100e6 / us * 1e-6);
    }
    return 0;
}
