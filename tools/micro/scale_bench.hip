// micro-benchmark: out[k] = A[k] * sf[col16[k]] streaming variants (g! of the tanh model)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ void __launch_bounds__(1024) k_scale(long long nnz4, const unsigned short *__restrict__ col16,
                                                const double *__restrict__ A, const double *__restrict__ x, int n,
                                                double *__restrict__ out) {
    extern __shared__ double sf[];
    for (int i = threadIdx.x; i < n; i += 1024) {
        double t = tanh(x[i]);
        sf[i] = 1.0 - t * t;
    }
    __syncthreads();
    const long long stride = (long long)gridDim.x * 1024;
    auto ld = [&](long long q, d2 &a0, d2 &a1, u2 &c) {
        const long long k = 4 * q;
        if (VAR & 4) {
            a0 = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(A + k));
            a1 = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(A + k + 2));
            c = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(col16 + k));
        } else {
            a0 = *reinterpret_cast<const d2 *>(A + k);
            a1 = *reinterpret_cast<const d2 *>(A + k + 2);
            c = *reinterpret_cast<const u2 *>(col16 + k);
        }
    };
    auto st = [&](long long q, d2 a0, d2 a1, u2 c) {
        const long long k = 4 * q;
        d2 o0, o1;
        o0.x = a0.x * sf[c.x & 0xffffu];
        o0.y = a0.y * sf[c.x >> 16];
        o1.x = a1.x * sf[c.y & 0xffffu];
        o1.y = a1.y * sf[c.y >> 16];
        if (VAR & 2) {
            __builtin_nontemporal_store(o0, reinterpret_cast<d2 *>(out + k));
            __builtin_nontemporal_store(o1, reinterpret_cast<d2 *>(out + k + 2));
        } else {
            *reinterpret_cast<d2 *>(out + k) = o0;
            *reinterpret_cast<d2 *>(out + k + 2) = o1;
        }
    };
    long long q = blockIdx.x * 1024LL + threadIdx.x;
    if (VAR & 1) {
        for (; q + stride < nnz4; q += 2 * stride) {
            d2 a0, a1, b0, b1;
            u2 ca, cb;
            ld(q, a0, a1, ca);
            ld(q + stride, b0, b1, cb);
            st(q, a0, a1, ca);
            st(q + stride, b0, b1, cb);
        }
    }
    for (; q < nnz4; q += stride) {
        d2 a0, a1;
        u2 ca;
        ld(q, a0, a1, ca);
        st(q, a0, a1, ca);
    }
}

// contiguous chunk per block instead of grid-stride (DRAM page locality per CU)
template <int VAR>
__global__ void __launch_bounds__(1024) k_scale_chunk(long long nnz4, const unsigned short *__restrict__ col16,
                                                      const double *__restrict__ A, const double *__restrict__ x, int n,
                                                      double *__restrict__ out) {
    extern __shared__ double sf[];
    for (int i = threadIdx.x; i < n; i += 1024) {
        double t = tanh(x[i]);
        sf[i] = 1.0 - t * t;
    }
    __syncthreads();
    const long long per = (nnz4 + gridDim.x - 1) / gridDim.x;
    const long long q0 = blockIdx.x * per, q1 = q0 + per < nnz4 ? q0 + per : nnz4;
    for (long long q = q0 + threadIdx.x; q < q1; q += 1024) {
        const long long k = 4 * q;
        d2 a0 = *reinterpret_cast<const d2 *>(A + k);
        d2 a1 = *reinterpret_cast<const d2 *>(A + k + 2);
        u2 c = *reinterpret_cast<const u2 *>(col16 + k);
        d2 o0, o1;
        o0.x = a0.x * sf[c.x & 0xffffu];
        o0.y = a0.y * sf[c.x >> 16];
        o1.x = a1.x * sf[c.y & 0xffffu];
        o1.y = a1.y * sf[c.y >> 16];
        if (VAR & 2) {
            __builtin_nontemporal_store(o0, reinterpret_cast<d2 *>(out + k));
            __builtin_nontemporal_store(o1, reinterpret_cast<d2 *>(out + k + 2));
        } else {
            *reinterpret_cast<d2 *>(out + k) = o0;
            *reinterpret_cast<d2 *>(out + k + 2) = o1;
        }
    }
}

__global__ void k_copy(long long n2, const d2 *__restrict__ a, d2 *__restrict__ o) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) o[i] = a[i];
}
__global__ void k_read(long long n2, const d2 *__restrict__ a, double *o) {
    d2 acc = {0, 0};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) acc += a[i];
    if (acc.x == 123.456) *o = acc.y;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F>
float timeit(F f, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
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
    const long long nnz = 10000000, nnz4 = nnz / 4;
    const int n = 10000;
    // two A/out pairs so that back-to-back launches alternate like the real g! (CSR then BCSC mirror)
    double *A[2], *O[2], *x;
    unsigned short *c16[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&A[i], (nnz + 8) * 8)); CK(hipMalloc(&O[i], (nnz + 8) * 8)); CK(hipMalloc(&c16[i], (nnz + 8) * 2));
        std::vector<unsigned short> h(nnz + 8);
        for (long long k = 0; k < nnz; ++k) h[k] = (unsigned short)(rand() % n);
        CK(hipMemcpy(c16[i], h.data(), (nnz + 8) * 2, hipMemcpyHostToDevice));
        CK(hipMemset(A[i], 0, (nnz + 8) * 8));
    }
    CK(hipMalloc(&x, n * 8)); CK(hipMemset(x, 0, n * 8));
    const size_t lds = n * 8;
    int flip = 0;
#define RUN(name, kern, grid)                                                                            \
    {                                                                                                    \
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 12000 * 8)); \
        float us = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, nnz4, c16[flip], A[flip], x, n, O[flip]); flip ^= 1; }); \
        printf("%-28s grid %4d: %7.2f us  %.2f TB/s\n", name, grid, us, 180e6 / us * 1e-6);                \
    }
    RUN("base", k_scale<0>, 256);
    RUN("unroll2", k_scale<1>, 256);
    RUN("nt-store", k_scale<2>, 256);
    RUN("unroll2+nt-store", k_scale<3>, 256);
    RUN("nt-load", k_scale<4>, 256);
    RUN("nt-load+nt-store", k_scale<6>, 256);
    RUN("all", k_scale<7>, 256);
    RUN("base 512", k_scale<0>, 512);
    RUN("unroll2 512", k_scale<1>, 512);
    RUN("chunk", k_scale_chunk<0>, 256);
    RUN("chunk nt", k_scale_chunk<2>, 256);
    RUN("chunk 512", k_scale_chunk<0>, 512);
    {
        float us = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, nnz / 2, (const d2 *)A[flip], (d2 *)O[flip]); flip ^= 1; });
        printf("plain copy 80MB->80MB: %7.2f us  %.2f TB/s\n", us, 160e6 / us * 1e-6);
        us = timeit([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, nnz / 2, (const d2 *)A[flip], O[0]); flip ^= 1; });
        printf("plain read 80MB: %7.2f us  %.2f TB/s\n", us, 80e6 / us * 1e-6);
        us = timeit([&] { hipMemcpyAsync(O[flip], A[flip], nnz * 8, hipMemcpyDeviceToDevice, 0); flip ^= 1; });
        printf("hipMemcpy D2D 80MB: %7.2f us  %.2f TB/s\n", us, 160e6 / us * 1e-6);
    }
    return 0;
}
