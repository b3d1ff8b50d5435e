// micro-benchmark: cost of a grid barrier inside a 313-block kernel (normal vs cooperative launch)
// against the two-launch equivalent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ void __launch_bounds__(256) k_phase1(double *part, const double *in) {
    if (threadIdx.x == 0) part[blockIdx.x] = in[blockIdx.x] * 2.0;
}
__global__ void __launch_bounds__(256) k_phase2(const double *part, int nb, double *out) {
    __shared__ double s;
    if (threadIdx.x == 0) {
        double a = 0;
        for (int i = 0; i < nb; ++i) a += part[i];
        s = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_fused(double *part, const double *in, int nb, double *out, unsigned *bar, unsigned *gen,
                                               unsigned seq) {
    __shared__ double s;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[blockIdx.x], in[blockIdx.x] * 2.0, RLX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = __hip_atomic_fetch_add(bar, 1u, RLX);
        if (t == (unsigned)nb - 1) {
            __hip_atomic_store(bar, 0u, RLX);
            __hip_atomic_store(gen, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, RLX) != seq) __builtin_amdgcn_s_sleep(1);
        }
        double a = 0;
        for (int i = 0; i < nb; ++i) a += __hip_atomic_load(&part[i], RLX);
        s = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

template <class F>
float timeit(F f, int reps = 200) {
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
    const int nb = 313;
    double *part, *in, *out; unsigned *bar;
    CK(hipMalloc(&part, 4096 * 8)); CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&bar, 256));
    CK(hipMemset(in, 0, 4096 * 8)); CK(hipMemset(bar, 0, 256));
    unsigned seq = 0;
    float us = timeit([&] {
        hipLaunchKernelGGL(k_phase1, dim3(nb), dim3(256), 0, 0, part, in);
        hipLaunchKernelGGL(k_phase2, dim3(nb), dim3(256), 0, 0, part, nb, out);
    });
    printf("two launches:            %6.2f us per pair\n", us);
    us = timeit([&] { ++seq; hipLaunchKernelGGL(k_fused, dim3(nb), dim3(256), 0, 0, part, in, nb, out, bar, bar + 32, seq); });
    printf("fused, normal launch:    %6.2f us\n", us);
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("cooperative launch supported: %d\n", coop);
    if (coop) {
        int nbv = nb;
        unsigned *gen = bar + 32;
        us = timeit([&] {
            ++seq;
            void *args[] = {&part, &in, &nbv, &out, &bar, &gen, &seq};
            (void)hipLaunchCooperativeKernel((const void *)k_fused, dim3(nb), dim3(256), args, 0, 0);
        });
        printf("fused, cooperative:      %6.2f us\n", us);
    }
    us = timeit([&] { hipLaunchKernelGGL(k_phase1, dim3(nb), dim3(256), 0, 0, part, in); });
    printf("single trivial launch:   %6.2f us\n", us);
    return 0;
}
