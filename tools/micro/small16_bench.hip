// per-call cost of the 16 x 16 register kernels of lsq_small64.h and the accuracy of v_rsq_f64 / v_rcp_f64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lsq_small64.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_acc(const double *x, double *r1, double *r2, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i < n) { r1[i] = __builtin_amdgcn_rsq(x[i]); r2[i] = __builtin_amdgcn_rcp(x[i]); }
}
template <int V>
__global__ void __launch_bounds__(64) k_c16(const double *G, double *out, int reps) {
    __shared__ double M[S64_MAT], W[S64_MAT];
    const int lane = threadIdx.x;
    int bad = 0;
    for (int it = 0; it < reps; ++it) {
        for (int e = lane; e < 256; e += 64) M[(e >> 4) * S64_LS + (e & 15)] = G[(e >> 4) * 64 + (e & 15)];
        __syncthreads();
        if (V == 0) bad |= s64_chol16(M, W, 0, lane);
        else { __shared__ double sS[64], sR[64]; s64_lu16(M, W, sS, sR, 0, lane); }
        __syncthreads();
    }
    for (int e = lane; e < 256; e += 64) out[e] = M[(e >> 4) * S64_LS + (e & 15)] + W[(e >> 4) * S64_LS + (e & 15)] + bad;
}
int main() {
    srand(3);
    const int n = 1 << 16;
    std::vector<double> x(n), a(n), b(n);
    for (int i = 0; i < n; ++i) x[i] = exp(((double)rand() / RAND_MAX - 0.5) * 40.0);
    double *dx, *da, *db;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_acc, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    CK(hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) { e1 = fmax(e1, fabs(a[i] * sqrt(x[i]) - 1.0)); e2 = fmax(e2, fabs(b[i] * x[i] - 1.0)); }
    printf("v_rsq_f64 max rel err %.3e (2^%.1f)   v_rcp_f64 max rel err %.3e (2^%.1f)\n", e1, log2(e1), e2, log2(e2));
    std::vector<double> G(4096);
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) G[i * 64 + j] = (i == j ? 20.0 : 0.0) + 0.3 * sin(i * 7 + j * 3) + 0.3 * sin(j * 7 + i * 3);
    double *dG, *dO; CK(hipMalloc(&dG, 4096 * 8)); CK(hipMalloc(&dO, 4096 * 8));
    CK(hipMemcpy(dG, G.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1v; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1v));
    auto timeit = [&](auto kern) {
        const int R = 2000;
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dG, dO, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dG, dO, R); CK(hipEventRecord(e1v)); CK(hipEventSynchronize(e1v));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1v)); return ms * 1e3 / R;
    };
    printf("chol16: %.3f us per call   lu16: %.3f us per call (one wavefront, incl. a 256-element LDS refill + 2 barriers)\n",
           timeit(k_c16<0>), timeit(k_c16<1>));
    return 0;
}
