// Validation + timing of the 64 x 64 in-LDS routines of csrc/lsq_small64.h (one workgroup, 256 threads).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I leastsquaresoptim.jl_amd/csrc tools/micro/small64_bench.hip -o /tmp/small64_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lsq_small64.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_chol(const double *G, double *Uo, double *Xo, int reps, int *failo) {
    extern __shared__ double sm[];
    double *M = sm, *W = sm + S64_MAT, *X = sm + 2 * S64_MAT, *T = sm + 3 * S64_MAT;
    __shared__ int fail;
    const int tid = threadIdx.x;
    int f = 0;
    for (int it = 0; it < reps; ++it) {
        for (int e = tid; e < 4096; e += 256) { M[(e >> 6) * S64_LS + (e & 63)] = G[e]; W[(e >> 6) * S64_LS + (e & 63)] = 0.0; }
        __syncthreads();
        f |= s64_chol(M, W, &fail, tid);
        s64_chol_inverse(M, W, T, tid);
    }
    for (int e = tid; e < 4096; e += 256) { Uo[e] = M[(e >> 6) * S64_LS + (e & 63)]; Xo[e] = W[(e >> 6) * S64_LS + (e & 63)]; }
    if (tid == 0) *failo = f;
}
__global__ void __launch_bounds__(256) k_load_only(const double *G, double *Uo, int reps) {
    extern __shared__ double sm[];
    double *M = sm, *W = sm + S64_MAT;
    const int tid = threadIdx.x;
    for (int it = 0; it < reps; ++it) {
        for (int e = tid; e < 4096; e += 256) { M[(e >> 6) * S64_LS + (e & 63)] = G[e]; W[(e >> 6) * S64_LS + (e & 63)] = 0.0; }
        __syncthreads();
    }
    for (int e = tid; e < 4096; e += 256) Uo[e] = M[(e >> 6) * S64_LS + (e & 63)];
}
__global__ void __launch_bounds__(256) k_lu(const double *Q, double *LUo, double *So, double *Xo, double *To, int reps) {
    extern __shared__ double sm[];
    double *M = sm, *Li = sm + S64_MAT, *X = sm + 2 * S64_MAT, *Z = sm + 3 * S64_MAT, *T = sm + 4 * S64_MAT;
    __shared__ double sS[64], sR[64];
    const int tid = threadIdx.x;
    for (int it = 0; it < reps; ++it) {
        for (int e = tid; e < 4096; e += 256) M[(e >> 6) * S64_LS + (e & 63)] = Q[e];
        __syncthreads();
        s64_lu_modified(M, Li, sS, sR, tid);
        // X = inv(U): U = upper part of M.  Build a clean upper copy in Z first (M's strictly lower part is L)
        for (int e = tid; e < 4096; e += 256) { const int r = e >> 6, c = e & 63; Z[r * S64_LS + c] = r <= c ? M[r * S64_LS + c] : 0.0; }
        __syncthreads();
        s64_diaginv_upper(Z, X, tid);
        s64_triinv_levels(Z, X, T, tid);
    }
    for (int e = tid; e < 4096; e += 256) { LUo[e] = M[(e >> 6) * S64_LS + (e & 63)]; Xo[e] = X[(e >> 6) * S64_LS + (e & 63)]; }
    if (tid < 64) So[tid] = sS[tid];
    (void)To;
}

static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

int main() {
    srand(7);
    const int n = 64, m = 256;
    // random tall matrix, orthonormalised twice (MGS) -> Q; G = B'B for another random B
    std::vector<double> B(m * n), Qm(m * n);
    for (auto &v : B) v = rnd();
    for (auto &v : Qm) v = rnd();
    for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < n; ++j) {
            for (int k = 0; k < j; ++k) {
                double d = 0; for (int i = 0; i < m; ++i) d += Qm[i * n + k] * Qm[i * n + j];
                for (int i = 0; i < m; ++i) Qm[i * n + j] -= d * Qm[i * n + k];
            }
            double nr = 0; for (int i = 0; i < m; ++i) nr += Qm[i * n + j] * Qm[i * n + j];
            nr = sqrt(nr); for (int i = 0; i < m; ++i) Qm[i * n + j] /= nr;
        }
    std::vector<double> G(n * n), Qtop(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0; for (int k = 0; k < m; ++k) s += B[k * n + i] * B[k * n + j];
        G[i * n + j] = s;                      // row-major [r][c]
        Qtop[i * n + j] = Qm[i * n + j];
    }
    double *dG, *dU, *dX, *dQ, *dLU, *dS, *dT; int *dF;
    CK(hipMalloc(&dG, 4096 * 8)); CK(hipMalloc(&dU, 4096 * 8)); CK(hipMalloc(&dX, 4096 * 8)); CK(hipMalloc(&dQ, 4096 * 8));
    CK(hipMalloc(&dLU, 4096 * 8)); CK(hipMalloc(&dS, 64 * 8)); CK(hipMalloc(&dT, 4096 * 8)); CK(hipMalloc(&dF, 4));
    CK(hipMemcpy(dG, G.data(), 4096 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ, Qtop.data(), 4096 * 8, hipMemcpyHostToDevice));
    const size_t lds = (4 * S64_MAT + S64_TMP) * sizeof(double);
    CK(hipFuncSetAttribute((const void *)k_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_lu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_load_only, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, int reps) {
        launch(2); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / reps;
    };
    const int R = 400;
    double t_load = timeit([&](int r) { hipLaunchKernelGGL(k_load_only, dim3(1), dim3(256), lds, 0, dG, dU, r); }, R);
    double t_chol = timeit([&](int r) { hipLaunchKernelGGL(k_chol, dim3(1), dim3(256), lds, 0, dG, dU, dX, r, dF); }, R);
    double t_lu = timeit([&](int r) { hipLaunchKernelGGL(k_lu, dim3(1), dim3(256), lds, 0, dQ, dLU, dS, dX + 0, dT, r); }, R);
    // --- validate chol + inverse
    hipLaunchKernelGGL(k_chol, dim3(1), dim3(256), lds, 0, dG, dU, dX, 1, dF);
    std::vector<double> U(4096), X(4096); int fl;
    CK(hipMemcpy(U.data(), dU, 4096 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(X.data(), dX, 4096 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&fl, dF, 4, hipMemcpyDeviceToHost));
    double e_fact = 0, e_inv = 0, e_low = 0, gmax = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k < n; ++k) { s += U[k * n + i] * U[k * n + j]; t += X[i * n + k] * U[k * n + j]; }
        e_fact = fmax(e_fact, fabs(s - G[i * n + j])); gmax = fmax(gmax, fabs(G[i * n + j]));
        e_inv = fmax(e_inv, fabs(t - (i == j ? 1.0 : 0.0)));
        if (i > j) e_low = fmax(e_low, fmax(fabs(U[i * n + j]), fabs(X[i * n + j])));
    }
    printf("chol+inverse: %.2f us per call (load loop alone %.2f us); fail=%d  |U'U-G|/|G| %.2e  |XU-I| %.2e  below-diag %.1e\n",
           t_chol, t_load, fl, e_fact / gmax, e_inv, e_low);
    // --- validate modified LU + inverse of U
    hipLaunchKernelGGL(k_lu, dim3(1), dim3(256), lds, 0, dQ, dLU, dS, dX, dT, 1);
    std::vector<double> LU(4096), S(64);
    CK(hipMemcpy(LU.data(), dLU, 4096 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(S.data(), dS, 64 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(X.data(), dX, 4096 * 8, hipMemcpyDeviceToHost));
    double e_lu = 0, e_ui = 0, minpiv = 1e300;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k <= (i < j ? i : j); ++k) s += (k == i ? 1.0 : LU[i * n + k]) * LU[k * n + j];
        double want = Qtop[i * n + j] - (i == j ? S[i] : 0.0);
        e_lu = fmax(e_lu, fabs(s - want));
        for (int k = 0; k < n; ++k) t += X[i * n + k] * (k <= j ? LU[k * n + j] : 0.0);
        e_ui = fmax(e_ui, fabs(t - (i == j ? 1.0 : 0.0)));
        if (i == j) minpiv = fmin(minpiv, fabs(LU[i * n + i]));
    }
    printf("modified LU + inv(U): %.2f us per call; |LU-(Q-S)| %.2e  |inv(U)U-I| %.2e  min|pivot| %.3f\n", t_lu, e_lu, e_ui, minpiv);
    return 0;
}
