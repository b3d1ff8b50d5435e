// fp64 MFMA issue rate, the OTHER shape: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks, 512 flops per instruction) beside
// v_mfma_f64_16x16x4_f64 (2048 flops) -- round 6: does the small shape sustain more than the 46-49 TFLOP/s of the big one?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k16(double *out, int iters) {
    v4d acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (v4d){0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k4(double *out, int iters) {
    double acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class K>
static void run(const char *name, K kern, int nacc, double flops_per, double *d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * nacc * flops_per;
    printf("%-10s NACC %2d blocks %4d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  (%.1f clocks per MFMA per SIMD at 2.4 GHz)\n", name, nacc, blocks,
           blocks / 256, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * nacc * (blocks / 256.0)));
}
int main() {
    double *d; hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 8192;
    for (int blocks : {256, 512, 1024}) run("16x16x4", k16<4>, 4, 2048.0, d, blocks, iters);
    for (int blocks : {256, 512, 1024}) run("16x16x4", k16<8>, 8, 2048.0, d, blocks, iters);
    for (int blocks : {256, 512, 1024, 2048}) run("4x4x4_4b", k4<4>, 4, 512.0, d, blocks, iters);
    for (int blocks : {256, 512, 1024, 2048}) run("4x4x4_4b", k4<8>, 8, 512.0, d, blocks, iters);
    for (int blocks : {256, 1024}) run("4x4x4_4b", k4<16>, 16, 512.0, d, blocks, iters);
    return 0;
}
