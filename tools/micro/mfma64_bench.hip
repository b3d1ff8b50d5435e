// fp64 MFMA issue rate on one CU-set: N dependent-free accumulators per wave, 4 waves per block
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
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
int main() {
    double *d; hipMalloc(&d, 8 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 4096;
        k<4><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k<4><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", blocks, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * iters * 4));
    }
    return 0;
}
