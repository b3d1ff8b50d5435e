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
template <int NACC>
static void run(double *d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC %d blocks %d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", NACC, blocks, blocks / 256, ms, flops / ms / 1e9);
}
// Run under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE` to get the
// MFMA-busy fraction and the MEASURED shader clock of each launch (tools/c3_pmc.sh): whether a shortfall against 78.6 TFLOP/s
// is the clock (power) or the issue rate.
int main() {
    double *d; hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 8192;
    for (int blocks : {256, 512, 1024, 2048}) run<4>(d, blocks, iters);
    for (int blocks : {256, 512, 1024}) run<8>(d, blocks, iters);
    for (int blocks : {256, 1024}) run<2>(d, blocks, iters);
    for (int blocks : {256, 1024}) run<1>(d, blocks, iters);
    return 0;
}
