// Round 6: WHICH instruction streams run v_mfma_f64_16x16x4_f64 at ~68 clocks (74 TFLOP/s) and which at ~106 (47 TFLOP/s)?
//   NACC accumulators in round robin; operands: SAME = one (a, b) register pair for every MFMA, DIST = a fresh pair per MFMA
//   (16 pairs in registers), SHA = consecutive MFMAs share the a operand in groups of NACC (the kernels' pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, int OPS>   // OPS 0: SAME, 1: DIST, 2: SHA
__global__ void __launch_bounds__(256) k(const double *in, double *out, int iters) {
    const int l = threadIdx.x;
    double a[16], b[16];
    for (int s = 0; s < 16; ++s) { a[s] = in[s * 256 + l]; b[s] = in[4096 + s * 256 + l]; }
    v4d c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (v4d){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const double x = OPS == 0 ? a[0] : OPS == 1 ? a[(s * NACC + i) & 15] : a[s];
                const double y = OPS == 0 ? b[0] : OPS == 1 ? b[(s * NACC + i + 5) & 15] : b[(s + i) & 15];
                c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i], 0, 0, 0);
            }
    }
    double sum = 0;
    for (int i = 0; i < NACC; ++i) sum += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + l] = sum;
}
template <int NACC, int OPS>
static void run(double *in, double *out, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, OPS><<<blocks, 256>>>(in, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC, OPS><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * NACC;
    printf("NACC %d  operands %-5s blocks %4d (%d waves/SIMD): %.1f TFLOP/s  %.1f clocks per MFMA per SIMD\n", NACC,
           OPS == 0 ? "SAME" : OPS == 1 ? "DIST" : "SHA", blocks, blocks / 256, (double)blocks * 4 * n * 2048.0 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (n * (blocks / 256.0)));
}
int main() {
    double *in, *out; hipMalloc(&in, 8192 * 8); hipMalloc(&out, 2048 * 256 * 8);
    static double h[8192];
    for (int i = 0; i < 8192; ++i) h[i] = (rand() / (double)RAND_MAX * 2.0 - 1.0) * 1e-2;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 1024;
    for (int blocks : {256, 512}) {
        run<1, 0>(in, out, blocks, iters); run<1, 1>(in, out, blocks, iters);
        run<2, 0>(in, out, blocks, iters); run<2, 1>(in, out, blocks, iters); run<2, 2>(in, out, blocks, iters);
        run<4, 0>(in, out, blocks, iters); run<4, 1>(in, out, blocks, iters); run<4, 2>(in, out, blocks, iters);
        run<8, 0>(in, out, blocks, iters); run<8, 1>(in, out, blocks, iters); run<8, 2>(in, out, blocks, iters);
    }
    return 0;
}
