// Round 6: why does the 4x4x4 form of the trailing update run at ~40 clocks per MFMA in the kernel when the bare instruction
// sustains one per 17?  The update's inner loop without memory: per k step, rotate the A operand (3 x 2 v_mov_dpp) and issue
// 8 MFMAs (4 rotations x 2 row tiles) into 8 accumulators; variants: rotations hoisted out (NOROT), MFMAs only on ONE operand
// pair (SAME), rotations by ds_swizzle-free register copies (COPY: plain v_mov instead of dpp).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int MODE>   // 0: as in the kernel; 1: rotations hoisted out of the loop; 2: one operand pair for all MFMAs; 3: the 16x16x4 form
__global__ void __launch_bounds__(256) k(const double *in, double *out, int iters) {
    const int l = threadIdx.x;
    double wf[16], v0[16], v1[16];
    for (int s = 0; s < 16; ++s) { wf[s] = in[s * 256 + l]; v0[s] = in[4096 + s * 256 + l]; v1[s] = in[8192 + s * 256 + l]; }
    double q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0};
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d c0 = {0, 0, 0, 0}, c1 = c0;
    double r1[16], r2[16], r3[16];
    if (MODE == 1) for (int s = 0; s < 16; ++s) { r1[s] = dpp<0x124>(wf[s]); r2[s] = dpp<0x128>(wf[s]); r3[s] = dpp<0x12C>(wf[s]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (MODE == 3) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s], v0[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[s], v1[s], c1, 0, 0, 0);
                continue;
            }
            double a0 = wf[s], a1, a2, a3;
            if (MODE == 0) { a1 = dpp<0x124>(a0); a2 = dpp<0x128>(a0); a3 = dpp<0x12C>(a0); }
            else if (MODE == 1) { a1 = r1[s]; a2 = r2[s]; a3 = r3[s]; }
            else { a1 = a0; a2 = a0; a3 = a0; }
            const double b0 = MODE == 2 ? v0[0] : v0[s], b1 = MODE == 2 ? v0[0] : v1[s];
            q0[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, q0[0], 0, 0, 0);
            q0[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b0, q0[1], 0, 0, 0);
            q0[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b0, q0[2], 0, 0, 0);
            q0[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b0, q0[3], 0, 0, 0);
            q1[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b1, q1[0], 0, 0, 0);
            q1[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, q1[1], 0, 0, 0);
            q1[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b1, q1[2], 0, 0, 0);
            q1[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b1, q1[3], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + l] = q0[0] + q0[1] + q0[2] + q0[3] + q1[0] + q1[1] + q1[2] + q1[3] + c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}
template <int MODE>
static void run(const char *name, double *in, double *out, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(in, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles = (double)iters * 2;                       // 16x16x64-equivalents per wave
    const double flops = (double)blocks * 4 * tiles * 16 * 2048.0;
    printf("%-34s blocks %4d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  %.1f clocks per 16x16x4-equivalent per SIMD\n", name, blocks, blocks / 256, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (tiles * 16 * (blocks / 256.0)));
}
int main() {
    double *in, *out; hipMalloc(&in, 12288 * 8); hipMalloc(&out, 2048 * 256 * 8);
    const int iters = 2048;
    for (int pass = 0; pass < 2; ++pass) {
    // pass 0: all operands ZERO; pass 1: random operands in (-1, 1) -- the same instruction stream, other bits on the wires
    static double h[12288];
    for (int i = 0; i < 12288; ++i) h[i] = pass ? (rand() / (double)RAND_MAX * 2.0 - 1.0) * 1e-2 : 0.0;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    printf("---- operands: %s\n", pass ? "random" : "zero");
    for (int blocks : {256, 512}) {
        run<3>("16x16x4", in, out, blocks, iters);
        run<0>("4x4x4 + dpp rotations in the loop", in, out, blocks, iters);
        run<1>("4x4x4, rotations hoisted", in, out, blocks, iters);
        run<2>("4x4x4, one operand pair", in, out, blocks, iters);
    }
    }
    return 0;
}
