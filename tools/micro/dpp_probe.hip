// Probe (gfx950): semantics and issue cost of v_fmac_f64_dpp row_newbcast and v_permlane16_swap_b32, the two instructions the
// register-resident 16 x 16 Cholesky (lsq_small64.h: s64_chol16) uses instead of v_readlane pairs.
//   hipcc -O3 --offload-arch=gfx950 dpp_probe.hip -o dpp_probe && ./dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_sem(const double *in, double *out, unsigned *uo) {
    const int l = threadIdx.x;
    double a = in[l], b = in[64 + l], c = in[128 + l];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b));
    out[l] = c;
    unsigned x = 1000 + l, y = 2000 + l;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    uo[l] = x;
    uo[64 + l] = y;
}
template <int MODE>
__global__ void k_rate(double *p, long long *cyc, int iters) {
    const int l = threadIdx.x;
    double a = p[l], b = p[64 + l];
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_fmac_f64_e32 %0, %8, %9\nv_fmac_f64_e32 %1, %8, %9\nv_fmac_f64_e32 %2, %8, %9\nv_fmac_f64_e32 %3, %8, %9\n"
                         "v_fmac_f64_e32 %4, %8, %9\nv_fmac_f64_e32 %5, %8, %9\nv_fmac_f64_e32 %6, %8, %9\nv_fmac_f64_e32 %7, %8, %9\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
        } else if (MODE == 2) {     // the readlane pair + fma with an SGPR operand
            asm volatile("v_readlane_b32 s20, %8, 1\nv_readlane_b32 s21, %9, 1\nv_readlane_b32 s22, %8, 2\nv_readlane_b32 s23, %9, 2\n"
                         "v_readlane_b32 s24, %8, 3\nv_readlane_b32 s25, %9, 3\nv_readlane_b32 s26, %8, 4\nv_readlane_b32 s27, %9, 4\n"
                         "s_nop 3\n"
                         "v_fma_f64 %0, s[20:21], %10, %0\nv_fma_f64 %1, s[22:23], %10, %1\nv_fma_f64 %2, s[24:25], %10, %2\nv_fma_f64 %3, s[26:27], %10, %3\n"
                         "v_readlane_b32 s20, %8, 5\nv_readlane_b32 s21, %9, 5\nv_readlane_b32 s22, %8, 6\nv_readlane_b32 s23, %9, 6\n"
                         "v_readlane_b32 s24, %8, 7\nv_readlane_b32 s25, %9, 7\nv_readlane_b32 s26, %8, 8\nv_readlane_b32 s27, %9, 8\n"
                         "s_nop 3\n"
                         "v_fma_f64 %4, s[20:21], %10, %4\nv_fma_f64 %5, s[22:23], %10, %5\nv_fma_f64 %6, s[24:25], %10, %6\nv_fma_f64 %7, s[26:27], %10, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                         : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else {                    // a dependent chain of v_fma_f64 (latency)
            asm volatile("v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
                         : "+v"(c0) : "v"(a), "v"(b));
        }
    }
    const long long t1 = clock64();
    if (l == 0) *cyc = t1 - t0;
    p[128 + l] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main() {
    double *d; unsigned *u; long long *c;
    hipMalloc(&d, 4096); hipMalloc(&u, 1024); hipMalloc(&c, 8);
    std::vector<double> h(192);
    for (int i = 0; i < 64; ++i) { h[i] = 100 + i; h[64 + i] = 2.0; h[128 + i] = 0.5; }
    hipMemcpy(d, h.data(), 192 * 8, hipMemcpyHostToDevice);
    double *o; hipMalloc(&o, 512);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d, o, u);
    std::vector<double> ho(64); std::vector<unsigned> hu(128);
    hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost); hipMemcpy(hu.data(), u, 512, hipMemcpyDeviceToHost);
    printf("fmac_dpp row_newbcast:3 (expect 0.5 + 2 * (100 + 16 * (lane / 16) + 3)):");
    for (int i = 0; i < 64; i += 9) printf(" [%d] %.1f", i, ho[i]);
    printf("\npermlane16_swap x (was 1000 + lane):");
    for (int i = 0; i < 64; i += 8) printf(" [%d] %u", i, hu[i]);
    printf("\npermlane16_swap y (was 2000 + lane):");
    for (int i = 0; i < 64; i += 8) printf(" [%d] %u", i, hu[64 + i]);
    printf("\n");
    const int iters = 10000;
    const char *names[4] = {"8 x v_fmac_f64_dpp", "8 x v_fmac_f64", "8 x (2 readlane + fma)", "8 dependent v_fma_f64"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, d, c, iters);
            if (m == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, d, c, iters);
            if (m == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(64), 0, 0, d, c, iters);
            if (m == 3) hipLaunchKernelGGL(k_rate<3>, dim3(1), dim3(64), 0, 0, d, c, iters);
            hipDeviceSynchronize();
        }
        long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("%-26s %.1f clock64 ticks per group of 8\n", names[m], (double)hc / iters);
    }
    return 0;
}
