// Does hipExtStreamCreateWithCUMask partition the device on this box?  Two masked streams (the first R mask bits / the rest);
// a kernel on each records, per workgroup, the XCC and CU it ran on (HW_ID / XCC_ID registers) -> the two sets must be disjoint.
// Second part: a streaming kernel on the big partition with and without a latency-bound chain beside it on the small one.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/cumask_probe.hip -o build/micro/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}
__global__ void __launch_bounds__(256) k_stream(const double *__restrict__ a, double *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) b[i] = a[i] * 1.0000001;
}
__global__ void __launch_bounds__(256) k_chain(double *x, int iters) {      // one workgroup, dependent LDS/ALU chain
    __shared__ double s[256];
    s[threadIdx.x] = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) { __syncthreads(); double v = s[(threadIdx.x * 7 + i) & 255]; __syncthreads(); s[threadIdx.x] = v * 0.999 + 1e-3; }
    x[threadIdx.x] = s[threadIdx.x];
}
int main(int argc, char **argv) {
    int R = argc > 1 ? atoi(argv[1]) : 32;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("CUs %d, reserved mask bits %d\n", ncu, R);
    std::vector<uint32_t> ms((ncu + 31) / 32, 0), mb((ncu + 31) / 32, 0);
    for (int i = 0; i < ncu; ++i) (i < R ? ms : mb)[i / 32] |= 1u << (i % 32);
    hipStream_t ss, sb, plain;
    CK(hipExtStreamCreateWithCUMask(&ss, ms.size(), ms.data()));
    CK(hipExtStreamCreateWithCUMask(&sb, mb.size(), mb.data()));
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    unsigned *d; CK(hipMalloc(&d, 4096 * 4));
    std::vector<unsigned> h(4096);
    const char *names[3] = {"small", "big", "plain"};
    hipStream_t st[3] = {ss, sb, plain};
    std::set<unsigned> sets[3];
    for (int k = 0; k < 3; ++k) {
        CK(hipMemsetAsync(d, 0xff, 4096 * 4, st[k]));
        hipLaunchKernelGGL(k_where, dim3(2048), dim3(64), 0, st[k], d, 2000);
        CK(hipStreamSynchronize(st[k]));
        CK(hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost));
        int perx[16] = {0};
        for (int i = 0; i < 2048; ++i) {
            unsigned v = h[i], xcc = v >> 16, hw = v & 0xffff;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;      // gfx9 HW_ID: CU_ID 11:8, SH_ID 12, SE_ID 15:13
            unsigned key = xcc << 12 | se << 8 | sh << 4 | cu;
            if (sets[k].insert(key).second) perx[xcc]++;
        }
        printf("%-5s stream: %zu distinct CUs; per XCC:", names[k], sets[k].size());
        for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
        printf("\n");
    }
    int common = 0;
    for (unsigned k : sets[0]) common += sets[1].count(k);
    printf("CUs common to small and big: %d\n", common);
    // ---- interference: 512 MB copy on 'big' alone / with a chain on 'small' / both on plain streams
    size_t n = 64u << 20;
    double *a, *b, *x; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&x, 4096));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(x, 0, 4096));
    hipEvent_t e0, e1, c0, c1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    hipStream_t plain2; CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
    for (int mode = 0; mode < 5; ++mode) {
        // 0: copy on big alone; 1: copy on big + chain on small; 2: copy plain + chain plain2; 3: chain alone small; 4: copy on plain alone
        hipStream_t cs = (mode == 2 || mode == 4) ? plain : sb, ks = mode == 2 ? plain2 : ss;
        float best_c = 1e9, best_k = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            if (mode != 3) hipEventRecord(e0, cs);
            if (mode == 1 || mode == 2 || mode == 3) { hipEventRecord(c0, ks); hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, ks, x, 20000); hipEventRecord(c1, ks); }
            if (mode != 3) { for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, cs, a, b, n); hipEventRecord(e1, cs); }
            CK(hipDeviceSynchronize());
            float t;
            if (mode != 3) { hipEventElapsedTime(&t, e0, e1); best_c = t < best_c ? t : best_c; }
            if (mode == 1 || mode == 2 || mode == 3) { hipEventElapsedTime(&t, c0, c1); best_k = t < best_k ? t : best_k; }
        }
        printf("mode %d: 4 copies %.3f ms (%.2f TB/s)   chain %.3f ms\n", mode, best_c, mode == 3 ? 0.0 : 4.0 * n * 16 / best_c * 1e-9, best_k);
    }
    return 0;
}
