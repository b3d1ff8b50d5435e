// Phase timing of the CholeskyQR2 panel kernels (csrc/lsq_qr_cholqr.hip) on one 16384 x 64 panel.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DCQ_TIMING -I leastsquaresoptim.jl_amd/csrc tools/micro/cqr_bench.hip -o tools/micro/cqr_bench
#include "../../leastsquaresoptim.jl_amd/csrc/lsq_qr_cholqr.hip"
#include <vector>
int lsq_dbg_jitter_us = 0, lsq_dbg_serial = 0;
void lsq_dbg_stall() {}
void lsq_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
int main() {
    const int M = 16384, n = 128, c0 = 0;
    lsq_ctx c;
    c.device = 0; c.stream = nullptr; c.num_cus = 256;
    std::vector<double> A((size_t)M * n);
    srand(1);
    for (auto &v : A) v = ((double)rand() / RAND_MAX * 2 - 1) / 128.0;
    double *dA, *dVb; int *derr;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dVb, (size_t)M * 64 * 8); hipMalloc(&derr, 4); hipMemset(derr, 0, 4);
    CqrWork w;
    if (lsq_cqr_alloc(&c, &w, M) != LSQ_OK) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        if (lsq_cqr_panel(&c, &w, dA, M, c0, dVb, M - c0, derr, nullptr) != LSQ_OK) return 1;
        hipDeviceSynchronize();
    }
    unsigned long long t[64];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(cq_tbuf), sizeof(t));
    auto us = [&](int a, int b) { return (double)(t[b] - t[a]) * 0.01; };   // wall_clock64: 100 MHz
    printf("pass0: load %.2f  gram %.2f\n", us(0, 1), us(1, 2));
    for (int p = 1; p <= 2; ++p)
        printf("pass%d: load G + factor + inverse %.2f  R store %.2f  slab product %.2f  write-out %.2f  gram %.2f | total %.2f\n", p,
               us(16 * p, 16 * p + 2), us(16 * p + 2, 16 * p + 3), us(16 * p + 3, 16 * p + 4),
               us(16 * p + 4, 16 * p + 5), us(16 * p + 5, 16 * p + 6), us(16 * p, 16 * p + 6));
    if (t[52] > t[53] && t[53] > t[49])       // Neumann path of k_cqr_top (Q1 form, tall panel)
        printf("top (Neumann): G2 + factor + R + Q_top %.2f  E, E^2, decision, S R %.2f  squaring loop + inv(B) store %.2f | total %.2f\n",
               us(48, 49), us(49, 53), us(53, 52), us(48, 52));
    else
        printf("top: factor + R + Q_top %.2f  LU %.2f  S R + inv(L) + inv(U) %.2f  product %.2f | total %.2f\n", us(48, 49), us(49, 50), us(50, 51), us(51, 52), us(48, 52));
    int e; hipMemcpy(&e, derr, 4, hipMemcpyDeviceToHost); printf("err word %d\n", e);
    return 0;
}
