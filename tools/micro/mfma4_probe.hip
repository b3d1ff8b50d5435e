// Operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 and its block broadcast (cbsz / abid), found by one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
static void probe(double *da, double *db, double *dd) {
    printf("---- cbsz %d abid %d: for a one-hot A at lane la (B[l] = l + 1): result lanes and values\n", CBSZ, ABID);
    std::vector<double> a(64), b(64), d(64);
    for (int l = 0; l < 64; ++l) b[l] = l + 1;
    hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) a[l] = l == la ? 1.0 : 0.0;
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
        k<CBSZ, ABID><<<1, 64>>>(da, db, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        printf("la %2d:", la);
        for (int l = 0; l < 64; ++l)
            if (d[l] != 0.0) printf("  d[%2d]=B[%2d]", l, (int)d[l] - 1);
        printf("\n");
    }
}
int main() {
    double *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    probe<0, 0>(da, db, dd);
    probe<2, 0>(da, db, dd);
    probe<2, 1>(da, db, dd);
    probe<2, 3>(da, db, dd);
    probe<1, 0>(da, db, dd);
    probe<1, 1>(da, db, dd);
    return 0;
}
