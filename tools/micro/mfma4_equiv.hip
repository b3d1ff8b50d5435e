// Round 6: v_mfma_f64_16x16x4_f64 (one per ~106 clocks) rebuilt from FOUR v_mfma_f64_4x4x4_4b_f64 (one per ~17 clocks) on the SAME
// operand registers: the 4-block instruction forms the block-diagonal 4x4 products of the 16 x 16 tile (block = bits 2-3 of the
// lane, k = lane >> 4: the operand layout IS the 16x16x4 layout); rotating the A operand by 4, 8, 12 lanes inside its 16-lane rows
// (DPP row_ror) brings every A block to every B block.  This probe finds the result layout and checks the values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(const double *a, const double *b, double *ref, double *q, int steps) {
    const int l = threadIdx.x;
    v4d r = {0, 0, 0, 0};
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    for (int s = 0; s < steps; ++s) {
        const double x = a[s * 64 + l], y = b[s * 64 + l];
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, r, 0, 0, 0);
        const double x1 = dpp_f64<0x124>(x), x2 = dpp_f64<0x128>(x), x3 = dpp_f64<0x12C>(x);   // row_ror:4, 8, 12
        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y, p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, y, p2, 0, 0, 0);
        p3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, y, p3, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) ref[i * 64 + l] = r[i];
    q[l] = p0; q[64 + l] = p1; q[128 + l] = p2; q[192 + l] = p3;
}
int main() {
    const int steps = 8;
    std::vector<double> a(steps * 64), b(steps * 64), ref(256), q(256);
    srand(1);
    for (auto &v : a) v = rand() / (double)RAND_MAX - 0.5;
    for (auto &v : b) v = rand() / (double)RAND_MAX - 0.5;
    double *da, *db, *dr, *dq;
    hipMalloc(&da, a.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dr, 2048); hipMalloc(&dq, 2048);
    hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dr, dq, steps);
    hipMemcpy(ref.data(), dr, 2048, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, 2048, hipMemcpyDeviceToHost);
    // 16x16x4 result layout: ref[r][l] = P[4 r + (l >> 4)][l & 15].  Candidates for q[rot][l]: P[4 ((blk +- rot) & 3) + (l >> 4)][l & 15]
    for (int sign = -1; sign <= 1; sign += 2) {
        double md = 0; int bitdiff = 0;
        for (int rot = 0; rot < 4; ++rot)
            for (int l = 0; l < 64; ++l) {
                const int blk = (l >> 2) & 3, rt = (blk + sign * rot) & 3;
                const double want = ref[rt * 64 + l], got = q[rot * 64 + l];
                md = fmax(md, fabs(want - got));
                bitdiff += want != got;
            }
        printf("row group of register rot in a lane = (blk %c rot) & 3: max |diff| %.3e, differing values %d of 256\n", sign > 0 ? '+' : '-', md, bitdiff);
    }
    return 0;
}
