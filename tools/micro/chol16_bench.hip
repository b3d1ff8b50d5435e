// s64_chol16 (lsq_small64.h: v_readlane pairs + fma with an SGPR operand) against a form whose updates are single
// v_fmac_f64_dpp row_newbcast instructions: same bits; time of each in clock64 ticks and in wall-clock nanoseconds.
// Measured on MI355X: 4434 ticks (1.85 us) for the readlane form, 4797 (2.02 us) for the DPP form -- not adopted.    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../leastsquaresoptim.jl_amd/csrc chol16_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "lsq_small64.h"

// ---- cross-lane helpers of the register-resident 16 x 16 factorisations ------------------------------------------------
// A single wavefront issues its instructions one after the other (a v_fma_f64 every 8 clocks, measured:
// tools/micro/dpp_probe.hip), so such a kernel costs (instructions) x 8 clocks and the way to make it faster is fewer
// instructions.  Broadcasting element i of a row held one-entry-per-lane used to take two v_readlane_b32 + the fma with an
// SGPR operand (22 clocks per update, measured); v_fmac_f64_dpp with row_newbcast:i reads lane i of the lane's own 16-lane
// row as its first factor -- one instruction, 9.5 clocks.  The rows of a wavefront do not see each other through DPP, so
// what the identity lanes (row 1) need from the matrix lanes (row 0) is copied across with v_permlane16_swap_b32 (gfx950).
// The hazard rules (VALU write -> DPP / permlane read: 2 wait states) are not applied to inline asm by the compiler: the
// s_nop 1 in front of an instruction whose DPP source may have just been written is part of the asm.
template <int I, bool FRESH>     // d += r[lane I of this lane's row] * x;  FRESH: r may have been written by the previous instruction
__device__ __forceinline__ void s64_fmac_rowbcast(double &d, double r, double x) {
    if (FRESH)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(r), "v"(x), "n"(I));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(r), "v"(x), "n"(I));
}
template <int I>                 // r[lane I of this lane's row]
__device__ __forceinline__ double s64_rowbcast(double r) {
    double a;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(r), "n"(I));
    return a;
}
// rows (16 lanes) 0 and 2 of x, duplicated into rows 1 and 3
__device__ __forceinline__ double s64_dup_rows(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x), tl = lo, th = hi;
    // v_permlane16_swap vdst, src: rows 1, 3 of vdst <-> rows 0, 2 of src (checked: tools/micro/dpp_probe.hip)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo), "+v"(tl), "+v"(hi), "+v"(th));
    return __hiloint2double(hi, lo);
}

// ---- 16 x 16 diagonal block of a Cholesky factorisation G = U'U, ONE wavefront, registers ----------------------
// lanes 0..15: column c of the block (rows r <= c); lanes 16..31: column c of the identity, which the same row
// operations turn into inv(U_kk)' (lower triangular).  Writes U_kk (zeros below the diagonal) into M in place and
// inv(U_kk)' into W at the same position.  Returns 0, or 1 + the index (inside the block) of the first pivot that is not
// positive (wave-uniform).
// Pivot j: the (unscaled) row j of the matrix lanes is duplicated into the identity lanes' row, its element j broadcast,
// 1/sqrt of it formed in every lane, row j scaled, and every later row i gets  u_i -= U[j][i] * u_j  as ONE
// v_fmac_f64_dpp: 13 + (15 - j) instructions per pivot instead of ~50 (1.1 us for the block instead of 2.5 us).  The
// arithmetic -- and so every bit of the result -- is that of the v_readlane form it replaces.
template <int J, int I>
__device__ __forceinline__ void chol16_dpp_updates(double (&u)[16], double rn) {
    if constexpr (I < 16) {
        s64_fmac_rowbcast<I, I == J + 1>(u[I], rn, u[J]);      // u_I -= U[J][I] * U[J][c]
        chol16_dpp_updates<J, I + 1>(u, rn);
    }
}
template <int J>
__device__ __forceinline__ void chol16_dpp_pivots(double (&u)[16], int &badv) {
    if constexpr (J < 16) {
        const double r0 = s64_dup_rows(u[J]);                   // matrix row J, in the matrix AND the identity lanes
        const double ajj = s64_rowbcast<J>(r0);
        badv = (badv == 0 && !(ajj > 0.0)) ? J + 1 : badv;      // (such a pivot turns the rest into NaN / Inf: reported, not used)
        const double sj = s64_rsqrt(ajj);
        u[J] = u[J] * sj;                                       // U[J][c]; identity lanes: row J of inv(U)'
        chol16_dpp_updates<J, J + 1>(u, -(r0 * sj));
        chol16_dpp_pivots<J + 1>(u, badv);
    }
}
__device__ __forceinline__ int chol16_dpp(double *__restrict__ M, double *__restrict__ W, int o, int lane) {
    const int c = lane & 15;
    const bool mat = lane < 16, idn = lane >= 16 && lane < 32;
    double u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = mat ? (r <= c ? M[(o + r) * S64_LS + o + c] : 0.0) : ((idn && r == c) ? 1.0 : 0.0);
    int badv = 0;                                // per lane; lane 0 (a matrix lane) is the one that is read
    chol16_dpp_pivots<0>(u, badv);
    if (mat) {
#pragma unroll
        for (int r = 0; r < 16; ++r) M[(o + r) * S64_LS + o + c] = r <= c ? u[r] : 0.0;
    } else if (idn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) W[(o + r) * S64_LS + o + c] = c <= r ? u[r] : 0.0;
    }
    return __builtin_amdgcn_readfirstlane(badv);
}


template <int WHICH>
__global__ void k(const double *G, double *out, int *bad, long long *ticks, int reps) {
    __shared__ double M[S64_MAT], W[S64_MAT];
    const int lane = threadIdx.x;
    long long tc = 0, tw = 0;
    int b = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = lane; e < 64 * 64; e += 64) { M[(e >> 6) * S64_LS + (e & 63)] = G[e]; W[(e >> 6) * S64_LS + (e & 63)] = 0.0; }
        __syncthreads();
        const long long c0 = clock64(), w0 = wall_clock64();
        b = WHICH ? chol16_dpp(M, W, 16, lane) : s64_chol16(M, W, 16, lane);
        __syncthreads();
        tc += clock64() - c0;
        tw += wall_clock64() - w0;
    }
    for (int e = lane; e < 64 * 64; e += 64) { out[e] = M[(e >> 6) * S64_LS + (e & 63)]; out[4096 + e] = W[(e >> 6) * S64_LS + (e & 63)]; }
    if (lane == 0) { *bad = b; ticks[0] = tc; ticks[1] = tw; }
}

int main() {
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> A(64 * 80), G(4096);
    int fails = 0;
    double *dG, *dO; int *dB; long long *dT;
    hipMalloc(&dG, 4096 * 8); hipMalloc(&dO, 8192 * 8); hipMalloc(&dB, 4); hipMalloc(&dT, 16);
    for (int trial = 0; trial < 4; ++trial) {
        for (auto &a : A) a = nd(rng);
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = 0;
                for (int k = 0; k < 80; ++k) s += A[i * 80 + k] * A[j * 80 + k];
                G[i * 64 + j] = s;
            }
        if (trial == 2) G[(16 + 5) * 64 + 16 + 5] = -1.0;          // a pivot that is not positive
        if (trial == 3) for (int j = 0; j < 64; ++j) G[(16 + 7) * 64 + j] = G[j * 64 + 16 + 7] = 0.0;   // a zero row / column
        hipMemcpy(dG, G.data(), 4096 * 8, hipMemcpyHostToDevice);
        std::vector<double> o0(8192), o1(8192);
        int b0, b1; long long t0[2], t1[2];
        const int reps = 200;
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dG, dO, dB, dT, reps); hipDeviceSynchronize();
        hipMemcpy(o0.data(), dO, 8192 * 8, hipMemcpyDeviceToHost); hipMemcpy(&b0, dB, 4, hipMemcpyDeviceToHost); hipMemcpy(t0, dT, 16, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dG, dO, dB, dT, reps); hipDeviceSynchronize();
        hipMemcpy(o1.data(), dO, 8192 * 8, hipMemcpyDeviceToHost); hipMemcpy(&b1, dB, 4, hipMemcpyDeviceToHost); hipMemcpy(t1, dT, 16, hipMemcpyDeviceToHost);
        int diff = 0;
        for (int e = 0; e < 8192; ++e) diff += std::memcmp(&o0[e], &o1[e], 8) != 0;
        printf("trial %d: bad %d / %d, entries that differ %d;  readlane form %.0f ticks = %.0f ns, dpp form %.0f ticks = %.0f ns\n", trial, b0, b1,
               diff, (double)t0[0] / reps, t0[1] * 10.0 / reps, (double)t1[0] / reps, t1[1] * 10.0 / reps);
        fails += (b0 != b1) || (b0 == 0 && diff != 0);
    }
    printf(fails ? "MISMATCH\n" : "same bits\n");
    return fails != 0;
}
