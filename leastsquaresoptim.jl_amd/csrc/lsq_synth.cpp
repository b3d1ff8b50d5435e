// Deterministic synthetic inputs: counter-based splitmix64 streams so that the CPU
// baseline and every GPU rank regenerate bit-identical problems from (seed, index) alone.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/lsqhip.h"

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t stream(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ (a * 0xD6E8FEB86659FD93ull)) + b);
}
static inline double u01(uint64_t r) { return ((r >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0,1)
static inline double normal(uint64_t seed, uint64_t a, uint64_t b) {  // Box-Muller
    double u1 = u01(stream(seed, a, 2 * b)), u2 = u01(stream(seed, a, 2 * b + 1));
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
}

extern "C" int lsq_synth_sparse(int m, int n, int per_col, unsigned long long seed, int *colptr, int *rowval,
                                double *nzval) {
    if (m <= 0 || n <= 0 || per_col <= 0 || per_col > m) return LSQ_EARG;
    const double scale = 1.0 / std::sqrt((double)per_col);
    std::vector<int> rows(per_col);
    // open-addressing set sized to a power of two >= 2*per_col
    size_t cap = 1;
    while (cap < (size_t)per_col * 2) cap <<= 1;
    std::vector<int> table(cap);
    colptr[0] = 0;
    for (int j = 0; j < n; ++j) {
        std::fill(table.begin(), table.end(), -1);
        int got = 0;
        uint64_t ctr = 0;
        while (got < per_col) {
            int r = (int)(u01(stream(seed, (uint64_t)j + 1, ctr++)) * m);
            if (r >= m) r = m - 1;
            size_t h = (size_t)(splitmix64((uint64_t)r) & (cap - 1));
            bool dup = false;
            while (table[h] != -1) {
                if (table[h] == r) { dup = true; break; }
                h = (h + 1) & (cap - 1);
            }
            if (dup) continue;
            table[h] = r;
            rows[got++] = r;
        }
        std::sort(rows.begin(), rows.end());
        const long long base = (long long)j * per_col;
        for (int k = 0; k < per_col; ++k) {
            rowval[base + k] = rows[k];
            nzval[base + k] = scale * normal(seed ^ 0xA5A5A5A5ull, (uint64_t)j + 1, (uint64_t)k);
        }
        colptr[j + 1] = (int)(base + per_col);
    }
    return LSQ_OK;
}

extern "C" int lsq_synth_dense(int m, int n, unsigned long long seed, double *v) {
    if (m <= 0 || n <= 0) return LSQ_EARG;
    const double scale = 1.0 / std::sqrt((double)m);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) v[(size_t)j * m + i] = scale * normal(seed, (uint64_t)j + 1, (uint64_t)i);
    return LSQ_OK;
}

extern "C" int lsq_synth_uniform(int n, unsigned long long seed, double lo, double hi, double *out) {
    for (int i = 0; i < n; ++i) out[i] = lo + (hi - lo) * u01(stream(seed, 0x51ull, (uint64_t)i));
    return LSQ_OK;
}

extern "C" int lsq_synth_normal(int n, unsigned long long seed, double *out) {
    for (int i = 0; i < n; ++i) out[i] = normal(seed, 0x77ull, (uint64_t)i);
    return LSQ_OK;
}
