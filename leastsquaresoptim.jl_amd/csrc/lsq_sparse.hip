// Jacobian handles (dense / CSC + CSR mirror), the generic operator products mul!(y,J,x,a,b) and
// mul!(x,J',y,a,b), and colsumabs2! for sparse Jacobians (utils.jl:146-151).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "lsq_spmv.h"

// ---------------------------------------------------------------------------------------------
// plan construction (host, once per pattern)
// ---------------------------------------------------------------------------------------------
static int choose_plan(const char *envname, long long nnz, int nseg, int group_hint = 0) {
    if (const char *e = getenv(envname)) {
        if (!strcmp(e, "stream")) return LSQ_PLAN_STREAM;
        if (!strcmp(e, "wave")) return LSQ_PLAN_WAVE;
        if (!strcmp(e, "block")) return LSQ_PLAN_BLOCK;
    }
    if (group_hint > 0 && nseg > 0 && (double)nnz / nseg < 48.0) return LSQ_PLAN_STREAM;
    double avg = nseg > 0 ? (double)nnz / nseg : 0.0;
    if (avg < 48.0) return LSQ_PLAN_STREAM;
    if (avg < 384.0) return LSQ_PLAN_WAVE;
    return LSQ_PLAN_BLOCK;
}

// `group` > 0: tiles never straddle a multiple of `group` segments (window boundaries)
static void build_tiles(const std::vector<int> &ptr, int nseg, std::vector<int> &tiles, int group = 0,
                        int max_segs = LSQ_TILE_SEGS, int max_nnz = LSQ_TILE_NNZ) {
    tiles.clear();
    int s = 0;
    tiles.push_back(0);
    while (s < nseg) {
        int start = s;
        long long base = ptr[s];
        int limit = group > 0 ? std::min(nseg, (start / group + 1) * group) : nseg;
        while (s < limit && s - start < max_segs && ptr[s + 1] - base <= max_nnz) ++s;
        if (s == start) ++s;  // one segment longer than a tile: handled by the block-stride path
        tiles.push_back(s);
    }
}

// XCD-aware placement: work item k (a tile, or a block of 4 segments) of window w should run on
// XCD w % 8, and the dispatcher places block b on XCD b % 8 (observed, used for speed only): deal
// the work items of each residue class out to the grid positions of that residue.
static void build_order(const std::vector<int> &item_window, std::vector<int> &order) {
    const int nw = (int)item_window.size();
    std::vector<std::vector<int>> byx(8);
    for (int k = 0; k < nw; ++k) byx[item_window[k] % 8].push_back(k);
    order.assign(nw, -1);
    std::vector<size_t> next(8, 0);
    std::vector<int> leftover;
    for (int p = 0; p < nw; ++p) {
        int x = p % 8;
        if (next[x] < byx[x].size()) order[p] = byx[x][next[x]++];
    }
    for (int x = 0; x < 8; ++x)
        for (size_t k = next[x]; k < byx[x].size(); ++k) leftover.push_back(byx[x][k]);
    size_t q = 0;
    for (int p = 0; p < nw; ++p)
        if (order[p] < 0) order[p] = leftover[q++];
}

static int upload_segs(lsq_ctx *c, LsqSegs &S, const std::vector<int> &ptr, const std::vector<int> &idx,
                       const char *envname, int group = 0) {
    S.nseg = (int)ptr.size() - 1;
    S.nnz = ptr.back();
    S.plan = choose_plan(envname, S.nnz, S.nseg);
    const size_t pad = 8;
    LSQ_HIP(hipMalloc(&S.d_ptr, (S.nseg + 1) * sizeof(int)));
    LSQ_HIP(hipMalloc(&S.d_idx, (S.nnz + pad) * sizeof(int)));
    LSQ_HIP(hipMalloc(&S.d_val, (S.nnz + pad) * sizeof(double)));
    LSQ_ZERO(S.d_idx, 0, (S.nnz + pad) * sizeof(int));
    LSQ_ZERO(S.d_val, 0, (S.nnz + pad) * sizeof(double));
    LSQ_HIP(hipMemcpy(S.d_ptr, ptr.data(), (S.nseg + 1) * sizeof(int), hipMemcpyHostToDevice));
    if (S.nnz) LSQ_HIP(hipMemcpy(S.d_idx, idx.data(), S.nnz * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> tiles;
    build_tiles(ptr, S.nseg, tiles, group);
    S.ntiles = (int)tiles.size() - 1;
    LSQ_HIP(hipMalloc(&S.d_tiles, tiles.size() * sizeof(int)));
    LSQ_HIP(hipMemcpy(S.d_tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
    // big tiles for the LDS-staged stream kernel (only worth it on large patterns)
    if (S.plan == LSQ_PLAN_STREAM && group == 0 && S.nnz >= (1 << 20)) {
        // size the big tiles so that every persistent workgroup (one per CU) gets the same count
        std::vector<int> big;
        long long per_round = (long long)LSQ_BIG_NNZ * c->num_cus;
        long long rounds = (S.nnz + per_round - 1) / per_round;
        int target = (int)std::min<long long>(LSQ_BIG_NNZ, S.nnz / (rounds * c->num_cus) + 16);
        build_tiles(ptr, S.nseg, big, 0, LSQ_BIG_SEGS, std::max(target, 1024));
        S.nbig = (int)big.size() - 1;
        for (int t = 0; t < S.nbig && S.nbig > 0; ++t)
            if (ptr[big[t + 1]] - ptr[big[t]] > LSQ_BIG_NNZ) S.nbig = 0;  // a segment too long for LDS tiles
        std::vector<int> bm((size_t)4 * S.nbig + 4);
        for (int t = 0; t < S.nbig; ++t) {
            bm[4 * t + 0] = big[t];
            bm[4 * t + 1] = big[t + 1];
            bm[4 * t + 2] = ptr[big[t]];
            bm[4 * t + 3] = ptr[big[t + 1]];
        }
        LSQ_HIP(hipMalloc(&S.d_big, (bm.size() + 4) * sizeof(int)));
        LSQ_HIP(hipMemcpy(S.d_big, bm.data(), bm.size() * sizeof(int), hipMemcpyHostToDevice));
        if (S.nbig > 0 && S.nx <= 65535) {  // 10 B/nnz instead of 12
            std::vector<unsigned short> i16(S.nnz + pad, 0);
            for (long long k = 0; k < S.nnz; ++k) i16[k] = (unsigned short)idx[k];
            LSQ_HIP(hipMalloc(&S.d_idx16, i16.size() * sizeof(unsigned short)));
            LSQ_HIP(hipMemcpy(S.d_idx16, i16.data(), i16.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        }
    }
    if (group > 0 && S.plan != LSQ_PLAN_BLOCK) {
        std::vector<int> win, order;
        if (S.plan == LSQ_PLAN_STREAM) {
            for (int t = 0; t < S.ntiles; ++t) win.push_back(tiles[t] / group);
        } else {
            const int per = LSQ_NT / 64;
            for (int b = 0; b * per < S.nseg; ++b) win.push_back((b * per) / group);
        }
        build_order(win, order);
        LSQ_HIP(hipMalloc(&S.d_order, (order.size() + 1) * sizeof(int)));
        LSQ_HIP(hipMemcpy(S.d_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    (void)c;
    return LSQ_OK;
}

static void free_segs(LsqSegs &S) {
    hipFree(S.d_ptr);
    hipFree(S.d_idx);
    hipFree(S.d_idx16);
    hipFree(S.d_col16);
    hipFree(S.d_val);
    hipFree(S.d_tiles);
    hipFree(S.d_order);
    hipFree(S.d_big);
    hipFree(S.d_wtile);
    S = LsqSegs();
}

// ---------------------------------------------------------------------------------------------
// sliced layouts (lsq_sell.h)
// ---------------------------------------------------------------------------------------------
// Segments [first, first+count) of `ptr` are the outputs of block b (first/count from seg_range);
// entry e of a segment gathers with index idx16_of(e), comes from CSC position srcmap[e] and
// belongs to column col_of(e) (only stored when want_col16).
// skip_empty: segments without entries get no lane (the kernel must not rely on every output of a block being written).
template <class SegRange, class Idx16, class ColOf>
static int build_sell(LsqSell &S, int nblocks, const std::vector<int> &ptr, const std::vector<int> &srcmap,
                      SegRange seg_range, Idx16 idx16_of, ColOf col_of, bool want_col16, bool skip_empty = false, bool allow_odd = true) {
    std::vector<int> map;
    std::vector<int2> smeta;
    std::vector<unsigned> info;
    std::vector<unsigned short> idx16, col16;
    long long nstore = 0;
    std::vector<int> ord;
    // every block gets the same number of slices (empty ones at the end of the shorter blocks): a kernel then knows
    // its first slice from its block index alone, one dependent load less at the head of every launch
    int spw = 0;
    auto block_order = [&](int b, int &first) {   // the block's segments that get a lane, in their original order
        int count;
        seg_range(b, first, count);
        ord.clear();
        for (int i = 0; i < count; ++i)
            if (!skip_empty || ptr[first + i + 1] > ptr[first + i]) ord.push_back(i);
        return (int)ord.size();
    };
    for (int b = 0; b < nblocks; ++b) {
        int first;
        spw = std::max(spw, (block_order(b, first) + 63) / 64);
    }
    for (int b = 0; b < nblocks; ++b) {
        int first;
        const int count = block_order(b, first);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int c2) {
            return ptr[first + a + 1] - ptr[first + a] > ptr[first + c2 + 1] - ptr[first + c2];
        });
        const int ngroups = (count + 63) / 64;
        for (int slot = 0; slot < ngroups; ++slot) {
            // snake order: the 16 waves visit slices s0+wave, s0+wave+16, ...; alternate the direction of
            // the deal every round so that every wave gets the same share of long and short slices
            const int round = slot / 16, r = slot % 16;
            const bool full = (round + 1) * 16 <= ngroups;
            const int g = round * 16 + ((round & 1) && full ? 15 - r : r);   // sorted group stored at this slot
            int L = 0;
            for (int l = 0; l < 64 && g * 64 + l < count; ++l) {
                const int sg = first + ord[g * 64 + l];
                L = std::max(L, ptr[sg + 1] - ptr[sg]);
            }
            // (round 5: L is NOT rounded up to even any more.  An odd slice ends with one UNPAIRED entry per lane, stored as a
            //  compact group of 64 values behind the pairs -- at 3.3 entries per sub-row (wide n) the rounding was 13 % of the
            //  stream, on C4's rows 4.7 %.  lsq_sell.h: sell_tail_*)
            if (!allow_odd) L = (L + 1) & ~1;        // (the column-windowed J*v kernel: lsq_sell.h, k_sell_rows_wide)
            if (nstore + (long long)L * 64 > 2147480000LL) return LSQ_EDIM;
            const long long off = nstore;
            nstore += (long long)L * 64;
            map.resize(nstore, -1);
            idx16.resize(nstore, 0);
            if (want_col16) col16.resize(nstore, 0);
            smeta.push_back(make_int2((int)off, L));
            for (int l = 0; l < 64; ++l) {
                if (g * 64 + l >= count) {
                    info.push_back(LSQ_SELL_POS_MASK);
                    continue;
                }
                const int pos = ord[g * 64 + l], sg = first + pos;
                const int len = ptr[sg + 1] - ptr[sg];
                if (len >= (1 << (32 - LSQ_SELL_POS_BITS))) return LSQ_EDIM;
                info.push_back((unsigned)pos | ((unsigned)len << LSQ_SELL_POS_BITS));
                for (int j = 0; j < len; ++j) {
                    const int e = ptr[sg] + j;
                    const size_t slot_e = (L & 1) && j == L - 1 ? (size_t)off + (size_t)(L >> 1) * 128 + l
                                                                : (size_t)off + ((size_t)(j / 2) * 64 + l) * 2 + (j & 1);
                    map[slot_e] = srcmap[e];
                    idx16[slot_e] = idx16_of(e);
                    if (want_col16) col16[slot_e] = col_of(e);
                }
            }
        }
        for (int slot = ngroups; slot < spw; ++slot) {      // empty slices: no entries, no outputs
            smeta.push_back(make_int2((int)nstore, 0));
            for (int l = 0; l < 64; ++l) info.push_back(LSQ_SELL_POS_MASK);
        }
    }
    S.spw = spw;
    S.nblocks = nblocks;
    S.nslices = (int)smeta.size();
    S.nstore = nstore;
    const size_t pad = 1024;   // the unrolled loads of the last slice may run a few groups past the end
    map.resize(nstore + pad, -1);
    idx16.resize(nstore + pad, 0);
    LSQ_HIP(hipMalloc(&S.d_smeta, (smeta.size() + 1) * sizeof(int2)));
    LSQ_HIP(hipMemcpy(S.d_smeta, smeta.data(), smeta.size() * sizeof(int2), hipMemcpyHostToDevice));
    LSQ_HIP(hipMalloc(&S.d_info, (info.size() + 64) * sizeof(unsigned)));
    LSQ_HIP(hipMemcpy(S.d_info, info.data(), info.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    LSQ_HIP(hipMalloc(&S.d_idx16, idx16.size() * sizeof(unsigned short)));
    LSQ_HIP(hipMemcpy(S.d_idx16, idx16.data(), idx16.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    if (want_col16) {
        col16.resize(nstore + pad, 0);
        LSQ_HIP(hipMalloc(&S.d_col16, col16.size() * sizeof(unsigned short)));
        LSQ_HIP(hipMemcpy(S.d_col16, col16.data(), col16.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    LSQ_HIP(hipMalloc(&S.d_map, map.size() * sizeof(int)));
    LSQ_HIP(hipMemcpy(S.d_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    LSQ_HIP(hipMalloc(&S.d_val, (nstore + pad) * sizeof(double)));
    LSQ_ZERO(S.d_val, 0, (nstore + pad) * sizeof(double));
    S.active = true;
    return LSQ_OK;
}

static void free_sell(LsqSell &S) {
    hipFree(S.d_smeta); hipFree(S.d_info); hipFree(S.d_idx16); hipFree(S.d_col16);
    hipFree(S.d_val); hipFree(S.d_map); hipFree(S.d_part); hipFree(S.d_sx);
    S = LsqSell();
}

static int csc_create_impl(lsq_ctx *c, int m, int n, const int *colptr, const int *rowval, lsq_mat **out);
extern "C" int lsq_csc_create(lsq_ctx *c, int m, int n, const int *colptr, const int *rowval, lsq_mat **out) {
    LSQ_RANGE("lsq_csc_create");
    // the layouts are built in host vectors of up to a few hundred MB: an allocation failure must come back as a status, not
    // as a C++ exception through the C ABI
    try {
        return csc_create_impl(c, m, n, colptr, rowval, out);
    } catch (const std::bad_alloc &) {
        lsq_set_error("lsq_csc_create: out of host memory while building the layouts of a %d x %d matrix", m, n);
        return LSQ_EHIP;
    } catch (const std::exception &e) {
        lsq_set_error("lsq_csc_create: %s", e.what());
        return LSQ_EHIP;
    }
}
static int csc_create_impl(lsq_ctx *c, int m, int n, const int *colptr, const int *rowval, lsq_mat **out) {
    if (!c || !out || m < 0 || n < 0 || !colptr) {
        lsq_set_error("lsq_csc_create: bad arguments");
        return LSQ_EARG;
    }
    LSQ_HIP(hipSetDevice(c->device));
    const long long nnz = colptr[n];
    if (colptr[0] != 0 || nnz < 0 || nnz > 2147483000LL) {
        lsq_set_error("lsq_csc_create: colptr must start at 0 and nnz must fit int32");
        return LSQ_EDIM;
    }
    for (int j = 0; j < n; ++j)
        if (colptr[j + 1] < colptr[j]) {
            lsq_set_error("lsq_csc_create: colptr not monotone at column %d", j);
            return LSQ_EDIM;
        }
    for (long long k = 0; k < nnz; ++k)
        if (rowval[k] < 0 || rowval[k] >= m) {
            lsq_set_error("lsq_csc_create: row index out of range at entry %lld", k);
            return LSQ_EDIM;
        }
    lsq_mat *J = new lsq_mat();
    J->ctx = c;
    J->kind = LSQ_MAT_CSC;
    J->m = m;
    J->n = n;
    J->nnz = nnz;
    // CSR mirror by counting sort; column order inside a row is increasing (stable)
    std::vector<int> cptr(colptr, colptr + n + 1), cidx(rowval, rowval + nnz);
    std::vector<int> rptr(m + 1, 0), ridx(nnz), map(nnz);
    for (long long k = 0; k < nnz; ++k) rptr[rowval[k] + 1]++;
    for (int i = 0; i < m; ++i) rptr[i + 1] += rptr[i];
    {
        std::vector<int> fill(rptr.begin(), rptr.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int k = colptr[j]; k < colptr[j + 1]; ++k) {
                int p = fill[rowval[k]]++;
                ridx[p] = j;
                map[p] = k;
            }
    }
    J->csc.nx = m;
    J->csr.nx = n;
    J->bcsc.nx = m;
    LSQ_TRY(upload_segs(c, J->csc, cptr, cidx, "LSQ_PLAN_CSC"));
    LSQ_TRY(upload_segs(c, J->csr, rptr, ridx, "LSQ_PLAN_CSR"));
    LSQ_HIP(hipMalloc(&J->d_map, (nnz + 4) * sizeof(int)));
    if (nnz) LSQ_HIP(hipMemcpy(J->d_map, map.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
    LSQ_HIP(hipMalloc(&J->d_colsum, (n > 0 ? n : 1) * sizeof(double)));
    const bool sell_force = getenv("LSQ_SELL_FORCE") != nullptr;   // tests: sliced layouts on small patterns too
    const bool sell_ok = !getenv("LSQ_NO_SELL") && (sell_force || nnz >= (1 << 20)) && nnz > 0;
    // sliced rows for J*x: the gather vector in LDS next to the 4096-row output window -- all of it when n <= LSQ_LDS_X_MAX,
    // else one COLUMN WINDOW of it at a time (k_sell_rows_wide: a row block's workgroup walks the windows in ascending
    // order and keeps the rows' running sums in the output window, so a row is still summed left to right)
    int xmax = LSQ_LDS_X_MAX;
    if (const char *e = getenv("LSQ_SELL_XMAX")) xmax = std::min(LSQ_LDS_X_MAX, std::max(64, atoi(e)));   // tests: narrow windows
    else if (n > LSQ_LDS_X_MAX) xmax = LSQ_SELL_WIDE_X_MAX;      // windows may be a little wider than what the one-window kernels hold
    const int ncw = n >= 1 ? (n + xmax - 1) / xmax : 1;
    // a window pass costs a row block ~4 us whatever it holds (two barriers, one exposed round trip), so the windows pay while
    // every pass still moves >= 60 KB per CU.  MI355X, nnz = 1e7, m = 1e6: n = 25000 (3 windows) 32.7 us against 49.6 us with
    // the segment kernel (x gathered through L1/L2), n = 50000 (5) 40.7 against 51.9, n = 100000 (9) 56.0 against 53.2;
    // 200000 x 50000 with nnz 2e6: 26 against 13 us.
    const bool wide_pays = ncw <= LSQ_SELL_CW_AUTO && nnz / c->num_cus >= (long long)ncw * 6000;
    const bool wide_ok = ncw <= LSQ_SELL_CW_MAX && (long long)ncw * m < 64000000LL && !getenv("LSQ_NO_SELL_WIDE") &&
                         (wide_pays || sell_force || getenv("LSQ_SELL_WIDE"));
    if (sell_ok && J->csr.plan == LSQ_PLAN_STREAM && n >= 1 && (ncw == 1 || wide_ok)) {
        int per_cu = (m + c->num_cus - 1) / c->num_cus;
        int rounds = (per_cu + LSQ_SELL_ROWS_MAX - 1) / LSQ_SELL_ROWS_MAX;
        int wrows = (m + rounds * c->num_cus - 1) / (rounds * c->num_cus);
        wrows = std::min(LSQ_SELL_ROWS_MAX, (wrows + 63) & ~63);
        if (const char *e = getenv("LSQ_SELL_ROWS")) wrows = std::min(LSQ_SELL_ROWS_MAX, std::max(64, atoi(e) & ~63));
        const int nrb = (m + wrows - 1) / wrows;
        int st;
        if (ncw == 1) {
            st = build_sell(
                J->srows, nrb, rptr, map,
                [&](int b, int &first, int &count) { first = b * wrows; count = std::min(wrows, m - first); },
                [&](int e) { return (unsigned short)ridx[e]; }, [&](int e) { return (unsigned short)ridx[e]; }, false, false,
                /*allow_odd=*/!getenv("LSQ_SELL_EVEN_ROWS"));
        } else {
            // sub-rows (window, row): entries of a row keep their column order inside every window
            const int cwidth = (((n + ncw - 1) / ncw) + 1) & ~1;
            std::vector<int> wptr((size_t)ncw * m + 1, 0), widx(nnz), wmap(nnz);
            for (int i = 0; i < m; ++i)
                for (int e = rptr[i]; e < rptr[i + 1]; ++e) wptr[(size_t)(ridx[e] / cwidth) * m + i + 1]++;
            for (size_t s2 = 0; s2 < (size_t)ncw * m; ++s2) wptr[s2 + 1] += wptr[s2];
            // (a row's entries are sorted by column, so its entries of one window are consecutive: a running offset per
            //  (row, window) instead of a second ncw * m array of fill positions)
            for (int i = 0; i < m; ++i) {
                int w_cur = -1, off = 0;
                for (int e = rptr[i]; e < rptr[i + 1]; ++e) {
                    const int w = ridx[e] / cwidth;
                    if (w != w_cur) { w_cur = w; off = 0; }
                    const int p = wptr[(size_t)w * m + i] + off++;
                    widx[p] = ridx[e] % cwidth;
                    wmap[p] = map[e];
                }
            }
            st = build_sell(
                J->srows, nrb * ncw, wptr, wmap,
                [&](int b, int &first, int &count) {
                    const int rb = b / ncw, cw = b % ncw;
                    first = cw * m + rb * wrows;
                    count = std::min(wrows, m - rb * wrows);
                },
                [&](int e) { return (unsigned short)widx[e]; }, [&](int e) { return (unsigned short)widx[e]; }, false, true,
                /*allow_odd=*/false);
            J->srows.ncw = ncw;
            J->srows.cwidth = cwidth;
            if (st == LSQ_OK) LSQ_HIP(hipMalloc(&J->srows.d_sx, (size_t)n * sizeof(double)));
        }
        if (st == LSQ_OK) {
            J->srows.wrows = wrows;
            hipFree(J->csr.d_val);   // the sliced layout carries the values; misuse of the mirror must fail loudly
            J->csr.d_val = nullptr;
        } else if (st != LSQ_EDIM) {
            return st;
        } else {
            free_sell(J->srows);
        }
    }
    // sliced columns for J'*y: gather windows of <= 16384 rows of y in LDS, <= 2560 output columns per block (lsq_sell.h)
    if (sell_ok && (sell_force || m > 131072) && n >= 1 && !getenv("LSQ_PLAN_BCSC") && !getenv("LSQ_WINDOW_ROWS")) {
        const int ncb = (n + LSQ_SELL_CCOLS_MAX - 1) / LSQ_SELL_CCOLS_MAX;
        const int ccols = (n + ncb - 1) / ncb;
        const int ngw_min = (m + LSQ_SELL_GROWS_MAX - 1) / LSQ_SELL_GROWS_MAX;
        const int k = std::max(1, (int)(((long long)ngw_min * ncb + c->num_cus - 1) / c->num_cus));
        int ngw = std::max(ngw_min, (k * c->num_cus) / ncb);
        int grows = (m + ngw - 1) / ngw;
        grows = std::min(LSQ_SELL_GROWS_MAX, (grows + 7) & ~7);
        if (const char *e = getenv("LSQ_SELL_GROWS")) grows = std::min(LSQ_SELL_GROWS_MAX, std::max(64, atoi(e) & ~7));
        ngw = (m + grows - 1) / grows;
        if ((long long)ngw * n < 200000000LL) {
            std::vector<int> gptr((size_t)ngw * n + 1, 0), gidx(nnz), gmap(nnz);
            for (int j = 0; j < n; ++j)
                for (int k2 = colptr[j]; k2 < colptr[j + 1]; ++k2) gptr[(size_t)(rowval[k2] / grows) * n + j + 1]++;
            for (size_t s2 = 0; s2 < (size_t)ngw * n; ++s2) gptr[s2 + 1] += gptr[s2];
            {
                std::vector<int> fill(gptr.begin(), gptr.end() - 1);
                for (int j = 0; j < n; ++j)
                    for (int k2 = colptr[j]; k2 < colptr[j + 1]; ++k2) {
                        int p = fill[(size_t)(rowval[k2] / grows) * n + j]++;
                        gidx[p] = rowval[k2];
                        gmap[p] = k2;
                    }
            }
            std::vector<unsigned short> gcol(nnz);
            for (size_t sg = 0; sg < (size_t)ngw * n; ++sg)
                for (int e = gptr[sg]; e < gptr[sg + 1]; ++e) gcol[e] = (unsigned short)(sg % n);
            int st = build_sell(
                J->scols, ngw * ncb, gptr, gmap,
                [&](int b, int &first, int &count) {
                    const int gw = b / ncb, cb = b % ncb;
                    first = gw * n + cb * ccols;
                    count = std::max(0, std::min(ccols, n - cb * ccols));
                },
                [&](int e) { return (unsigned short)(gidx[e] % grows); }, [&](int e) { return gcol[e]; }, n <= 65535, false,
                /*allow_odd=*/!getenv("LSQ_SELL_EVEN_COLS"));
            if (st == LSQ_OK) {
                J->scols.ncb = ncb; J->scols.ccols = ccols; J->scols.ngw = ngw; J->scols.grows = grows;
                LSQ_HIP(hipMalloc(&J->scols.d_part, (size_t)ngw * n * 2 * sizeof(double)));
            } else if (st != LSQ_EDIM) {
                return st;
            } else {
                free_sell(J->scols);
            }
        }
    }
    // window-blocked CSC when the gathered m-vector is larger than ~1 MiB.  Default plan: windows
    // of <= 4096 rows staged in LDS (k_bcsc_lds), sized so that every CU gets a whole number of
    // windows; LSQ_WINDOW_ROWS / LSQ_PLAN_BCSC select the L2-resident variants instead.
    {
        const char *eplan = getenv("LSQ_PLAN_BCSC");
        const char *erows = getenv("LSQ_WINDOW_ROWS");
        bool ldswin = !eplan || !strcmp(eplan, "ldswin");
        int rows_per_win = 131072;
        if (erows) rows_per_win = std::max(64, atoi(erows));
        const bool wanted = !J->scols.active && (erows ? m > rows_per_win : m > 131072);
        if (ldswin && wanted) {
            int per_cu = (m + c->num_cus - 1) / c->num_cus;
            int rounds = (per_cu + LSQ_WIN_ROWS_MAX - 1) / LSQ_WIN_ROWS_MAX;
            rows_per_win = (m + rounds * c->num_cus - 1) / (rounds * c->num_cus);
            rows_per_win = std::min(LSQ_WIN_ROWS_MAX, (rows_per_win + 7) & ~7);
            if (erows) rows_per_win = std::min(LSQ_WIN_ROWS_MAX, std::max(64, atoi(erows)));
        }
        int nwin = (m + rows_per_win - 1) / rows_per_win;
        if (wanted && nwin > 1 && nnz > 0 && (long long)nwin * n < 200000000LL) {
            const int rw = ldswin ? rows_per_win : (m + nwin - 1) / nwin;
            nwin = (m + rw - 1) / rw;
            std::vector<int> bptr((size_t)nwin * n + 1, 0), bidx(nnz), bmap(nnz);
            for (int j = 0; j < n; ++j)
                for (int k = colptr[j]; k < colptr[j + 1]; ++k) bptr[(size_t)(rowval[k] / rw) * n + j + 1]++;
            for (size_t s2 = 0; s2 < (size_t)nwin * n; ++s2) bptr[s2 + 1] += bptr[s2];
            std::vector<int> fill(bptr.begin(), bptr.end() - 1);
            for (int j = 0; j < n; ++j)
                for (int k = colptr[j]; k < colptr[j + 1]; ++k) {
                    int p = fill[(size_t)(rowval[k] / rw) * n + j]++;
                    bidx[p] = rowval[k];
                    bmap[p] = k;
                }
            LSQ_TRY(upload_segs(c, J->bcsc, bptr, bidx, "LSQ_PLAN_BCSC", n));
            J->bcsc.rw = rw;
            J->bcsc.nwin = nwin;
            if (ldswin) {
                // big tiles that never straddle a window; segments longer than a tile disable the plan
                std::vector<int> big;
                build_tiles(bptr, J->bcsc.nseg, big, n, LSQ_WIN_SEGS, LSQ_BIG_NNZ);
                int nb = (int)big.size() - 1;
                bool ok = true;
                for (int t = 0; t < nb; ++t)
                    if (bptr[big[t + 1]] - bptr[big[t]] > LSQ_BIG_NNZ) ok = false;
                if (ok) {
                    std::vector<int> bm((size_t)4 * nb + 4), wt(nwin + 1, 0);
                    for (int t = 0; t < nb; ++t) {
                        bm[4 * t + 0] = big[t];
                        bm[4 * t + 1] = big[t + 1];
                        bm[4 * t + 2] = bptr[big[t]];
                        bm[4 * t + 3] = bptr[big[t + 1]];
                        wt[big[t] / n + 1]++;
                    }
                    for (int w = 0; w < nwin; ++w) wt[w + 1] += wt[w];
                    hipFree(J->bcsc.d_big);
                    LSQ_HIP(hipMalloc(&J->bcsc.d_big, bm.size() * sizeof(int)));
                    LSQ_HIP(hipMemcpy(J->bcsc.d_big, bm.data(), bm.size() * sizeof(int), hipMemcpyHostToDevice));
                    LSQ_HIP(hipMalloc(&J->bcsc.d_wtile, wt.size() * sizeof(int)));
                    LSQ_HIP(hipMemcpy(J->bcsc.d_wtile, wt.data(), wt.size() * sizeof(int), hipMemcpyHostToDevice));
                    J->bcsc.nbig = nb;
                    J->bcsc.plan = LSQ_PLAN_LDSWIN;
                    {  // in-window row offsets (< 4096) in 16 bits
                        std::vector<unsigned short> i16(nnz + 8, 0);
                        for (long long k = 0; k < nnz; ++k) i16[k] = (unsigned short)(bidx[k] % rw);
                        LSQ_HIP(hipMalloc(&J->bcsc.d_idx16, i16.size() * sizeof(unsigned short)));
                        LSQ_HIP(hipMemcpy(J->bcsc.d_idx16, i16.data(), i16.size() * sizeof(unsigned short),
                                          hipMemcpyHostToDevice));
                        if (n <= 65535) {  // column of each entry, for column-scaling g! kernels
                            for (size_t sg = 0; sg < (size_t)nwin * n; ++sg)
                                for (int k = bptr[sg]; k < bptr[sg + 1]; ++k) i16[k] = (unsigned short)(sg % n);
                            LSQ_HIP(hipMalloc(&J->bcsc.d_col16, i16.size() * sizeof(unsigned short)));
                            LSQ_HIP(hipMemcpy(J->bcsc.d_col16, i16.data(), i16.size() * sizeof(unsigned short),
                                              hipMemcpyHostToDevice));
                        }
                    }
                }
            }
            LSQ_HIP(hipMalloc(&J->d_bmap, (nnz + 8) * sizeof(int)));
            LSQ_HIP(hipMemcpy(J->d_bmap, bmap.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
            LSQ_HIP(hipMalloc(&J->d_bpart, (size_t)nwin * n * 2 * sizeof(double)));  // [w][dots | squares]
            J->nwin = nwin;
        }
    }
    J->csr_fresh = true;  // all zeros
    *out = J;
    return LSQ_OK;
}

extern "C" int lsq_dense_create(lsq_ctx *c, int m, int n, lsq_mat **out) {
    if (!c || !out || m < 0 || n < 0) return LSQ_EARG;
    LSQ_HIP(hipSetDevice(c->device));
    lsq_mat *J = new lsq_mat();
    J->ctx = c;
    J->kind = LSQ_MAT_DENSE;
    J->m = m;
    J->n = n;
    J->nnz = (long long)m * n;
    size_t bytes = (size_t)(J->nnz + 8) * sizeof(double);
    LSQ_HIP(hipMalloc(&J->d_dense, bytes));
    LSQ_ZERO(J->d_dense, 0, bytes);
    LSQ_HIP(hipMalloc(&J->d_colsum, (n > 0 ? n : 1) * sizeof(double)));
    *out = J;
    return LSQ_OK;
}

extern "C" int lsq_op_create(lsq_ctx *c, int m, int n, lsq_op_mul_callback mul, lsq_op_colsum_callback colsumabs2,
                             void *user, lsq_mat **out) {
    if (!c || !out || m < 0 || n < 0 || !mul || !colsumabs2) {
        lsq_set_error("lsq_op_create: bad arguments");
        return LSQ_EARG;
    }
    LSQ_HIP(hipSetDevice(c->device));
    lsq_mat *J = new lsq_mat();
    J->ctx = c;
    J->kind = LSQ_MAT_OP;
    J->m = m;
    J->n = n;
    J->nnz = (long long)m * n;   // (logical size; keeps the small-problem reference-order path off for big operators)
    J->op_mul = mul;
    J->op_colsum = colsumabs2;
    J->op_user = user;
    LSQ_HIP(hipMalloc(&J->d_optmp, (size_t)(std::max(std::max(m, n), 1) + 8) * sizeof(double)));
    LSQ_HIP(hipMalloc(&J->d_colsum, (n > 0 ? n : 1) * sizeof(double)));
    *out = J;
    return LSQ_OK;
}

extern "C" int lsq_mat_destroy(lsq_mat *J) {
    if (!J) return LSQ_OK;
    hipStreamSynchronize(J->ctx->stream);
    if (J->ctx->workspace && *(lsq_mat **)J->ctx->workspace == J) {  // LsqWorkspace::J is its first member
        lsq_workspace_free(J->ctx->workspace);
        J->ctx->workspace = nullptr;
    }
    hipFree(J->d_dense);
    free_segs(J->csc);
    free_segs(J->csr);
    hipFree(J->d_map);
    free_segs(J->bcsc);
    free_sell(J->srows);
    free_sell(J->scols);
    hipFree(J->d_bmap);
    hipFree(J->d_bpart);
    hipFree(J->d_dpart);
    hipFree(J->d_optmp);
    hipFree(J->d_colsum);
    hipFree(J->d_cs_base);
    hipFree(J->d_colsum_base);
    delete J;
    return LSQ_OK;
}

extern "C" int lsq_mat_size(const lsq_mat *J, int *m, int *n, long long *nnz) {
    if (m) *m = J->m;
    if (n) *n = J->n;
    if (nnz) *nnz = J->nnz;
    return LSQ_OK;
}

extern "C" double *lsq_mat_values(lsq_mat *J) {
    J->version++;
    J->base_version++;
    if (J->kind == LSQ_MAT_OP) return nullptr;   // no entries to expose
    if (J->d_cs_base) return J->d_cs_base;       // column-scaled, multiplied out on refresh: the stored values V live here
    if (J->kind == LSQ_MAT_DENSE) return J->d_dense;
    if (lsq_ensure_csc(J) != LSQ_OK) return nullptr;
    J->csr_fresh = false;
    return J->csc.d_val;
}

__global__ void __launch_bounds__(LSQ_NT)
k_permute(long long nnz, const int *__restrict__ map, const double *__restrict__ src, double *__restrict__ dst) {
    for (long long k = blockIdx.x * (long long)LSQ_NT + threadIdx.x; k < nnz;
         k += (long long)gridDim.x * LSQ_NT)
        dst[k] = src[map[k]];
}

// dst[k] = src[map[k]], 0 where map[k] < 0 (padding of the sliced layouts)
__global__ void __launch_bounds__(LSQ_NT)
k_permute_pad(long long nstore, const int *__restrict__ map, const double *__restrict__ src, double *__restrict__ dst) {
    for (long long k = blockIdx.x * (long long)LSQ_NT + threadIdx.x; k < nstore;
         k += (long long)gridDim.x * LSQ_NT) {
        const int p = map[k];
        dst[k] = p >= 0 ? src[p] : 0.0;
    }
}

static int permute_launch(lsq_mat *J, long long count, const int *map, const double *src, double *dst, bool pad) {
    if (count <= 0) return LSQ_OK;
    int grid = (int)std::min<long long>((count + LSQ_NT - 1) / LSQ_NT, (long long)J->ctx->num_cus * 16);
    if (pad) LSQ_LAUNCH(k_permute_pad, dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream, count, map, src, dst);
    else LSQ_LAUNCH(k_permute, dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream, count, map, src, dst);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

long long lsq_mirror_rows_len(const lsq_mat *J) { return J->srows.active ? J->srows.nstore : J->nnz; }
long long lsq_mirror_cols_len(const lsq_mat *J) {
    return J->scols.active ? J->scols.nstore : (J->nwin > 1 ? J->nnz : 0);
}
int lsq_mirror_rows(lsq_mat *J, const double *d_csc_vals, double *d_out) {
    if (J->srows.active) return permute_launch(J, J->srows.nstore, J->srows.d_map, d_csc_vals, d_out, true);
    return permute_launch(J, J->nnz, J->d_map, d_csc_vals, d_out, false);
}
int lsq_mirror_cols(lsq_mat *J, const double *d_csc_vals, double *d_out) {
    if (J->scols.active) return permute_launch(J, J->scols.nstore, J->scols.d_map, d_csc_vals, d_out, true);
    if (J->nwin > 1) return permute_launch(J, J->nnz, J->d_bmap, d_csc_vals, d_out, false);
    return LSQ_OK;
}

__global__ void __launch_bounds__(LSQ_NT)
k_unpermute(long long nnz, const int *__restrict__ map, const double *__restrict__ src, double *__restrict__ dst) {
    for (long long k = blockIdx.x * (long long)LSQ_NT + threadIdx.x; k < nnz;
         k += (long long)gridDim.x * LSQ_NT)
        dst[map[k]] = src[k];
}

__global__ void __launch_bounds__(LSQ_NT)
k_unpermute_pad(long long nstore, const int *__restrict__ map, const double *__restrict__ src, double *__restrict__ dst) {
    for (long long k = blockIdx.x * (long long)LSQ_NT + threadIdx.x; k < nstore;
         k += (long long)gridDim.x * LSQ_NT) {
        const int p = map[k];
        if (p >= 0) dst[p] = src[k];
    }
}

// A device g! may write the mirrors the products read and leave the CSC-ordered copy stale
// (csc_fresh = false); it is rebuilt from the row mirror the first time someone asks for it.
int lsq_ensure_csc(lsq_mat *J) {
    if (J->kind != LSQ_MAT_CSC || J->csc_fresh) return LSQ_OK;
    if (J->nnz > 0) {
        const long long count = J->srows.active ? J->srows.nstore : J->nnz;
        int grid = (int)std::min<long long>((count + LSQ_NT - 1) / LSQ_NT, (long long)J->ctx->num_cus * 16);
        if (J->srows.active)
            LSQ_LAUNCH(k_unpermute_pad, dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream, count, J->srows.d_map,
                               J->srows.d_val, J->csc.d_val);
        else
            LSQ_LAUNCH(k_unpermute, dim3(grid), dim3(LSQ_NT), 0, J->ctx->stream, count, J->d_map,
                               J->csr.d_val, J->csc.d_val);
        LSQ_HIP(hipGetLastError());
    }
    J->csc_fresh = true;
    return LSQ_OK;
}

// refreshes every mirror of the user-visible CSC values
int lsq_ensure_csr(lsq_mat *J) {
    if (J->kind != LSQ_MAT_CSC || J->csr_fresh) return LSQ_OK;
    LSQ_TRY(lsq_mirror_rows(J, J->csc.d_val, J->srows.active ? J->srows.d_val : J->csr.d_val));
    LSQ_TRY(lsq_mirror_cols(J, J->csc.d_val, J->scols.active ? J->scols.d_val : J->bcsc.d_val));
    J->csr_fresh = true;
    return LSQ_OK;
}

// ---- column-scaled Jacobians J = V diag(s) (include/lsqhip.h: lsq_mat_set_colscale) --------------------------------
// out[k] = base[k] * s[column of k]: CSC order (colptr) or dense column-major (m_dense rows per column)
__global__ void __launch_bounds__(LSQ_NT)
k_colscale_apply(int n, const int *__restrict__ colptr, int m_dense, const double *__restrict__ base,
                 const double *__restrict__ s, double *__restrict__ out) {
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const double f = s[j];
        const long long k0 = colptr ? colptr[j] : (long long)j * m_dense;
        const long long k1 = colptr ? colptr[j + 1] : (long long)(j + 1) * m_dense;
        for (long long k = k0 + threadIdx.x; k < k1; k += LSQ_NT) out[k] = base[k] * f;
    }
}
static bool colscale_can_fuse(const lsq_mat *J);
bool lsq_colscale_fusable(const lsq_mat *J) { return colscale_can_fuse(J); }
static bool colscale_can_fuse(const lsq_mat *J) {
    return J->kind == LSQ_MAT_CSC && J->srows.active && J->scols.active && !lsq_small_mat(J) && !getenv("LSQ_NO_COLSCALE");
}
// materialising mode: values = V .* s (the mirrors follow through lsq_ensure_csr)
static int colscale_multiply_out(lsq_mat *J) {
    if (!J->d_cs_base || J->nnz == 0) return LSQ_OK;
    lsq_ctx *c = J->ctx;
    const int grid = std::max(1, std::min(J->n, c->num_cus * 16));
    if (J->kind == LSQ_MAT_DENSE)
        LSQ_LAUNCH(k_colscale_apply, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->n, (const int *)nullptr, J->m,
                           J->d_cs_base, J->d_cs_user, J->d_dense);
    else
        LSQ_LAUNCH(k_colscale_apply, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->n, J->csc.d_ptr, 0, J->d_cs_base,
                           J->d_cs_user, J->csc.d_val);
    LSQ_HIP(hipGetLastError());
    if (J->kind == LSQ_MAT_CSC) {
        J->csr_fresh = false;
        J->csc_fresh = true;
    }
    return LSQ_OK;
}

extern "C" int lsq_mat_colscale_changed(lsq_mat *J) {
    if (!J || !J->d_cs_user) {
        lsq_set_error("lsq_mat_colscale_changed: the matrix has no column scale (lsq_mat_set_colscale)");
        return LSQ_EARG;
    }
    J->version++;
    if (J->d_colscale) return LSQ_OK;          // fused: the products read s themselves
    LSQ_TRY(colscale_multiply_out(J));
    return lsq_ensure_csr(J);
}

extern "C" int lsq_mat_set_colscale(lsq_mat *J, const double *d_s) {
    if (!J || J->kind == LSQ_MAT_OP) {
        lsq_set_error("lsq_mat_set_colscale: needs a matrix with stored values");
        return LSQ_EARG;
    }
    lsq_ctx *c = J->ctx;
    double *vals = J->kind == LSQ_MAT_DENSE ? J->d_dense : J->csc.d_val;
    if (!d_s) {   // back to an ordinary matrix holding V
        if (J->d_cs_base) {
            if (J->nnz) LSQ_TRY(lsq_d2d(c, vals, J->d_cs_base, (size_t)J->nnz * sizeof(double)));
            LSQ_HIP(hipStreamSynchronize(c->stream));
            hipFree(J->d_cs_base);
            J->d_cs_base = nullptr;
        }
        J->d_colscale = J->d_cs_user = nullptr;
        return lsq_mat_refresh(J);
    }
    if (!J->d_cs_user) {   // from now on the values held so far are V
        LSQ_TRY(lsq_ensure_csc(J));
        if (colscale_can_fuse(J)) {
            if (!J->d_colsum_base) LSQ_HIP(hipMalloc(&J->d_colsum_base, (size_t)std::max(J->n, 1) * sizeof(double)));
            J->colsum_base_version = ~0ull;
        } else {
            LSQ_HIP(hipMalloc(&J->d_cs_base, (size_t)(J->nnz + 8) * sizeof(double)));
            if (J->nnz) LSQ_TRY(lsq_d2d(c, J->d_cs_base, vals, (size_t)J->nnz * sizeof(double)));
        }
    }
    J->d_cs_user = d_s;
    J->d_colscale = J->d_cs_base ? nullptr : d_s;
    return lsq_mat_colscale_changed(J);
}

extern "C" int lsq_mat_refresh(lsq_mat *J) {
    LSQ_RANGE("lsq_mat_refresh");
    J->version++;
    J->base_version++;
    if (J->d_cs_base) LSQ_TRY(colscale_multiply_out(J));   // (the caller wrote V: multiply out again)
    if (J->kind == LSQ_MAT_CSC) {
        J->csr_fresh = false;
        J->csc_fresh = true;  // the CSC copy is the authority here
    }
    return lsq_ensure_csr(J);
}

extern "C" int lsq_mat_set_values(lsq_mat *J, const double *h) {
    LSQ_RANGE("lsq_mat_set_values");
    if (J->kind == LSQ_MAT_OP) {
        lsq_set_error("a matrix-free operator has no stored values");
        return LSQ_EARG;
    }
    double *dst = J->d_cs_base ? J->d_cs_base : J->kind == LSQ_MAT_DENSE ? J->d_dense : J->csc.d_val;
    if (J->nnz)
        LSQ_HIP(hipMemcpyAsync(dst, h, J->nnz * sizeof(double), hipMemcpyHostToDevice, J->ctx->stream));
    LSQ_HIP(hipStreamSynchronize(J->ctx->stream));  // h is borrowed for the call only
    return lsq_mat_refresh(J);
}

extern "C" int lsq_mat_set_values_async(lsq_mat *J, const double *h) {
    if (J->kind == LSQ_MAT_OP) {
        lsq_set_error("a matrix-free operator has no stored values");
        return LSQ_EARG;
    }
    lsq_ctx *c = J->ctx;
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, h) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned) return lsq_mat_set_values(J, h);     // pageable memory cannot be copied asynchronously
    if (!c->copy_stream) {
        LSQ_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        LSQ_HIP(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    }
    double *dst = J->d_cs_base ? J->d_cs_base : J->kind == LSQ_MAT_DENSE ? J->d_dense : J->csc.d_val;
    // the copy may overwrite the staging copy only after the compute stream is done reading it (a mirror refresh of the
    // previous upload), and the compute stream may touch J again only after the copy: two device-side waits, no host wait
    LSQ_HIP(hipEventRecord(c->copy_done, c->stream));
    LSQ_HIP(hipStreamWaitEvent(c->copy_stream, c->copy_done, 0));
    if (J->nnz) LSQ_HIP(hipMemcpyAsync(dst, h, J->nnz * sizeof(double), hipMemcpyHostToDevice, c->copy_stream));
    LSQ_HIP(hipEventRecord(c->copy_done, c->copy_stream));
    LSQ_HIP(hipStreamWaitEvent(c->stream, c->copy_done, 0));
    J->upload_pending = true;
    return lsq_mat_refresh(J);     // (device work only: the mirrors are rebuilt behind the copy)
}

extern "C" int lsq_mat_upload_wait(lsq_mat *J) {
    if (!J->upload_pending) return LSQ_OK;
    LSQ_HIP(hipEventSynchronize(J->ctx->copy_done));
    J->upload_pending = false;
    return LSQ_OK;
}

extern "C" int lsq_mat_get_values(const lsq_mat *J, double *h) {
    if (J->kind == LSQ_MAT_OP) {
        lsq_set_error("a matrix-free operator has no stored values");
        return LSQ_EARG;
    }
    LSQ_TRY(lsq_ensure_csc(const_cast<lsq_mat *>(J)));
    // (a column-scaled matrix hands back its stored values V, not V diag(s))
    const double *src = J->d_cs_base ? J->d_cs_base : J->kind == LSQ_MAT_DENSE ? J->d_dense : J->csc.d_val;
    if (J->nnz)
        LSQ_HIP(hipMemcpyAsync(h, src, J->nnz * sizeof(double), hipMemcpyDeviceToHost, J->ctx->stream));
    LSQ_HIP(hipStreamSynchronize(J->ctx->stream));
    return LSQ_OK;
}

// ---------------------------------------------------------------------------------------------
// generic products
// ---------------------------------------------------------------------------------------------
struct EpiAxpby {  // y[s] = alpha*dot + beta*y[s]   (beta == 0 overwrites: _rmul_or_fill! [stdlib])
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double alpha, beta;
    double *y;
    double *partials;
    unsigned *counter;
    __device__ void seg(int s, double dot, double &) const {
        y[s] = (beta == 0.0) ? alpha * dot : alpha * dot + beta * y[s];
    }
    using has_pre = void;
    __device__ double pre(int s) const { return beta == 0.0 ? 0.0 : y[s]; }
    __device__ void seg_pre(int s, double dot, double yo, double &) const {
        y[s] = (beta == 0.0) ? alpha * dot : alpha * dot + beta * yo;
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

int lsq_sparse_mul(lsq_mat *J, int trans, double alpha, const double *x, double beta, double *y) {
    EpiAxpby e{nullptr, 0, alpha, beta, y, nullptr, nullptr};
    return launch_product(J, trans, x, e);
}

struct EpiStore {  // out[s] = dot
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *out;
    double *partials;
    unsigned *counter;
    __device__ void seg(int s, double dot, double &) const { out[s] = dot; }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

int lsq_sparse_colsumabs2(lsq_mat *J, double *out) {
    LSQ_TRY(lsq_ensure_csc(J));
    EpiStore e{nullptr, 0, out, nullptr, nullptr};
    return launch_segs<true>(J->ctx, J->csc, nullptr, e);
}

// g = J'f and colsumabs2(J) from ONE pass over the window-blocked values (LM/Dogleg need both
// right after g!: levenberg_marquardt.jl:82 + :102, dogleg.jl:85 + :99).
struct EpiGradSq {
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    int n;
    double *g, *cs;
    double *partials;
    unsigned *counter;
    __device__ void seg(int j, double dot, double &) const {
        if (j < n) g[j] = dot;
        else cs[j - n] = dot;
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};

bool lsq_can_fuse_grad_colsum(const lsq_mat *J) {
    // 160 KiB of dynamic LDS: yl + products + squares
    if (J->kind == LSQ_MAT_CSC && J->scols.active && !lsq_small_mat(J)) return true;
    if (J->kind != LSQ_MAT_CSC || J->nwin <= 1 || J->bcsc.plan != LSQ_PLAN_LDSWIN || lsq_small_mat(J)) return false;
    const size_t lds = (size_t)(LSQ_WIN_ROWS_MAX + 2 * LSQ_BIG_WINDOW) * sizeof(double);
    return lsq_set_lds(J->ctx, (const void *)k_bcsc_lds<true, true>, lds) == LSQ_OK &&
           lsq_set_lds(J->ctx, (const void *)k_bcsc_lds<false, true>, lds) == LSQ_OK;
}

// column-scaled, fused: colsumabs2(J) = s.^2 .* colsumabs2(V), the latter formed once per V (reference order: CSC segments)
struct EpiGradCs {   // g[j] = (scaled dot), cs[j] = s[j]^2 * base[j]
    static constexpr bool REDUCE = false;
    const int *done;
    int extra_blocks;
    double *g, *cs;
    const double *s, *base;
    double *partials;
    unsigned *counter;
    __device__ void seg(int j, double dot, double &) const {
        g[j] = dot;
        const double f = s[j];
        cs[j] = (f * f) * base[j];
    }
    __device__ void extra(int, double &) const {}
    __device__ void finalize(double) const {}
};
__global__ void __launch_bounds__(LSQ_NT)
k_colsum_scaled(int n, const double *__restrict__ s, const double *__restrict__ base, double *__restrict__ cs) {
    for (int j = blockIdx.x * LSQ_NT + threadIdx.x; j < n; j += gridDim.x * LSQ_NT) cs[j] = (s[j] * s[j]) * base[j];
}
static int colsum_base_ready(lsq_mat *J) {
    if (J->colsum_base_version == J->base_version) return LSQ_OK;
    LSQ_TRY(lsq_sparse_colsumabs2(J, J->d_colsum_base));     // over csc.d_val = V
    J->colsum_base_version = J->base_version;
    return LSQ_OK;
}

int lsq_sparse_grad_colsum(lsq_mat *J, const double *f, double *g) {
    lsq_ctx *c = J->ctx;
    LSQ_TRY(lsq_ensure_csr(J));
    if (J->d_colscale) {
        LSQ_TRY(colsum_base_ready(J));
        LSQ_TRY(launch_sell_cols<false>(J, f, nullptr));
        EpiGradCs e{nullptr, 0, g, J->d_colsum, J->d_colscale, J->d_colsum_base, nullptr, nullptr};
        int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
        int grid = std::min(nb, c->num_cus * 8);
        LSQ_LAUNCH((k_combine<EpiGradCs>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->scols.d_part, J->n, J->scols.ngw,
                           e, nb, J->d_colscale);
        LSQ_HIP(hipGetLastError());
        J->colsum_version = J->version;
        return LSQ_OK;
    }
    if (J->scols.active) {
        LSQ_TRY(launch_sell_cols<true>(J, f, nullptr));
        EpiGradSq e{nullptr, 0, J->n, g, J->d_colsum, nullptr, nullptr};
        int nb = lsq_div_up(2 * J->n, LSQ_CMB_COLS);
        int grid = std::min(nb, c->num_cus * 8);
        LSQ_LAUNCH((k_combine<EpiGradSq>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->scols.d_part, 2 * J->n,
                           J->scols.ngw, e, nb);
        LSQ_HIP(hipGetLastError());
        J->colsum_version = J->version;
        return LSQ_OK;
    }
    const size_t lds = (size_t)(LSQ_WIN_ROWS_MAX + 2 * LSQ_BIG_WINDOW) * sizeof(double);
    auto kern = J->bcsc.d_idx16 ? k_bcsc_lds<true, true> : k_bcsc_lds<false, true>;
    int g2 = std::max(1, std::min(J->bcsc.nwin, c->num_cus));
    LSQ_LAUNCH(kern, dim3(g2), dim3(LSQ_BIG_NT), lds, c->stream, segs_dev(J->bcsc), (const int4 *)J->bcsc.d_big,
                       J->bcsc.d_wtile, J->bcsc.nwin, J->bcsc.rw, J->m, J->n, f, J->d_bpart, (const int *)nullptr);
    EpiGradSq e{nullptr, 0, J->n, g, J->d_colsum, nullptr, nullptr};
    int nb = lsq_div_up(2 * J->n, LSQ_CMB_COLS);
    int grid = std::min(nb, c->num_cus * 8);
    LSQ_LAUNCH((k_combine<EpiGradSq>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_bpart, 2 * J->n, J->nwin,
                       e, nb);
    LSQ_HIP(hipGetLastError());
    J->colsum_version = J->version;
    return LSQ_OK;
}

// The same pass for the NEXT Jacobian of a column-scaled handle, queued before the host knows whether there will be one: the
// handle's stored values with the factor vector `s_new` (what g! at the trial point would hand to lsq_mat_set_colscale) and
// every kernel skipping itself when *gate != 0.  Nothing of the handle's bookkeeping changes here; when the step is accepted
// and g! has installed s_new, lsq_sparse_grad_colsum_adopt marks the cache (J->d_colsum, written by this pass) as current.
// Returns LSQ_EARG when the pass does not apply (the caller then runs the ordinary one later).
int lsq_sparse_grad_colsum_spec(lsq_mat *J, const double *f, double *g, const double *s_new, const int *gate) {
    lsq_ctx *c = J->ctx;
    if (!(J->kind == LSQ_MAT_CSC && J->d_colscale && J->scols.active && J->srows.active && s_new && gate) ||
        J->colsum_base_version != J->base_version)
        return LSQ_EARG;
    LSQ_TRY(launch_sell_cols<false>(J, f, gate));
    EpiGradCs e{gate, 0, g, J->d_colsum, s_new, J->d_colsum_base, nullptr, nullptr};
    int nb = lsq_div_up(J->n, LSQ_CMB_COLS);
    int grid = std::min(nb, c->num_cus * 8);
    LSQ_LAUNCH((k_combine<EpiGradCs>), dim3(grid), dim3(LSQ_NT), 0, c->stream, J->scols.d_part, J->n, J->scols.ngw, e, nb, s_new);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
void lsq_sparse_grad_colsum_adopt(lsq_mat *J) { J->colsum_version = J->version; }
void lsq_sparse_colsum_forget(lsq_mat *J) { J->colsum_version = ~0ull; }

const double *lsq_cached_colsum(lsq_mat *J) {
    if (J->kind == LSQ_MAT_OP) {   // the operator owns its state: ask every time a new version is announced
        if (J->colsum_version != J->version) {
            if (hipStreamSynchronize(J->ctx->stream) != hipSuccess) return nullptr;
            if (J->op_colsum(J->d_colsum, J->op_user) != 0) {
                lsq_set_error("operator colsumabs2 callback reported failure");
                return nullptr;
            }
            J->colsum_version = J->version;
        }
        return J->d_colsum;
    }
    if (J->d_colscale && J->colsum_version != J->version) {
        if (colsum_base_ready(J) != LSQ_OK) return nullptr;
        LSQ_LAUNCH(k_colsum_scaled, dim3(std::max(1, std::min(lsq_div_up(J->n, LSQ_NT), J->ctx->num_cus * 4))), dim3(LSQ_NT),
                           0, J->ctx->stream, J->n, J->d_colscale, J->d_colsum_base, J->d_colsum);
        if (hipGetLastError() != hipSuccess) return nullptr;
        J->colsum_version = J->version;
    }
    if (J->colsum_version != J->version) {
        int st = lsq_small_mat(J) ? lsq_exact_colsumabs2(J, J->d_colsum)
                 : J->kind == LSQ_MAT_DENSE ? lsq_dense_colsumabs2(J, J->d_colsum)
                                            : lsq_sparse_colsumabs2(J, J->d_colsum);
        if (st != LSQ_OK) return nullptr;
        J->colsum_version = J->version;
    }
    return J->d_colsum;
}

extern "C" int lsq_mul(lsq_mat *J, int trans, double alpha, const double *x, double beta, double *y) {
    LSQ_RANGE("lsq_mul");
    if (!J || !x || !y) {
        lsq_set_error("lsq_mul: null argument");
        return LSQ_EARG;
    }
    if (J->kind == LSQ_MAT_OP) return lsq_sparse_mul(J, trans, alpha, x, beta, y);   // (launch_product dispatches on the kind)
    if (lsq_small_mat(J) && x != y) return lsq_exact_mul(J, trans, alpha, x, beta, y);  // reference order
    return J->kind == LSQ_MAT_DENSE ? lsq_dense_mul(J, trans, alpha, x, beta, y)
                                    : lsq_sparse_mul(J, trans, alpha, x, beta, y);
}

extern "C" int lsq_colsumabs2(lsq_mat *J, double *out) {
    LSQ_RANGE("lsq_colsumabs2");
    if (!J || !out) return LSQ_EARG;
    const double *cs = lsq_cached_colsum(J);
    if (!cs) return LSQ_EHIP;
    return lsq_d2d(J->ctx, out, cs, (size_t)J->n * sizeof(double));
}

// ---- rowsumabs2! (utils.jl:153-161): sums of squares along the rows -------------------------------
// sliced rows: the lane that owns a row adds its squares left to right (the reference's order: the
// CSC sweep reaches a row's entries in column order)
__global__ void __launch_bounds__(256)
k_sell_rowsq(SellDev S, int wrows, int m, int ncw, int cwidth, double *__restrict__ out, const double *__restrict__ cscale) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nrb = S.nblocks / ncw;
    for (int w = blockIdx.x; w < nrb; w += gridDim.x) {
        const int base = w * wrows;
        // column windows (ncw > 1, `out` zeroed by the caller): a row's sum continues where the previous window left it
        for (int cw = 0; cw < ncw; ++cw) {
            if (cw) __syncthreads();
            const int b = w * ncw + cw;
            for (int s = b * S.spw + wv; s < (b + 1) * S.spw; s += 4) {
                const int2 sm = S.smeta[s];
                const unsigned inf = S.info[(size_t)s * 64 + lane];
                const unsigned pos = inf & LSQ_SELL_POS_MASK;
                const int len = (int)(inf >> LSQ_SELL_POS_BITS);
                const bool valid = pos != LSQ_SELL_POS_MASK && base + (int)pos < m;
                const double *vp = S.val + (size_t)sm.x + lane * 2;
                const unsigned short *ip = S.idx16 + (size_t)sm.x + lane * 2;
                double acc = (ncw > 1 && valid) ? out[base + pos] : 0.0;
                for (int j = 0; j < len; ++j) {
                    // (pairs, then -- odd slice -- the unpaired last entry in its compact group: lsq_sell.h)
                    const long long o = (sm.y & 1) && j == sm.y - 1 ? (long long)(sm.y >> 1) * 128 - lane : (long long)(j / 2) * 128 + (j & 1);
                    double a = vp[o];
                    if (cscale) a *= cscale[cw * cwidth + ip[o]];   // (column-scaled: the entry of J)
                    acc += a * a;
                }
                if (valid) out[base + pos] = acc;
            }
        }
    }
}
// dense column-major: thread per row, columns in order
__global__ void __launch_bounds__(LSQ_NT)
k_dense_rowsq(const double *__restrict__ A, int m, int n, double *__restrict__ out) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < m; i += gridDim.x * LSQ_NT) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) {
            const double a = A[(size_t)j * m + i];
            acc += a * a;
        }
        out[i] = acc;
    }
}

extern "C" int lsq_rowsumabs2(lsq_mat *J, double *out) {
    LSQ_RANGE("lsq_rowsumabs2");
    if (!J || !out) return LSQ_EARG;
    if (J->kind == LSQ_MAT_OP) {
        lsq_set_error("rowsumabs2 of a matrix-free operator: not provided by its callbacks");
        return LSQ_EARG;
    }
    lsq_ctx *c = J->ctx;
    if (J->m <= 0) return LSQ_OK;
    if (J->kind == LSQ_MAT_DENSE) {
        int grid = std::min(lsq_div_up(J->m, LSQ_NT), c->num_cus * 8);
        LSQ_LAUNCH(k_dense_rowsq, dim3(grid), dim3(LSQ_NT), 0, c->stream, J->d_dense, J->m, J->n, out);
    } else {
        LSQ_TRY(lsq_ensure_csr(J));
        if (J->srows.active) {
            const int ncw = J->srows.ncw;
            int grid = std::max(1, std::min(J->srows.nblocks / ncw, c->num_cus * 4));
            if (ncw > 1) LSQ_HIP(hipMemsetAsync(out, 0, (size_t)J->m * sizeof(double), c->stream));   // rows without entries
            LSQ_LAUNCH(k_sell_rowsq, dim3(grid), dim3(256), 0, c->stream, sell_dev(J->srows), J->srows.wrows, J->m, ncw,
                               J->srows.cwidth, out, J->d_colscale);
        } else {
            EpiStore e{nullptr, 0, out, nullptr, nullptr};
            LSQ_TRY(launch_segs<true>(c, J->csr, nullptr, e));
        }
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

extern "C" int lsq_bench_mul(lsq_mat *J, int trans, int reps, const double *x, double *y, double beta,
                             float *ms) {
    hipEvent_t e0, e1;
    LSQ_HIP(hipEventCreate(&e0));
    LSQ_HIP(hipEventCreate(&e1));
    LSQ_TRY(lsq_mul(J, trans, 1.0, x, beta, y));  // warm-up (also refreshes the CSR mirror)
    LSQ_HIP(hipEventRecord(e0, J->ctx->stream));
    for (int i = 0; i < reps; ++i) LSQ_TRY(lsq_mul(J, trans, 1.0, x, beta, y));
    LSQ_HIP(hipEventRecord(e1, J->ctx->stream));
    LSQ_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    LSQ_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (reps > 0 ? reps : 1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return LSQ_OK;
}
