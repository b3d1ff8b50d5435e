#!/usr/bin/env python3
"""J*v / J'u on WIDE sparse patterns (n > 12160: x does not fit in LDS): column-windowed sliced rows (k_sell_rows_wide)
against the segment kernels (LSQ_NO_SELL_WIDE=1), back-to-back launches timed with HIP events, results checked against scipy.
usage: wide_bench.py [m,n,per_col ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import lsq_amd as lsq
L = lsq.lib()
ctx = lsq.Context(0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(1_000_000, 10000, 1000), (1_000_000, 25000, 400), (1_000_000, 50000, 200), (1_000_000, 100000, 100), (200000, 50000, 40)]
for m, n, pc in shapes:
    cp, rv, nz = lsq.synthetic.sparse_inputs(m, n, pc, 1)
    S = sp.csc_matrix((nz, rv, cp), shape=(m, n))
    rng = np.random.default_rng(5)
    xv, yv = rng.standard_normal(n), rng.standard_normal(m)
    ref0, ref1 = S @ xv, S.T @ yv
    nnz = len(nz)
    for combo in ("", "LSQ_NO_SELL_WIDE=1"):
        for kv in filter(None, combo.split(",")):
            k, v = kv.split("="); os.environ[k] = v
        h = C.c_void_p()
        lsq._lib.check(L.lsq_csc_create(ctx.h, m, n, cp.ctypes.data_as(lsq._lib.c_ip), rv.ctypes.data_as(lsq._lib.c_ip), C.byref(h)))
        lsq._lib.check(L.lsq_mat_set_values(h, nz.ctypes.data_as(lsq._lib.c_dp)))
        os.environ.pop("LSQ_NO_SELL_WIDE", None)
        x = lsq.DeviceVector(ctx, n, xv); y = lsq.DeviceVector(ctx, m, yv)
        o0 = lsq.DeviceVector(ctx, m); o1 = lsq.DeviceVector(ctx, n)
        lsq._lib.check(L.lsq_mul(h, 0, 1.0, x.ptr, 0.0, o0.ptr))
        lsq._lib.check(L.lsq_mul(h, 1, 1.0, y.ptr, 0.0, o1.ptr))
        e0, e1 = np.max(np.abs(o0.get() - ref0)), np.max(np.abs(o1.get() - ref1))
        res = []
        for trans in (0, 1):
            ms = C.c_float(0)
            if trans == 0: lsq._lib.check(L.lsq_bench_mul(h, 0, 30, x.ptr, o0.ptr, 1.0, C.byref(ms)))
            else: lsq._lib.check(L.lsq_bench_mul(h, 1, 30, y.ptr, o1.ptr, 1.0, C.byref(ms)))
            b = 12 * nnz + (4 * (m + 1) + 8 * n + 16 * m if trans == 0 else 4 * (n + 1) + 8 * m + 16 * n)
            res.append("%s %.1f us %.0f GB/s" % ("Jv" if trans == 0 else "J'u", ms.value * 1e3, b / ms.value / 1e6))
        print("%8dx%-6d nnz %9d %-20s %s | err %.1e %.1e" % (m, n, nnz, combo or "(default)", " | ".join(res), e0, e1), flush=True)
        L.lsq_mat_destroy(h)
