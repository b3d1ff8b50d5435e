#!/usr/bin/env python3
"""Micro-benchmark of the sparse product kernels (HIP events, back-to-back launches).
usage: spmv_bench.py [env-combo ...] where a combo is e.g. LSQ_WINDOW_ROWS=65536,LSQ_PLAN_BCSC=stream"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
L = lsq.lib()
ctx = lsq.Context(0)
m, n, pc = 1_000_000, 10000, 1000
cp, rv, nz = lsq.synthetic.sparse_inputs(m, n, pc, 1)
combos = sys.argv[1:] or [""]
KEYS = ("LSQ_PLAN_CSC", "LSQ_PLAN_CSR", "LSQ_PLAN_BCSC", "LSQ_WINDOW_ROWS")
for combo in combos:
    for k in KEYS: os.environ.pop(k, None)
    for kv in filter(None, combo.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    h = C.c_void_p()
    lsq._lib.check(L.lsq_csc_create(ctx.h, m, n, cp.ctypes.data_as(lsq._lib.c_ip), rv.ctypes.data_as(lsq._lib.c_ip), C.byref(h)))
    lsq._lib.check(L.lsq_mat_set_values(h, nz.ctypes.data_as(lsq._lib.c_dp)))
    x = lsq.DeviceVector(ctx, n, np.ones(n)); y = lsq.DeviceVector(ctx, m, np.ones(m))
    nnz = n * pc
    res = []
    for trans in (0, 1):
        ms = C.c_float(0)
        if trans == 0: lsq._lib.check(L.lsq_bench_mul(h, 0, 30, x.ptr, y.ptr, 1.0, C.byref(ms)))
        else: lsq._lib.check(L.lsq_bench_mul(h, 1, 30, y.ptr, x.ptr, 1.0, C.byref(ms)))
        b = 12 * nnz + (4 * (m + 1) + 8 * n + 16 * m if trans == 0 else 4 * (n + 1) + 8 * m + 16 * n)
        res.append("%s %.1f us %.0f GB/s" % ("Jv" if trans == 0 else "J'u", ms.value * 1e3, b / ms.value / 1e6))
    print("%-50s %s" % (combo or "(default)", " | ".join(res)), flush=True)
    L.lsq_mat_destroy(h)
