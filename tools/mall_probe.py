#!/usr/bin/env python3
"""Is the back-to-back J*v number flattered by the 256 MiB Infinity Cache?  Times J*v alone,
J'u alone, and the alternating sequence the LSMR loop really runs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
L = lsq.lib()
ctx = lsq.Context(0)
m, n, pc = 1_000_000, 10000, 1000
cp, rv, nz = lsq.synthetic.sparse_inputs(m, n, pc, 1)
h = C.c_void_p()
lsq._lib.check(L.lsq_csc_create(ctx.h, m, n, cp.ctypes.data_as(lsq._lib.c_ip), rv.ctypes.data_as(lsq._lib.c_ip), C.byref(h)))
lsq._lib.check(L.lsq_mat_set_values(h, nz.ctypes.data_as(lsq._lib.c_dp)))
x = lsq.DeviceVector(ctx, n, np.ones(n)); y = lsq.DeviceVector(ctx, m, np.ones(m))
def timed(fn, reps=40):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e6
jv = lambda: L.lsq_mul(h, 0, 1.0, x.ptr, 1.0, y.ptr)
jtu = lambda: L.lsq_mul(h, 1, 1.0, y.ptr, 0.0, x.ptr)
def both(): jv(); jtu()
a, b, c = timed(jv), timed(jtu), timed(both)
print("J*v alone %.1f us | J'u alone %.1f us | alternating pair %.1f us (sum of parts %.1f)" % (a, b, c, a + b))
