#!/usr/bin/env python3
"""the C4 problem (10^6 x 10^4, nnz 10^7) with a HOST-side g! -- the Jacobian values are computed on the
host (numpy) and uploaded after every accepted step (levenberg_marquardt.jl:77-81).  Reports the time per LM outer
iteration and where it goes: host g! arithmetic, upload (page-locked + asynchronous vs pageable + blocking), device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import scipy.sparse as sp
import lsq_amd as lsq
from lsq_amd import _lib

m, n, pc = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1_000_000, 10_000, 1000)))
ctx = lsq.Context(0)
colptr, rowval, A = lsq.synthetic.sparse_inputs(m, n, pc, lsq.synthetic.BASE_SEED)
cols = np.repeat(np.arange(n), np.diff(colptr))
S = sp.csc_matrix((A, rowval, colptr), shape=(m, n))
x_true, b = lsq.synthetic.rhs_for(lambda t: S @ t, m, n, lsq.synthetic.BASE_SEED)
tg = [0.0, 0]

def f_(out, x):
    out[:] = S @ np.tanh(x) - b

def g_(J, x):
    t0 = time.perf_counter()
    s = 1.0 - np.tanh(x) ** 2
    np.multiply(A, s[cols], out=J.data)
    tg[0] += time.perf_counter() - t0
    tg[1] += 1

J = sp.csc_matrix((np.zeros_like(A), rowval, colptr), shape=(m, n))
nls = lsq.LeastSquaresProblem(x=np.zeros(n), y=np.zeros(m), f_=f_, g_=g_, J=J)
nla = lsq.LeastSquaresProblemAllocated(nls, lsq.LevenbergMarquardt(lsq.LSMR()), ctx=ctx)
lsq.optimize_(nla, iterations=2, x_tol=0, f_tol=0, g_tol=0)       # warm-up
nla.x[:] = 0
tg[:] = [0.0, 0]
t0 = time.perf_counter()
r = lsq.optimize_(nla, iterations=8, x_tol=0, f_tol=0, g_tol=0)
dt = time.perf_counter() - t0
print("host g! LM+LSMR, %dx%d nnz %d: %.1f ms per outer iteration (%d iterations, %d g! calls: %.1f ms of numpy each)"
      % (m, n, len(A), dt / r.iterations * 1e3, r.iterations, tg[1], tg[0] / max(tg[1], 1) * 1e3))
# the upload alone: page-locked + asynchronous vs pageable + blocking
Jd = nla._Jd
stage = nla._stage
for name, go in (("pinned, lsq_mat_set_values_async (host returns at once; device-side wait)", lambda: Jd.set_values_async(stage)),
                 ("pageable, lsq_mat_set_values (blocking)", lambda: Jd.set_values(np.array(stage.array)))):
    go(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        go()
    t_host = (time.perf_counter() - t0) / 5
    ctx.sync(); Jd.upload_wait()
    t0 = time.perf_counter()
    for _ in range(5):
        go()
    Jd.upload_wait(); ctx.sync()
    t_all = (time.perf_counter() - t0) / 5
    print("  upload %s: host blocked %.2f ms, end to end %.2f ms (%.1f GB/s)" % (name, t_host * 1e3, t_all * 1e3, len(A) * 8 / t_all / 1e9))
nla.free()
