#!/usr/bin/env python3
"""Time per OUTER iteration of the dense tanh problem at the C3 size (Dogleg + QR), as bench.py's dense_secondary measures it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsq_amd as lsq
ctx = lsq.Context(0)
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 2048)
pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
pr.reset()
pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
for _ in range(3):
    pr.reset()
    t0 = time.perf_counter()
    r = pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False)
    ctx.sync()
    print("%.3f ms per outer iteration (%d iterations, ssr %.6g, f/g/mul calls %s)" % ((time.perf_counter() - t0) / r.iterations * 1e3, r.iterations, r.ssr, (r.f_calls, r.g_calls, r.mul_calls)), flush=True)
