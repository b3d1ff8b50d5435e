#!/usr/bin/env python3
"""One dense optimizer x solver combination of the tanh model, for kernel-level profiling:
   python tools/combo_one.py 16384x2048 dogleg qr [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsq_amd as lsq
ctx = lsq.Context(0)
L = lsq._lib
m, n = (int(v) for v in sys.argv[1].split("x"))
opt = {"lm": L.LEVENBERG_MARQUARDT, "dogleg": L.DOGLEG}[sys.argv[2]]
sol = {"qr": L.QR, "cholesky": L.CHOLESKY, "lsmr": L.LSMR}[sys.argv[3]]
its = int(sys.argv[4]) if len(sys.argv) > 4 else 6
pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
pr.reset()
pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
for _ in range(3):
    pr.reset()
    t0 = time.perf_counter()
    r = pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=its, fetch_x=False)
    ctx.sync()
    print("%dx%d %s %s: %.3f ms / outer iteration (%d iterations, ssr %.6e)" % (m, n, sys.argv[2], sys.argv[3],
          (time.perf_counter() - t0) / max(r.iterations, 1) * 1e3, r.iterations, r.ssr), flush=True)
pr.close()
