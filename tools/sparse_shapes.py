#!/usr/bin/env python3
"""LM(LSMR) / Dogleg(LSMR) outer-iteration time on sparse tanh problems of various shapes (per_col stored entries per column)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsq_amd as lsq
ctx = lsq.Context(0)
L = lsq._lib
shapes = [(1000000, 10000, 1000), (100000, 1000, 1000), (1000000, 100, 20000), (200000, 50000, 40), (2000000, 20, 200000),
          (50000, 50000, 10), (1000000, 1000, 5000)]
if len(sys.argv) > 1:      # m,n,per_col ...
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for m, n, pc in shapes:
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
    for oname, opt in (("LM", L.LEVENBERG_MARQUARDT), ("Dogleg", L.DOGLEG)):
        pr.reset()
        pr.optimize(opt, L.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
        best = None
        for _ in range(3):
            pr.reset()
            t0 = time.perf_counter()
            r = pr.optimize(opt, L.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False)
            ctx.sync()
            ms = (time.perf_counter() - t0) / max(r.iterations, 1) * 1e3
            best = ms if best is None else min(best, ms)
        nnz = pr.nnz
        print("%8dx%-6d nnz %9d  %-6s LSMR  %8.3f ms / outer  (%.1f us per stored entry per 1e6; %.1f LSMR inner iterations per outer)  ssr %.4e"
              % (m, n, nnz, oname, best, best * 1e3 / (nnz / 1e6), r.lsmr_iterations / max(r.iterations, 1), r.ssr), flush=True)
    pr.close()
