"""Dogleg(LSMR) on the C4 problem: outer / inner iteration rates (not the headline; checks that the
combination has no slow path)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.default_context()
m, n, pc = 1_000_000, 10_000, 1000
pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
for opt, name in ((lsq._lib.DOGLEG, "dogleg"), (lsq._lib.LEVENBERG_MARQUARDT, "lm")):
    pr.reset(); pr.optimize(opt, lsq._lib.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=3, fetch_x=False)
    pr.reset(); ctx.sync()
    t0 = time.perf_counter()
    r = pr.optimize(opt, lsq._lib.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=8, fetch_x=False)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("%s: %d outer, %d inner, %.3f ms per outer, %.1f us per inner, ssr %.6e" % (name, r.iterations, r.lsmr_iterations, dt / r.iterations * 1e3, dt / max(r.lsmr_iterations, 1) * 1e6, r.ssr))
