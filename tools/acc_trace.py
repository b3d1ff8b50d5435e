import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lsq_amd as lsq
ctx = lsq.Context(0)
pr = lsq.synthetic.TanhProblem(1_000_000, 10_000, sparse=True, per_col=1000, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
pr.reset()
r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, x_tol=0, f_tol=0, g_tol=0, iterations=30, trace=True)
t = r.trace
for k in range(30): print(k+1, t["accept"][k], t["inner"][k]//2, "%.6e"%t["ssr"][k], "%.3e"%t["gnorm"][k], "%.3e"%t["delta"][k], "%.3e"%t["rho"][k])
pr.reset()
r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=100)
print("default tolerances:", r.iterations, r.converged, r.x_converged, r.f_converged, r.g_converged, r.ssr, r.seconds)
