"""Ad-hoc check above the C4 size (4e7 nnz: several row blocks / gather windows per workgroup): three LM+LSMR
iterations on the GPU against the oracle."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import lsq_amd as lsq
from oracle import oracle as O
ctx = lsq.default_context()
m, n, pc = 4_000_000, 10_000, 4000
t0 = time.time()
inputs = lsq.synthetic.sparse_inputs(m, n, pc, 7)
pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=7, ctx=ctx, inputs=inputs)
print("setup %.1f s" % (time.time() - t0), flush=True)
LM, LSMR = lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR
pr.reset()
t0 = time.time()
r = pr.optimize(LM, LSMR, iterations=3, trace=True)
print("gpu 3 iterations %.3f s, ssr %.12e inner %s" % (time.time() - t0, r.ssr, list(r.trace["inner"])), flush=True)
colptr, rowval, nzval = inputs
A = O.Mat(csc=(m, n, colptr, rowval, nzval))
J = O.Mat(csc=(m, n, colptr, rowval, np.zeros_like(nzval)))
f, g, ud, keep = O.tanh_model(A, pr.b)
t0 = time.time()
ro = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=3)
print("oracle 3 iterations %.1f s, ssr %.12e inner %s" % (time.time() - t0, ro.ssr, list(ro.trace["inner"])), flush=True)
print("ssr rel diff %.2e, max|x - x_oracle| %.2e" % (abs(r.ssr - ro.ssr) / ro.ssr, np.max(np.abs(r.minimizer - ro.trace["x"][-1]))))
