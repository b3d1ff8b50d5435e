import os, sys, time, gc, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
L = lsq.lib()
ctx = lsq.Context(0)
def c3(m=16384, n=2048):
    pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
    pr.reset(); pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
    best = 1e9
    for _ in range(2):
        pr.reset(); t0 = time.perf_counter()
        r = pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False); ctx.sync()
        best = min(best, (time.perf_counter() - t0) / r.iterations * 1e3)
    pr.close(); return best
print("fresh: %.2f ms" % c3(), flush=True)
m, n = 16384, 2048
rng = np.random.default_rng(0)
A = rng.standard_normal((m, n)) / np.sqrt(m); yh = rng.standard_normal(m)
J = lsq.DeviceMatrix(ctx, A); y = lsq.DeviceVector(ctx, m, yh); x = lsq.DeviceVector(ctx, n)
sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
for _ in range(3): sv.ldiv_(x, y); ctx.sync()
print("after 3 operator-level QR ldiv! (solver alive): %.2f ms" % c3(), flush=True)
xg = x.get()
print("after x.get(): %.2f ms" % c3(), flush=True)
info = sv.info()
print("after sv.info(): %.2f ms" % c3(), flush=True)
J.free()
print("after J.free(): %.2f ms" % c3(), flush=True)
del sv; gc.collect()
print("after del sv: %.2f ms" % c3(), flush=True)
