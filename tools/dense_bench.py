#!/usr/bin/env python3
"""Timing of the dense solvers (C2: 4096x512 LM+Cholesky; C3: 16384x2048 Dogleg+QR)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
L = lsq.lib()
ctx = lsq.Context(0)
def bench(m, n, solver, for_lm, reps=5):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    J = lsq.DeviceMatrix(ctx, A)
    y = lsq.DeviceVector(ctx, m, rng.standard_normal(m)); x = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, solver, for_lm=for_lm)
    d = lsq.DeviceVector(ctx, n, np.full(n, 0.1)) if for_lm else None
    def go():
        if for_lm:
            sv.ldiv_(x, y, d)
        else:
            sv.ldiv_(x, y)
        ctx.sync()
    go()
    dt = 0.0
    for _ in range(reps):
        if for_lm:                      # (a damped solver may clobber damp: fresh operand, outside the timed part)
            d.set(np.full(n, 0.1)); ctx.sync()
        t0 = time.perf_counter()
        go()
        dt += (time.perf_counter() - t0) / reps
    xr = x.get()
    if for_lm: ref = np.linalg.solve(A.T @ A + 0.1 * np.eye(n), A.T @ y.get())
    else: ref = np.linalg.lstsq(A, y.get(), rcond=None)[0]
    inf = sv.info()
    print("%-9s %6dx%-5d for_lm=%d  %.3f ms  relerr %.2e  %s %s" % (type(solver).__name__, m, n, for_lm, dt * 1e3, np.linalg.norm(xr - ref) / np.linalg.norm(ref), inf.get("qr_path"), inf.get("qr_panel")), flush=True)
for arg in sys.argv[1:] or ["chol:4096:512:1", "qr:4096:512:0", "qr:4096:512:1"]:
    k, m, n, f = arg.split(":")
    bench(int(m), int(n), lsq.Cholesky() if k == "chol" else lsq.QR(), int(f) == 1, reps=3)
