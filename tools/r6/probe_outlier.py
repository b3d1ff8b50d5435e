#!/usr/bin/env python3
"""Which solve of a repeated dense QR ldiv! is slow, and did a bounded wait give up?  (round 6: 17 ms averages at 3000x700 damped)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
m, n, f, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(0)
A = rng.standard_normal((m, n)) / np.sqrt(m)
J = lsq.DeviceMatrix(ctx, A)
y = lsq.DeviceVector(ctx, m, rng.standard_normal(m)); x = lsq.DeviceVector(ctx, n)
sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=bool(f))
d = lsq.DeviceVector(ctx, n, np.full(n, 0.1)) if f else None
ts, ref = [], None
for i in range(reps):
    if f:
        d.set(np.full(n, 0.1)); ctx.sync()
    before = ctx.fallback_stats()
    t0 = time.perf_counter()
    sv.ldiv_(x, y, d) if f else sv.ldiv_(x, y)
    ctx.sync()
    dt = (time.perf_counter() - t0) * 1e3
    ts.append(dt)
    xr = x.get()
    if ref is None:
        ref = xr
    after = ctx.fallback_stats()
    if dt > 3 * np.median(ts) or before != after or not np.array_equal(xr, ref):
        print("  solve %d: %.2f ms  giveups %s -> %s  bits_equal=%s  info=%s" % (i, dt, before, after, np.array_equal(xr, ref), sv.info()), flush=True)
print("%s %dx%d f=%d: median %.3f ms, max %.2f ms, slow(>3x) %d of %d" % (os.environ.get("TAG", "?"), m, n, f, np.median(ts), max(ts), sum(t > 3 * np.median(ts) for t in ts), reps), flush=True)
