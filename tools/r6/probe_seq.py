#!/usr/bin/env python3
"""dense_bench's sequence of cases in ONE process with every solve timed (round 6: where do the 17 ms averages at 3000x700 come from?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
def bench(m, n, f, reps):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    J = lsq.DeviceMatrix(ctx, A)
    y = lsq.DeviceVector(ctx, m, rng.standard_normal(m)); x = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=bool(f))
    d = lsq.DeviceVector(ctx, n, np.full(n, 0.1)) if f else None
    ts, phases = [], []
    for i in range(reps):
        t0 = time.perf_counter()
        if f:
            d.set(np.full(n, 0.1)); ctx.sync()
        t1 = time.perf_counter()
        sv.ldiv_(x, y, d) if f else sv.ldiv_(x, y)
        t2 = time.perf_counter()
        ctx.sync()
        t3 = time.perf_counter()
        ts.append((t3 - t1) * 1e3)
        phases.append("set %.2f | ldiv %.2f | sync %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    print("%s %dx%d f=%d: %s   giveups %s" % (os.environ.get("TAG", "?"), m, n, f, " ".join("%.2f" % t for t in ts), ctx.fallback_stats()), flush=True)
    if max(ts[1:]) > 3 * np.median(ts):
        for p in phases: print("      " + p)
for arg in sys.argv[1:]:
    m, n, f = arg.split(":")
    bench(int(m), int(n), int(f), 5)
