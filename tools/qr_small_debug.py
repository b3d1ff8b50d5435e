#!/usr/bin/env python3
"""Stress of small two-stage QR solves: per-solve wall times, looking for outliers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
os.environ["LSQ_QR_TWO_STAGE"] = "1"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for m, n in [(300, 300), (600, 256), (2000, 400), (6000, 600)]:
    for fresh in (False, True):
        rng = np.random.default_rng(0)
        A = rng.standard_normal((m, n)) / np.sqrt(m)
        J = lsq.DeviceMatrix(ctx, A)
        y = lsq.DeviceVector(ctx, m, rng.standard_normal(m)); x = lsq.DeviceVector(ctx, n)
        sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
        sv.ldiv_(x, y); ctx.sync()
        ts = []
        for k in range(reps):
            if fresh and k % 10 == 0:
                sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
                sv.ldiv_(x, y); ctx.sync()
            t0 = time.perf_counter(); sv.ldiv_(x, y); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts)
        print(m, n, "fresh" if fresh else "reuse", "median %.3f" % np.median(ts), "max %.3f" % ts.max(),
              "outliers(>3x median): %d" % (ts > 3 * np.median(ts)).sum(), "top:", np.round(np.sort(ts)[-4:], 2), flush=True)
