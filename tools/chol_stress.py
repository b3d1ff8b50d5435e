#!/usr/bin/env python3
"""Stress of the one-launch blocked Cholesky (k_chol_chain; LSQ_CHOL_TILES_V1=1: k_chol_tiles): many damped solves on one solver; every result must equal the
first one bit for bit, the path must stay 'blocked-one-launch' (a wait that gave up would switch it to 'blocked'), and
the slowest solve is reported (a wait that limps shows up there).   python tools/chol_stress.py [n_solves] [m] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
ctx = lsq.Context(0)
rng = np.random.default_rng(5)
A = rng.standard_normal((m, n)) / np.sqrt(m)
J = lsq.DeviceMatrix(ctx, A)
y = lsq.DeviceVector(ctx, m, rng.standard_normal(m))
damp = lsq.DeviceVector(ctx, n, rng.random(n) + 0.01)
x = lsq.DeviceVector(ctx, n)
sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
sv.ldiv_(x, y, damp)
ref = x.get().copy()
worst, tot, bad, paths = 0.0, 0.0, 0, {}
for k in range(N):
    t0 = time.perf_counter()
    sv.ldiv_(x, y, damp)
    ctx.sync()
    dt = time.perf_counter() - t0
    worst, tot = max(worst, dt), tot + dt
    p = sv.info()["chol_path"]
    paths[p] = paths.get(p, 0) + 1
    if not np.array_equal(x.get(), ref):
        bad += 1
print("%d solves %dx%d: mean %.3f ms, worst %.3f ms, results differing from the first: %d, paths %s" % (N, m, n, tot / N * 1e3, worst * 1e3, bad, paths))
sys.exit(1 if bad or list(paths) != ["blocked-one-launch"] else 0)
