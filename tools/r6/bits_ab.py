#!/usr/bin/env python3
"""Same-box comparison of two builds of liblsqhip.so (LSQ_LIB_PATH per process): solutions of a few dense solves dumped to .npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
out = {}
for kind, m, n, f in (("chol", 4096, 512, 1), ("chol", 3000, 700, 1), ("chol", 1000, 130, 0), ("qr", 16384, 2048, 0), ("qr", 4099, 200, 1), ("qr", 20000, 300, 0)):
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    J = lsq.DeviceMatrix(ctx, A)
    y = lsq.DeviceVector(ctx, m, rng.standard_normal(m)); x = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky() if kind == "chol" else lsq.QR(), for_lm=bool(f))
    d = lsq.DeviceVector(ctx, n, rng.random(n) + 0.01) if f else None
    sv.ldiv_(x, y, d) if f else sv.ldiv_(x, y)
    out["%s_%d_%d_%d" % (kind, m, n, f)] = x.get().copy()
np.savez(sys.argv[1], **out)
