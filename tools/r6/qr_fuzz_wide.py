#!/usr/bin/env python3
"""Random multi-panel shapes (n up to 1500: many 64-column panels, row groups that do not divide the CUs, panels on either side of
the 256-slab look-ahead rule) through the dense QR, undamped and damped, against numpy; also the round-5 grid of the trailing
update (LSQ_QR_UPDATE_FLAT=0 in a second process) must give the same bits.   python tools/r6/qr_fuzz_wide.py <seed> <count> <out.npz>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
count = int(sys.argv[2])
bad, out = 0, {}
t0 = time.time()
for it in range(count):
    n = int(rng.integers(64, 1500))
    m = int(rng.integers(n, max(n + 1, min(40000, int(2.4e7 // n)))))
    if rng.random() < 0.3:
        m = int(rng.choice([16384, 16385, 16448, 16500, 17000, 18432, 20000])) if 16384 * n <= 2.4e7 else m
    f = int(rng.random() < 0.5)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    dmp = rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    x = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=bool(f))
    if f:
        sv.ldiv_(x, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, dmp))
        ref = np.linalg.solve(A.T @ A + np.diag(dmp), A.T @ y)
        tol = 1e-9
    else:
        sv.ldiv_(x, lsq.DeviceVector(ctx, m, y))
        ref = np.linalg.lstsq(A, y, rcond=None)[0]
        tol = 1e-9
    got = x.get()
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    out["%d_%d_%d" % (m, n, f)] = got
    if not (err <= tol):
        bad += 1
        print("FAIL %d x %d damped=%d: rel err %.3g  %s" % (m, n, f, err, sv.info()))
    J.free()
np.savez(sys.argv[3], **out)
print("shapes %d, failures %d, %.1f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
