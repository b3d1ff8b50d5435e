#!/usr/bin/env python3
"""Random rank-deficient operands through QR() (Dogleg form): the certificate must refuse, the pivoted sweep must find
the rank and the minimum-norm solution (compared with LAPACK's SVD-based minimum-norm solution via numpy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
t0 = time.time()
for it in range(count):
    n = int(rng.choice([2, 5, 16, 17, 20, 33, 64, 65, 100, 130, 200, 300, 513]))
    m = int(rng.choice([n, n + 3, 2 * n + 1, 1000, 5000, 20000, 40000, 100000]))
    m = max(m, n)
    if m * n > 2e7:
        m = max(n, int(2e7 // n))
    k = int(rng.integers(1, n)) if n > 1 else 1
    A = (rng.standard_normal((m, k)) @ rng.standard_normal((k, n))) / np.sqrt(m)
    y = rng.standard_normal(m)
    J = lsq.DeviceMatrix(ctx, A)
    x = lsq.DeviceVector(ctx, n)
    try:
        sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
        sv.ldiv_(x, lsq.DeviceVector(ctx, m, y))
        info = sv.info()
        ref = np.linalg.lstsq(A, y, rcond=None)[0]
        err = np.linalg.norm(x.get() - ref) / np.linalg.norm(ref)
        if info["qr_rank"] != k or info["qr_path"] == "two-stage-certified" or not (err <= 1e-7):
            bad += 1
            print("FAIL", m, n, k, "relerr %.3e" % err, info, flush=True)
        sv.free()
    except Exception as e:   # noqa
        bad += 1
        print("EXC ", m, n, k, repr(e)[:200], flush=True)
    J.free()
print("cases %d, failures %d, %.1f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
