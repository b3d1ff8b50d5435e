#!/usr/bin/env python3
"""Randomised shapes through the dense solvers (QR / Cholesky, undamped and damped) against numpy: the launch sequence
depends on the shape in many ways (one-workgroup / two-stage / TSQR, slab counts, ragged panels), this sweeps the
thresholds.  Prints failures; exit code 1 if any."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsq_amd as lsq
ctx = lsq.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
edges_m = [15, 16, 17, 63, 64, 65, 127, 129, 255, 257, 1000, 2047, 2048, 2049, 4095, 4097, 8191, 8193, 16383, 16385, 20479, 20481,
           32767, 32769, 50000, 131071, 131073, 300000]
edges_n = [1, 2, 3, 7, 8, 9, 12, 13, 15, 16, 17, 20, 21, 24, 25, 28, 29, 31, 32, 33, 40, 63, 64, 65, 100, 127, 128, 129, 200, 257]
bad = 0
t0 = time.time()
for it in range(count):
    n = int(rng.choice(edges_n)) if rng.random() < 0.7 else int(rng.integers(1, 300))
    m = int(rng.choice(edges_m)) if rng.random() < 0.7 else int(rng.integers(n, 200000))
    if m < n:
        m = n + int(rng.integers(0, 50))
    if m * n > 3e7:
        m = max(n, int(3e7 // n))
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.05
    J = lsq.DeviceMatrix(ctx, A)
    x = lsq.DeviceVector(ctx, n)
    ref0 = np.linalg.lstsq(A, y, rcond=None)[0]
    refd = np.linalg.solve(A.T @ A + np.diag(damp), A.T @ y)
    for solver, name in ((lsq.QR(), "QR"), (lsq.Cholesky(), "Chol")):
        for for_lm in (False, True):
            try:
                sv = lsq.AllocatedSolver(J, solver, for_lm=for_lm)
                if for_lm:
                    sv.ldiv_(x, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
                    ref = refd
                else:
                    sv.ldiv_(x, lsq.DeviceVector(ctx, m, y))
                    ref = ref0
                got = x.get()
                err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)
                cond = np.linalg.cond(A) if n <= 64 and m <= 5000 else 10.0
                tol = 1e-9 * max(1.0, cond * cond if name == "Chol" and not for_lm else cond)
                if not np.isfinite(err) or err > tol:
                    bad += 1
                    print("FAIL", name, "for_lm" if for_lm else "plain", m, n, "relerr %.3e" % err, sv.info(), flush=True)
                sv.free()
            except Exception as e:   # noqa
                bad += 1
                print("EXC ", name, for_lm, m, n, repr(e)[:200], flush=True)
    J.free()
print("shapes %d, failures %d, %.1f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
