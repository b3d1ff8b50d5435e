import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import lsq_amd as lsq
from oracle import oracle as O
ctx = lsq.default_context()
cases = [(300, 65, 65), (1100, 130, 130), (900, 100, 37), (700, 200, 1), (640, 128, 128), (2000, 321, 321), (2000, 321, 300), (500, 500, 500)]
for rnd in range(6):
    for (m, n, rank) in cases:
        rng = np.random.default_rng(300 + m + n + rank)
        A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))
        y = rng.standard_normal(m)
        J = lsq.DeviceMatrix(ctx, A)
        Jv = J.values().reshape((m, n), order="F")
        if not np.array_equal(Jv, A): print("J wrong right after upload", (m, n, rank), np.argwhere(Jv != A)[:3], flush=True)
        dxo = lsq.DeviceVector(ctx, n)
        xr, rk, *_ = O.qr_solve(A, y)
        for env in ("LSQ_QR_TWO_STAGE", "LSQ_QR_ONE_STAGE"):
            os.environ[env] = "1"
            sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
            _, nmul = sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
            r = sv.info()["qr_rank"]
            x = dxo.get()
            Jv = J.values().reshape((m, n), order="F")
            bad = np.argwhere(Jv != A)
            if len(bad):
                print("J CORRUPTED after", env, (m, n, rank), "entries", len(bad), "rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max(), flush=True)
            err = np.max(np.abs(x - xr))
            if r != rank or err > 1e-8:
                print("round", rnd, (m, n, rank), env, "rank", r, "err %.2e" % err, flush=True)
            if rank == n:
                damp = rng.random(n) + 0.01
                svd = lsq.AllocatedSolver(J, lsq.QR(), for_lm=True)
                svd.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
                st, xd, _, _ = O.ldiv(O.QR, O.Mat(dense=A), y, damp)
                e2 = np.max(np.abs(dxo.get() - xd))
                if e2 > 1e-8: print("round", rnd, (m, n, rank), env, "damped err %.2e" % e2, flush=True)
            del os.environ[env]
print("done")
