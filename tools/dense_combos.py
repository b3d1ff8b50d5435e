#!/usr/bin/env python3
"""Outer-iteration time of every dense optimizer x solver combination on the tanh model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsq_amd as lsq
ctx = lsq.Context(0)
L = lsq._lib
shapes = [(4096, 512), (16384, 2048)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for m, n in shapes:
    for oname, opt in (("LM", L.LEVENBERG_MARQUARDT), ("Dogleg", L.DOGLEG)):
        for sname, sol in (("QR", L.QR), ("Cholesky", L.CHOLESKY), ("LSMR", L.LSMR)):
            pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
            pr.reset()
            pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
            best = None
            for _ in range(3):
                pr.reset()
                t0 = time.perf_counter()
                r = pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False)
                ctx.sync()
                ms = (time.perf_counter() - t0) / max(r.iterations, 1) * 1e3
                best = ms if best is None else min(best, ms)
            print("%6dx%-5d %-6s %-9s %8.3f ms / outer iteration   ssr %.6e  (%d iterations)" % (m, n, oname, sname, best, r.ssr, r.iterations), flush=True)
            pr.close()
