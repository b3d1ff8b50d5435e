"""Diagnostic (not a test): per-problem trajectory deviation GPU vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import problems as P
from oracle import oracle as O
import lsq_amd as lsq
import test_gpu_parity as T
for opt, sol, sparse in T.GRID + [("dogleg", "cholesky", False), ("lm", "cholesky", False)]:
    probs = P.minpack_cholesky() if sol == "cholesky" else P.minpack_all()
    for p in probs:
        rg = T.gpu_run(p, T.OPT[opt][0], T.SOL[sol][0](), sparse)
        ro = T.oracle_run(p, T.OPT[opt][1], T.SOL[sol][1], sparse)
        k = min(rg.iterations, ro.iterations)
        dev = [np.max(np.abs(rg.trace["x"][i] - ro.trace["x"][i])) / max(1, np.max(np.abs(ro.trace["x"][i]))) for i in range(k)]
        same = (rg.iterations == ro.iterations and rg.mul_calls == ro.mul_calls and rg.f_calls == ro.f_calls)
        mx = max(dev) if dev else 0
        if not same or mx > 1e-9:
            print("%-7s %-8s %-5s %-30s it %3d/%3d mul %4d/%4d  maxdev %.2e first>1e-9 at %s  final dx %.2e ssr %.2e/%.2e" % (
                opt, sol, sparse, P.label(p), rg.iterations, ro.iterations, rg.mul_calls, ro.mul_calls, mx,
                next((i for i, d in enumerate(dev) if d > 1e-9), None), np.max(np.abs(rg.minimizer - ro.minimizer)), rg.ssr, ro.ssr))
