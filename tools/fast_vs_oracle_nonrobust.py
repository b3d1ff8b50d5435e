#!/usr/bin/env python3
"""How far the fast (tree-reduction) kernels drift from the oracle on the NON-robust runs of the MINPACK grid (the ones whose
counts the oracle itself does not keep under reordered sums): iteration difference and distance of the minimisers.  Input for the
bounds in tests/test_a_gpu_contract.py::test_minpack_fast_kernels."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lsq_amd as lsq  # noqa: E402
import problems as P  # noqa: E402
import test_a_gpu_contract as T  # noqa: E402

cs = json.load(open(os.path.join(ROOT, "tests", "golden", "count_stable.json")))
stable = {(r["problem"], r["optimizer"], r["solver"], r["sparse"]): r["robust"] for r in cs["runs"]}
lsq.set_exact(False)
worst = {}
for opt, sol, sparse in T.GRID + [("dogleg", "cholesky", False), ("lm", "cholesky", False)]:
    probs = P.minpack_cholesky() if sol == "cholesky" else P.minpack_all()
    for p in probs:
        key = (P.label(p), opt, sol, sparse)
        if stable[key]:
            continue
        rg = T.gpu_run(p, T.OPT[opt][0], T.SOL[sol][0](), sparse)
        ro = T.oracle_run(p, T.OPT[opt][1], T.SOL[sol][1], sparse)
        dx = float(np.max(np.abs(rg.minimizer - ro.minimizer)) / max(1.0, float(np.max(np.abs(ro.minimizer)))))
        print("%-28s %-7s %-9s sparse=%d  it %4d vs %4d  conv %d/%d  ssr %.2e / %.2e  |dx| %.2e  mul %d vs %d" % (
            key[0], opt, sol, sparse, rg.iterations, ro.iterations, rg.converged, ro.converged, rg.ssr, ro.ssr, dx, rg.mul_calls, ro.mul_calls), flush=True)
lsq.set_exact(None)
