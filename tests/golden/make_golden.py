"""Generates tests/golden/minpack_oracle.json.

The reference is Julia and cannot run in this environment (no julia binary, no network), so these
vectors are produced by the CPU ORACLE (oracle/lsq_oracle.c), after the oracle itself has been
pinned by tests/test_oracle.py against the hand-derived KATs, scipy LSMR and LAPACK.  They record
iteration / f / g / mul counts and the minimiser for the reference's MINPACK grid
(test/nonlinearsolvers.jl:505-595) so that (a) the oracle cannot drift silently and (b) the HIP
path has fixed numbers to match.  They are NOT a Julia run: trajectory parity with Julia stays
"unpinned" (DESIGN.md).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import problems as P  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    runs = []
    grid = [("dogleg", "qr", False), ("lm", "qr", False), ("dogleg", "lsmr", False),
            ("lm", "lsmr", False), ("dogleg", "lsmr", True), ("lm", "lsmr", True)]
    ids = {"dogleg": O.DOGLEG, "lm": O.LM, "qr": O.QR, "cholesky": O.CHOLESKY, "lsmr": O.LSMR}
    for opt, sol, sparse in grid:
        for p in P.minpack_all():
            runs.append((opt, sol, sparse, p))
    for opt in ("dogleg", "lm"):
        for p in P.minpack_cholesky():
            runs.append((opt, "cholesky", False, p))
    out = []
    for opt, sol, sparse, p in runs:
        name, f, g, x0 = p
        n = len(x0)
        J = (O.Mat(csc=(*P.full_csc_pattern(n, n), np.zeros(n * n))) if sparse
             else O.Mat(dense=np.zeros((n, n))))
        ff, gg = P.wrap_dense(f, g, n, n)
        r = O.optimize(ids[opt], ids[sol], J, x0, ff, gg, trace=False)
        out.append(dict(problem=P.label(p), optimizer=opt, solver=sol, sparse=sparse,
                        iterations=r.iterations, f_calls=r.f_calls, g_calls=r.g_calls,
                        mul_calls=r.mul_calls, converged=r.converged, ssr=r.ssr,
                        x=[float(v) for v in r.minimizer]))
    with open(os.path.join(HERE, "minpack_oracle.json"), "w") as fh:
        json.dump(dict(source="oracle/lsq_oracle.c (NOT a Julia run)", runs=out), fh, indent=0)
    print("wrote", len(out), "runs")


if __name__ == "__main__":
    main()
