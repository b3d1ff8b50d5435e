#!/usr/bin/env python3
"""Round-3 GPU-suite failure (test_dense_shape_sweep: 2049 x 129, QR(), for_lm=True, relative error 0.32, no status):
reproduction and proof of the fix.

  python tools/repro/qr_race.py [--lib PATH] [--jitter US] [--solves N] [--m M --n N] [--fresh-every K]

Solves the SAME damped dense-QR problem N times (dense_qr.jl:56-88: the stacked (m+n) x n operand) and compares every
result with LAPACK (numpy) and, bit for bit, with the first result.  With --jitter the library inserts random host stalls
in front of its kernel launches (lsq_debug_set).  --lib tools/repro/liblsqhip_r03war.so (tools/repro/build_r03_war.sh)
runs round 3's CholeskyQR2 panel, whose k_cqr_top (side stream) stored S*R into A's panel triangle while k_cqr_pass<2>
(main stream) could still be reading Q1's top rows there: a stall of more than the ~50 us the 64-step LU takes, between
those two launches, makes slab 0 of pass 2 build the reflector's top 64 rows from S*R -- the wrong answer the driver saw.
Prints one JSON line; exit code 1 if any solve was wrong or differed from the first."""
import argparse
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--jitter", type=int, default=0)
ap.add_argument("--serial", type=int, default=0)
ap.add_argument("--solves", type=int, default=10000)
ap.add_argument("--m", type=int, default=2049)
ap.add_argument("--n", type=int, default=129)
ap.add_argument("--fresh-every", type=int, default=0, help="a new solver (new workspace, streams) every K solves; 0: one solver")
ap.add_argument("--solver", default="qr", choices=["qr", "chol"])
ap.add_argument("--plain", action="store_true", help="undamped solve (for_lm=False)")
args = ap.parse_args()
if args.lib:
    os.environ["LSQ_LIB_PATH"] = os.path.abspath(args.lib)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import lsq_amd as lsq  # noqa: E402

ctx = lsq.Context(0)
m, n = args.m, args.n
# the sweep's own operand for this shape is not reproduced (its generator state depends on the shapes before it); any
# full-rank operand takes the same launch sequence
rng = np.random.default_rng(2026)
A = rng.standard_normal((m, n)) / np.sqrt(m)
y = rng.standard_normal(m)
damp = rng.random(n) + 0.05
ref = np.linalg.lstsq(A, y, rcond=None)[0] if args.plain else np.linalg.solve(A.T @ A + np.diag(damp), A.T @ y)
J = lsq.DeviceMatrix(ctx, A)
x = lsq.DeviceVector(ctx, n)
dy, dd = lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp)
kind = lsq.QR() if args.solver == "qr" else lsq.Cholesky()
sv = lsq.AllocatedSolver(J, kind, for_lm=not args.plain)
lsq.debug_set(args.jitter, args.serial)
first, wrong, differ, worst, info = None, 0, 0, 0.0, None
first_wrong = None
for k in range(args.solves):
    if args.fresh_every and k and k % args.fresh_every == 0:
        sv.free()
        sv = lsq.AllocatedSolver(J, kind, for_lm=not args.plain)
    if args.plain:
        sv.ldiv_(x, dy)
    else:
        dd.set(damp)          # (ldiv! may overwrite damp: iterative_lsmr.jl:243 does, the dense solvers do not; keep it fixed)
        sv.ldiv_(x, dy, dd)
    got = x.get()
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    worst = max(worst, err) if np.isfinite(err) else float("inf")
    if not (err <= 1e-8):
        wrong += 1
        if first_wrong is None:
            first_wrong = {"solve": k, "rel_err": err, "info": sv.info()}
    if first is None:
        first, info = got, sv.info()
    elif not np.array_equal(got, first):
        differ += 1
lsq.debug_set(0, 0)
out = {"lib": args.lib or "product", "shape": [m, n], "solver": args.solver, "for_lm": not args.plain, "jitter_us": args.jitter,
       "serial": args.serial, "solves": args.solves, "wrong": wrong, "differ_from_first": differ, "worst_rel_err": worst,
       "first_wrong": first_wrong, "info": {k: info[k] for k in ("qr_path", "qr_panel", "chol_path") if k in info},
       "stalls_injected": lsq.debug_get()[2], "num_cus": ctx.num_cus if hasattr(ctx, "num_cus") else None,
       "fallback_giveups": ctx.fallback_stats() if hasattr(ctx, "fallback_stats") else None}
print(json.dumps(out))
sys.exit(1 if wrong or differ else 0)
