"""Generates tests/golden/count_stable.json: which of the 162 runs of the reference's MINPACK grid
(test/nonlinearsolvers.jl:505-595; the runs of minpack_oracle.json) keep IDENTICAL iteration / f / g / mul counts,
accept patterns and inner-solver counts under every summation-order model of the oracle
(oracle/lsq_oracle.c: orc_set_sum_mode 0..5 -- index order, Base.mapreduce, 4 / 8 / 16 SIMD lanes, extended-precision
nrm2).

Why: the reference's trajectory depends on how Julia's stdlib associates its sums (`sum(abs2, .)`, BLAS nrm2 / gemv),
which cannot be observed here (no Julia).  A run whose counts survive all of those orders is pinned as firmly as is
possible without a Julia run: whatever order the real stdlib uses among the modelled ones, the counts are these.  The
other runs are round-off chaotic (LSMR far past the loss of orthogonality on ill-conditioned Jacobians): for them the
oracle's counts are ONE valid outcome, not THE outcome.

`stable_under_wave_trees` additionally requires the same signature under mode 6 (64-lane halving trees at EVERY
reduction, including the sparse products and wdot that the reference writes as sequential loops): the class of orders
the wavefront reductions of the HIP fast path use.  `robust` further requires the same signature under 16 random
summation orders (modes 100..115) and 32 random last-bit perturbations of every reduction result (modes 1000..1031: the
effect of any algebraically equivalent reformulation, e.g. the 1/beta the fast kernels fold into the next product).
The fast-path parity test (tests/test_a_gpu_contract.py::test_minpack_fast_kernels) compares counts on the robust set.

Run from the repo root:  python tests/golden/make_count_stable.py
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

IDS = {"dogleg": O.DOGLEG, "lm": O.LM, "qr": O.QR, "cholesky": O.CHOLESKY, "lsmr": O.LSMR}


def signature(rec, prob, mode):
    name, f, g, x0 = prob
    n = len(x0)
    O.set_sum_mode(mode)
    try:
        J = (O.Mat(csc=(*P.full_csc_pattern(n, n), np.zeros(n * n))) if rec["sparse"] else O.Mat(dense=np.zeros((n, n))))
        ff, gg = P.wrap_dense(f, g, n, n)
        r = O.optimize(IDS[rec["optimizer"]], IDS[rec["solver"]], J, x0, ff, gg, trace=True, trace_x=False)
    finally:
        O.set_sum_mode(0)
    return dict(iterations=r.iterations, f_calls=r.f_calls, g_calls=r.g_calls, mul_calls=r.mul_calls,
                converged=bool(r.converged), accept=[int(v) for v in r.trace["accept"]],
                inner=[int(v) for v in r.trace["inner"]], ssr_ok=bool(r.ssr <= 1e-3))


def main():
    gold = json.load(open(os.path.join(HERE, "minpack_oracle.json")))
    probs = {P.label(p): p for p in P.minpack_all()}
    modes = sorted(O.SUM_MODES)
    out, nstable = [], 0
    for rec in gold["runs"]:
        sigs = {m: signature(rec, probs[rec["problem"]], m) for m in modes}
        differing = [m for m in modes if sigs[m] != sigs[0]]
        counts_only = [m for m in modes if any(sigs[m][k] != sigs[0][k] for k in ("iterations", "f_calls", "g_calls", "mul_calls"))]
        stable = not differing
        nstable += stable
        fast = signature(rec, probs[rec["problem"]], O.FAST_PATH_MODE)
        perturbed = [m for m in O.RANDOM_ORDER_MODES + O.ROUNDING_NOISE_MODES
                     if signature(rec, probs[rec["problem"]], m) != sigs[0]]
        out.append(dict(problem=rec["problem"], optimizer=rec["optimizer"], solver=rec["solver"], sparse=rec["sparse"],
                        stable=stable, stable_under_wave_trees=bool(stable and fast == sigs[0]),
                        robust=bool(stable and fast == sigs[0] and not perturbed), perturbed_modes_differing=perturbed,
                        modes_differing=differing, modes_with_other_counts=counts_only,
                        outcome_pin_holds_in_every_mode=all(s["ssr_ok"] for s in sigs.values()),
                        iterations_by_mode=[sigs[m]["iterations"] for m in modes],
                        mul_calls_by_mode=[sigs[m]["mul_calls"] for m in modes]))
    with open(os.path.join(HERE, "count_stable.json"), "w") as fh:
        json.dump(dict(source="oracle/lsq_oracle.c under orc_set_sum_mode(0..5) (NOT a Julia run)",
                       modes={str(k): v for k, v in O.SUM_MODES.items()}, stable=nstable, total=len(out), runs=out), fh, indent=0)
    print("count-stable: %d of %d runs; also under 64-lane trees everywhere (mode %d): %d; robust (also %d random orders "
          "and %d last-bit perturbations of every reduction): %d"
          % (nstable, len(out), O.FAST_PATH_MODE, sum(r["stable_under_wave_trees"] for r in out),
             len(O.RANDOM_ORDER_MODES), len(O.ROUNDING_NOISE_MODES), sum(r["robust"] for r in out)))
    for sol in ("qr", "cholesky", "lsmr"):
        sub = [r for r in out if r["solver"] == sol]
        print("  %-8s %d / %d stable; outcome pin (ssr <= 1e-3) holds in every mode: %d / %d"
              % (sol, sum(r["stable"] for r in sub), len(sub), sum(r["outcome_pin_holds_in_every_mode"] for r in sub), len(sub)))


if __name__ == "__main__":
    main()
