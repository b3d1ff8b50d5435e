#!/usr/bin/env python3
"""Generates tests/golden/nist_outcomes.json: for every NIST StRD problem and starting point the reference's tests hold
(tests/golden/nist.json), (1) what three INDEPENDENT third-party solvers do from that start and (2) the class
(tests/nist_cases.py) the CPU oracle assigns to each of the 16 runs {Dogleg, LM} x {QR, Cholesky, LSMR dense, LSMR CSC} x
{central, analytic Jacobian}.

(1) is evidence, not an oracle: scipy.optimize.least_squares(method="lm") is MINPACK's lmder (the algorithm the reference's
LevenbergMarquardt descends from), "dogbox" a dogleg method, "trf" a reflective trust-region method.  They tell which
starts are hard for ANY implementation (NIST's "start 1" of the higher-difficulty problems); where they end elsewhere
(MINPACK on BoxBOD start 1: ssr 9771.5; dogbox on MGH09 start 1: ssr 1.79454e-3) it is on the same asymptotic plateaus
the reference's algorithms run onto.  The plateaus themselves are pinned independently of any trust-region code: they are
the minima of LIMIT MODELS (nist_cases.LIMIT_MODELS), fitted here by scipy (`plateaus`).

(2) is produced by the oracle, NOT by a Julia run (there is no Julia here): tests/test_oracle.py re-derives every class
with its evidence each time and compares; tests/test_a_gpu_contract.py holds the HIP path to the same table.  The oracle
is additionally run under every summation-order model of the stdlib reductions (orc_set_sum_mode 1..5) and under
one-rounding-error perturbations of every reduction result: a class that changes there is a round-off accident, not a
property of the algorithm; `order_dependent` lists those runs with every class seen, and an implementation whose sums
associate differently (Julia's stdlib, the HIP kernels) may legitimately land on any of them.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import nist            # noqa: E402
import nist_cases as NC  # noqa: E402
from oracle import oracle as O  # noqa: E402


NOISE = 8       # how many of the oracle's rounding-noise modes every run is repeated under


def independent(p, start):
    from scipy.optimize import least_squares

    def fun(b):
        o = np.zeros(p.m)
        p.f(o, b)
        return o

    def jac(b):
        J = np.zeros((p.m, p.n), order="F")
        p.g(J, b)
        return J

    out = {}
    for meth in ("lm", "dogbox", "trf"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = least_squares(fun, start, jac=jac, method=meth, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000,
                              x_scale="jac" if meth == "lm" else 1.0)
        out[meth] = {"hit": NC.hit(p, r.x), "nfev": int(r.nfev), "ssr": float(2 * r.cost)}
    return out


def limit_model_minimum(p, model, c0):
    """Minimum ssr of a LIMIT MODEL (fewer parameters: what the full model degenerates to when parameters run off to
    infinity), fitted by scipy alone -- neither the oracle nor the HIP path is involved."""
    from scipy.optimize import least_squares
    r = least_squares(lambda c: p.y - eval(model, {"x": p.x, "c": c, "exp": np.exp}), np.array(c0), xtol=1e-15, ftol=1e-15,
                      gtol=1e-15, max_nfev=20000)
    return float(2 * r.cost)


def main():
    probs = nist.problems()
    run = NC.oracle_runner()
    out = {"made_by": "tests/golden/make_nist_outcomes.py (oracle classes; scipy %s as the independent solvers)"
                      % __import__("scipy").__version__,
           "settings": dict(NC.NIST_KW, iterations=NC.CAP, success="norm(minimizer - certified) <= 1e-3"),
           "independent": {}, "plateaus": {}, "limit_models": {}, "classes": {}, "order_dependent": {}}
    for p in probs:
        for si, s in enumerate(p.starts):
            out["independent"]["%s/%d" % (p.name, si)] = independent(p, s)
        for name, (what, model, c0) in NC.LIMIT_MODELS.get(p.name, {}).items():
            out["plateaus"].setdefault(p.name, {})[name] = limit_model_minimum(p, model, c0)
            out["limit_models"]["%s/%s" % (p.name, name)] = {"limit": what, "model": model}
    print("independent solvers done; plateaus:", out["plateaus"], flush=True)
    for (opt, solver, storage) in NC.CONFIGS:
        for jac in ("central", "analytic"):
            table = {}
            for p in probs:
                for si in range(len(p.starts)):
                    cls, ev = NC.classify(p, si, opt, solver, storage, jac, run, out["plateaus"])
                    table["%s/%d" % (p.name, si)] = {"class": cls, **ev} if cls != "hit" else {"class": "hit"}
                    # the same run under the summation-order models of the stdlib reductions (1..5) and under one-rounding-error
                    # perturbations of every reduction result (1000..): a class that changes there is decided by round-off,
                    # not by the algorithm -- such runs are listed with every class seen (`order_dependent`)
                    seen = {cls}
                    for mode in (1, 2, 3, 4, 5) + tuple(O.ROUNDING_NOISE_MODES[:NOISE]):
                        O.set_sum_mode(mode)
                        try:
                            seen.add(NC.classify(p, si, opt, solver, storage, jac, run, out["plateaus"])[0])
                        finally:
                            O.set_sum_mode(0)
                    if len(seen) > 1:
                        out["order_dependent"].setdefault(NC.config_key(opt, solver, storage, jac), {})[
                            "%s/%d" % (p.name, si)] = sorted(seen)
            key = NC.config_key(opt, solver, storage, jac)
            out["classes"][key] = table
            print(key, {k: v["class"] for k, v in table.items() if v["class"] != "hit"}, flush=True)
    with open(NC.OUTCOMES, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", NC.OUTCOMES)


if __name__ == "__main__":
    main()
