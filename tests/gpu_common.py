"""Shared pieces of the GPU parity tests (tests/test_a_gpu_contract.py, test_b_gpu_kernels.py, test_zz_gpu_stress.py):
random sparse operands and the trajectory runners / comparator against the oracle.  Tolerances (fp64; summation order
differs: tree reductions vs the serial loops of the reference):
    kernels                         |err| <= 1e-12 * scale
    one linear solve (ldiv!)        rel 1e-9 (direct), LSMR: identical iteration count, rel 1e-8
    trust-region trajectories       small problems run the reference-order kernels (lsq_exact.hip):
                                    identical iteration / f / g / mul counts, accept pattern and inner
                                    iteration counts on the whole reference grid, iterates equal to
                                    1e-12 (LSMR: bitwise-equal arithmetic; QR / Cholesky: the dense
                                    factorisations use wave-parallel dot products, so 1e-5 on the
                                    ill-conditioned instances);
                                    the fast (tree-reduction) kernels on the same small problems:
                                    the reference's outcome pins, and identical counts wherever the
                                    solve is not round-off chaotic
"""
import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle import oracle as O

lsq = pytest.importorskip("lsq_amd")


def rand_csc(m, n, density, seed):
    rng = np.random.default_rng(seed)
    S = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    S.sort_indices()
    return S


# ------------------------------------------------------------------- trust-region trajectories
def gpu_run(p, optimizer, solver, sparse=False, **kw):
    name, f, g, x0 = p[:4]
    n = len(x0)
    if sparse:
        m_, n_, colptr, rowval = P.full_csc_pattern(n, n)
        J = sp.csc_matrix((np.zeros(n * n), rowval, colptr), shape=(n, n))

        def g_(Jm, x):
            g(Jm.data.reshape((n, n), order="F"), x)
    else:
        J = np.zeros((n, n), order="F")
        g_ = g
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(n), f_=f, g_=g_, J=J)
    return lsq.optimize_(nls, optimizer(solver), full_trace=True, **kw)


def oracle_run(p, optimizer, solver, sparse=False, **kw):
    name, f, g, x0 = p[:4]
    n = len(x0)
    J = (O.Mat(csc=(*P.full_csc_pattern(n, n), np.zeros(n * n))) if sparse else O.Mat(dense=np.zeros((n, n))))
    ff, gg = P.wrap_dense(f, g, n, n)
    return O.optimize(optimizer, solver, J, x0, ff, gg, **kw)


OPT = {"dogleg": (lsq.Dogleg, O.DOGLEG), "lm": (lsq.LevenbergMarquardt, O.LM)} if hasattr(lsq, "Dogleg") else {}
SOL = {"qr": (lsq.QR, O.QR), "cholesky": (lsq.Cholesky, O.CHOLESKY), "lsmr": (lsq.LSMR, O.LSMR)} if OPT else {}


def compare(rg, ro, label, xtol=1e-12):
    assert rg.ssr <= 1e-3, (label, rg.ssr)                       # the reference's own pin
    assert rg.iterations == ro.iterations, (label, rg.iterations, ro.iterations)
    assert (rg.f_calls, rg.g_calls, rg.mul_calls) == (ro.f_calls, ro.g_calls, ro.mul_calls), label
    assert (rg.converged, rg.x_converged, rg.f_converged, rg.g_converged) == \
           (ro.converged, ro.x_converged, ro.f_converged, ro.g_converged), label
    assert np.array_equal(rg.trace["accept"], ro.trace["accept"]), label
    assert np.array_equal(rg.trace["inner"], ro.trace["inner"]), label
    for k in range(ro.iterations):
        xr = ro.trace["x"][k]
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= xtol * max(1.0, np.max(np.abs(xr))), (label, k)


GRID = [("dogleg", "qr", False), ("lm", "qr", False), ("dogleg", "lsmr", False), ("lm", "lsmr", False),
        ("dogleg", "lsmr", True), ("lm", "lsmr", True)]


# ------------------------------------------------------------------- full-size trajectories (C4, C3)
SSR_NOISE = 1e-12


def compare_until_roundoff(rg, ro, xtol=1e-8, ssr_rtol=1e-9, ssr0=None):
    """Two traced runs of the same problem iteration by iteration (HIP vs oracle, zero tolerances = the bench schedule, where the
    loop runs on past convergence).  Iterate and ssr must agree at EVERY iteration; the accept decision and the LSMR inner
    count must agree at every iteration up to the first ROUND-OFF-DECIDED one: an iteration at which the two runs disagree
    about acceptance and the run that accepted changed the objective by less than SSR_NOISE = 1e-12 relative -- i.e. the gain
    ratio rho = (ssr - trial_ssr) / (ssr - predicted_ssr) (levenberg_marquardt.jl:118-122) is a quotient of rounding errors
    of two sums over m = 10^6 squares (sqrt(m) eps = 2e-13), and which side of MIN_STEP_QUALITY it lands on depends on the order
    in which those sums are associated (unobservable for the reference itself: SURVEY 8c).  From there on Delta differs by
    design (x3 after an accepted step, /2 after a refused one), so only iterates and ssr are compared.  A disagreement at a
    step that moved the objective by more than that is a failure.  Returns the index of that iteration (None: all agree)."""
    assert rg.iterations == ro.iterations
    excused = None
    prev = {"g": ssr0, "o": ssr0}
    for k in range(ro.iterations):
        ag, ao = int(rg.trace["accept"][k]), int(ro.trace["accept"][k])
        sg, so = float(rg.trace["ssr"][k]), float(ro.trace["ssr"][k])
        if excused is None:
            if ag != ao:
                s_prev, s_new = (prev["g"], sg) if ag else (prev["o"], so)
                assert s_prev is not None and abs(s_prev - s_new) <= SSR_NOISE * s_prev, \
                    ("accept decisions differ at a step that moved the objective", k, ag, ao, s_prev, s_new)
                excused = k
            else:
                assert rg.trace["inner"][k] == ro.trace["inner"][k], (k, rg.trace["inner"], ro.trace["inner"])
        assert abs(sg - so) <= ssr_rtol * so, (k, sg, so)
        xr = ro.trace["x"][k]
        # (from the round-off-decided iteration on one run has taken a step the other refused: |dx| of such a step is itself
        #  of the order 1e-8 here, so the iterates are held to 5x the tolerance there)
        tol_k = xtol if excused is None else 5.0 * xtol
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= tol_k * max(1.0, np.max(np.abs(xr))), k
        prev = {"g": sg, "o": so}
    return excused
