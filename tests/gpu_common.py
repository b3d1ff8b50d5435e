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
