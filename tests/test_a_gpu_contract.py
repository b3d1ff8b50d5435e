"""GPU parity, tier 1 -- THE CONTRACT (runs first: `pytest -x` must not be able to lose these rows behind a stress test).
The HIP path, through the C ABI, against the CPU oracle on the same inputs: the BASELINE configurations at full size
(C4, C2, C3), the bench family's tanh model against the oracle, the reference's trajectory grid (MINPACK, golden fixtures,
known-answer trajectories, factor model, bounds, NIST certified values).  Tolerances: tests/gpu_common.py.
Then come tests/test_b_gpu_kernels.py (kernels and single solves), the row-sharding / RCCL files, and LAST
tests/test_zz_gpu_stress.py (shape sweeps, launch jitter, injected time-outs, busy neighbours)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle import oracle as O
from gpu_common import GRID, OPT, SOL, compare, gpu_run, lsq, oracle_run, rand_csc

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------- full BASELINE sizes (C2, C4): properties
def test_c2_dense_lm_cholesky_full_size(ctx):
    """C2: dense 4096 x 512, LevenbergMarquardt(Cholesky()) -- MFMA SYRK + blocked Cholesky path.
    Oracle comparison on the first iterations, then size-independent properties of one ldiv!:
    the normal equations hold, and the solve is run-to-run bit-identical."""
    m, n = 4096, 512
    pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED + 2, ctx=ctx)
    pr.reset()
    rg = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.CHOLESKY, trace=True, iterations=3)
    A = O.Mat(dense=pr.A.reshape((m, n), order="F"))
    J = O.Mat(dense=np.zeros((m, n)))
    f, g, ud, keep = O.tanh_model(A, pr.b)
    ro = O.optimize(O.LM, O.CHOLESKY, J, np.zeros(n), f, g, ud=ud, iterations=3)
    assert rg.iterations == ro.iterations == 3 and rg.mul_calls == ro.mul_calls
    assert np.array_equal(rg.trace["accept"], ro.trace["accept"])
    for k in range(3):
        assert np.max(np.abs(rg.trace["x"][k] - ro.trace["x"][k])) <= 1e-9 * max(1.0, np.max(np.abs(ro.trace["x"][k])))
    # one damped solve: (J'J + D) x = J'y to round-off, deterministic
    Jm = pr.A.reshape((m, n), order="F")
    Jd = lsq.DeviceMatrix(ctx, Jm)
    rng = np.random.default_rng(5)
    y, damp = rng.standard_normal(m), rng.random(n) + 0.1
    sv = lsq.AllocatedSolver(Jd, lsq.Cholesky(), for_lm=True)
    xs = []
    for _ in range(2):
        xo = lsq.DeviceVector(ctx, n)
        sv.ldiv_(xo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        xs.append(xo.get())
    assert np.array_equal(xs[0], xs[1])
    res = Jm.T @ (Jm @ xs[0] - y) + damp * xs[0]
    assert np.max(np.abs(res)) <= 1e-11 * np.max(np.abs(Jm.T @ y))
    pr.close()


@pytest.mark.parametrize("n,pc", [(10_000, 1000), (30_000, 333)])
def test_c4_sparse_full_size_properties(ctx, n, pc):
    """C4: sparse 10^6 x 10^4, nnz = 10^7 -- the kernels the bench times -- and the same entry count spread over n = 30000
    columns (x no longer fits in LDS: J*v takes the column-windowed k_sell_rows_wide), checked through
    size-independent properties: linearity and adjointness of the two products
    (<J x, y> == <x, J'y>), colsumabs2 against the product with unit vectors' squares, sampled rows of J*x bit for bit against
    a sequential left-to-right sum, run-to-run
    determinism of a full LM+LSMR solve, and the reference's convergence on the tanh model."""
    m = 1_000_000
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
    L = lsq.lib()
    rng = np.random.default_rng(1)
    x1, x2, y1 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
    # J currently holds zeros: load A's values into it through the model's g! at x = 0 (J = A)
    pr.reset()
    r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=1, x_tol=0, f_tol=0, g_tol=0)
    class _J:  # handle wrapper for mul_
        h = pr.J
    dx1, dx2, dy1 = (lsq.DeviceVector(ctx, len(v), v) for v in (x1, x2, y1))
    out = lambda k: lsq.DeviceVector(ctx, k)
    jx1 = lsq.mul_(out(m), _J, dx1).get()
    jx2 = lsq.mul_(out(m), _J, dx2).get()
    jsum = lsq.mul_(out(m), _J, lsq.DeviceVector(ctx, n, 2.0 * x1 - 3.0 * x2)).get()
    assert np.max(np.abs(jsum - (2 * jx1 - 3 * jx2))) <= 1e-12 * (1 + np.max(np.abs(jsum)))      # linearity
    jty = lsq.mul_(out(n), _J, dy1, trans=True).get()
    assert abs(np.dot(jx1, y1) - np.dot(x1, jty)) <= 1e-10 * np.linalg.norm(jx1) * np.linalg.norm(y1)  # adjoint
    # against the host CSC arrays (J = A .* (1 - tanh(0)^2) = A at the first iteration's x = 0)
    ref = lsq.synthetic.csc_matvec(m, pr.colptr, pr.rowval, pr.A, x1)
    assert np.max(np.abs(jx1 - ref)) <= 1e-12 * (1 + np.max(np.abs(ref)))
    cs = lsq.colsumabs2_(out(n), _J).get()
    assert np.allclose(cs, np.add.reduceat(pr.A * pr.A, pr.colptr[:-1]), rtol=1e-12)
    # one lane adds a row's products left to right (across the column windows when n > 12160): the reference's order
    Sr = sp.csc_matrix((pr.A, pr.rowval, pr.colptr), shape=(m, n)).tocsr()
    Sr.sort_indices()
    for i in rng.integers(0, m, 200):
        dot = 0.0
        for k in range(Sr.indptr[i], Sr.indptr[i + 1]):
            dot += Sr.data[k] * x1[Sr.indices[k]]
        assert jx1[i] == dot, i
    # determinism + convergence of the whole loop
    runs = []
    for _ in range(2):
        pr.reset()
        rr = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=30)
        runs.append((rr.iterations, rr.mul_calls, rr.ssr, rr.minimizer.copy()))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert np.array_equal(runs[0][3], runs[1][3])
    assert rr.converged and rr.iterations <= 10
    assert np.max(np.abs(rr.minimizer - pr.x_true)) < 0.05      # recovers the planted parameters
    pr.close()


@pytest.mark.parametrize("n,pc", [(10_000, 1000), (30_000, 333), (25_000, 400)], ids=["C4", "n30000_wide", "n25000_two_windows"])
def test_c4_full_size_matches_oracle(ctx, n, pc):
    """C4 AT FULL SIZE against the oracle: LevenbergMarquardt(LSMR()) on the bench's own problem (10^6 x 10^4,
    nnz 10^7, BASE_SEED) -- the kernels the headline times (k_sell_rows<EpiU>, k_sell_cols + k_combine, k_sell_rows_pair, the
    speculative gradient pass) -- and the same entry count over n = 30000 columns (k_sell_rows_wide: three column windows of 10000; two-launch tail) and over
    n = 25000 (two windows of 12500: the widest the kernel's LDS holds, the sparse_secondary shape of the bench) vs
    O.optimize on the same inputs: levenberg_marquardt.jl:72-140, iterative_lsmr.jl:238-259.
    (1) The reference's own run (default tolerances): identical iteration / f / g / mul counts, convergence flags, LSMR inner
        counts per outer iteration and accept pattern; every iterate to 1e-8 max(1, |x|_inf), ssr to 1e-9, Delta exactly equal
        (it is a product of the same factors when the decisions agree).
    (2) The bench schedule (8 iterations, zero tolerances: the loop runs on past convergence, where steps change the objective
        by ~1e-15 relative and rho is a quotient of rounding errors): gpu_common.compare_until_roundoff -- everything as in
        (1) up to the first round-off-decided iteration, iterates and ssr throughout; f and mul counts identical."""
    from gpu_common import compare_until_roundoff
    m = 1_000_000
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
    A = O.Mat(csc=(m, n, pr.colptr, pr.rowval, pr.A))
    J = O.Mat(csc=(m, n, pr.colptr, pr.rowval, np.zeros_like(pr.A)))
    f, g, ud, keep = O.tanh_model(A, pr.b)
    # (1) default tolerances
    pr.reset()
    rg = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, trace=True, iterations=30)
    ro = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=30)
    assert rg.iterations == ro.iterations and 4 <= ro.iterations <= 10, (rg.iterations, ro.iterations)
    assert (rg.converged, rg.x_converged, rg.f_converged, rg.g_converged) == (ro.converged, ro.x_converged, ro.f_converged, ro.g_converged)
    assert ro.converged
    assert (rg.f_calls, rg.g_calls, rg.mul_calls) == (ro.f_calls, ro.g_calls, ro.mul_calls)
    assert np.array_equal(rg.trace["inner"], ro.trace["inner"]), (rg.trace["inner"], ro.trace["inner"])
    assert np.array_equal(rg.trace["accept"], ro.trace["accept"]) and np.all(ro.trace["accept"] == 1)
    assert np.array_equal(rg.trace["delta"], ro.trace["delta"])
    np.testing.assert_allclose(rg.trace["ssr"], ro.trace["ssr"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(rg.trace["gnorm"], ro.trace["gnorm"], rtol=1e-7, atol=1e-13)
    for k in range(ro.iterations):
        xr = ro.trace["x"][k]
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= 1e-8 * max(1.0, np.max(np.abs(xr))), k
    assert np.max(np.abs(rg.minimizer - ro.minimizer)) <= 1e-8 * max(1.0, np.max(np.abs(ro.minimizer)))
    fc = pr.fcur.get()                                            # the residual the loop carries == the oracle's
    assert np.max(np.abs(fc - ro.fcur)) <= 1e-9 * max(1.0, np.max(np.abs(ro.fcur)))
    useful = ro.iterations
    # (2) the bench schedule
    pr.reset()
    rg = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, trace=True, iterations=8, x_tol=0.0, f_tol=0.0, g_tol=0.0)
    ro = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=8, x_tol=0.0, f_tol=0.0, g_tol=0.0)
    assert rg.iterations == ro.iterations == 8
    assert (rg.f_calls, rg.mul_calls) == (ro.f_calls, ro.mul_calls)
    excused = compare_until_roundoff(rg, ro, ssr0=float(np.sum(pr.b * pr.b)))
    assert excused is None or excused >= useful, (excused, useful)     # (only PAST convergence may round-off decide)
    pr.close()


def test_c3_full_size_dogleg_qr_matches_oracle(ctx):
    """C3 AT FULL SIZE against the oracle: 2 Dogleg(QR()) iterations on the 16384 x 2048 tanh problem,
    dogleg.jl:77-199 + dense_qr.jl:30-42.  The oracle's loop, dense products, rank decision (dlaic1), Q'b and triangular solve
    are its own; only its dgeqp3 is served by LAPACK (O.use_lapack_geqp3: scipy's dgeqp3, the routine Julia calls -- the
    scalar restatement of it needs minutes at 1.3e11 flops).  Identical counts / accept pattern; iterates to 1e-9; Delta and
    rho to 1e-8 (a ratio of differences of sums over 16384 rows)."""
    m, n = 16384, 2048
    pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED + 3, ctx=ctx)
    pr.reset()
    rg = pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, trace=True, iterations=2, x_tol=0.0, f_tol=0.0, g_tol=0.0)
    A = O.Mat(dense=pr.A.reshape((m, n), order="F"))
    J = O.Mat(dense=np.zeros((m, n)))
    f, g, ud, keep = O.tanh_model(A, pr.b)
    O.use_lapack_geqp3(True)
    try:
        ro = O.optimize(O.DOGLEG, O.QR, J, np.zeros(n), f, g, ud=ud, iterations=2, x_tol=0.0, f_tol=0.0, g_tol=0.0)
    finally:
        O.use_lapack_geqp3(False)
    assert rg.iterations == ro.iterations == 2
    assert (rg.f_calls, rg.g_calls, rg.mul_calls) == (ro.f_calls, ro.g_calls, ro.mul_calls)
    assert np.array_equal(rg.trace["accept"], ro.trace["accept"]) and np.array_equal(rg.trace["inner"], ro.trace["inner"])
    np.testing.assert_allclose(rg.trace["ssr"], ro.trace["ssr"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(rg.trace["delta"], ro.trace["delta"], rtol=1e-8, atol=0)
    np.testing.assert_allclose(rg.trace["rho"], ro.trace["rho"], rtol=1e-8, atol=0)
    for k in range(2):
        xr = ro.trace["x"][k]
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr))), k
    pr.close()


def test_c3_dense_dogleg_qr_full_size(ctx):
    """C3: dense 16384 x 2048, Dogleg(QR()).  The CPU oracle's plain-C pivoted QR needs minutes at
    this size, so the full-size checks are properties: the QR least-squares solve agrees with
    LAPACK (numpy lstsq) and satisfies the normal equations, the detected rank is n, and three
    Dogleg iterations on the tanh model decrease the objective with rho near 1."""
    m, n = 16384, 2048
    pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED + 3, ctx=ctx)
    Jm = pr.A.reshape((m, n), order="F")
    Jd = lsq.DeviceMatrix(ctx, Jm)
    y = np.random.default_rng(9).standard_normal(m)
    sv = lsq.AllocatedSolver(Jd, lsq.QR(), for_lm=False)
    xo = lsq.DeviceVector(ctx, n)
    _, nmul = sv.ldiv_(xo, lsq.DeviceVector(ctx, m, y))
    x = xo.get()
    assert nmul == 1 and sv.info()["qr_rank"] == n
    xl = np.linalg.lstsq(Jm, y, rcond=None)[0]
    assert np.linalg.norm(x - xl) <= 1e-11 * np.linalg.norm(xl)
    assert np.max(np.abs(Jm.T @ (Jm @ x - y))) <= 1e-11 * np.max(np.abs(Jm.T @ y))
    pr.reset()
    r = pr.optimize(lsq._lib.DOGLEG, lsq._lib.QR, iterations=3, trace=True)
    assert r.iterations == 3 and np.all(np.diff(r.trace["ssr"]) < 0) and np.all(r.trace["accept"] == 1)
    pr.close()


# --------------------------------------------------------------- synthetic model (bench family)
@pytest.mark.parametrize("sparse,opt,sol,big", [(True, "lm", "lsmr", False), (False, "lm", "cholesky", False),
                                                (False, "dogleg", "qr", False), (True, "dogleg", "lsmr", False),
                                                (True, "lm", "lsmr", True), (True, "dogleg", "lsmr", True),
                                                (True, "lm", "lsmr", "segments"), (True, "dogleg", "lsmr", "segments"),
                                                (True, "lm", "lsmr", "wide"), (True, "dogleg", "lsmr", "wide")])
def test_tanh_model_matches_oracle(ctx, sparse, opt, sol, big, monkeypatch):
    """Reduced-size C4/C2/C3 family: device f!/g! + device solver vs the oracle's C model.
    `big` is large enough (m > 131072 rows, nnz >= 2^20) to take the paths C4 takes: the sliced
    layouts (lsq_sell.h), or with "segments" the LDS-staged J*v kernel and the row-window-blocked J'*u."""
    m, n, per_col = (20000, 200, 100) if sparse else (1500, 48, None)
    if big:
        m, n, per_col = 300000, 2000, 600
    if big == "segments":   # the segment kernels (LDS-staged stream / row windows) instead of the sliced layouts
        monkeypatch.setenv("LSQ_NO_SELL", "1")
    if big == "wide":       # J*v with x in four column windows (k_sell_rows_wide: what n > 12160 gets), column-scaled handle
        monkeypatch.setenv("LSQ_SELL_XMAX", "500")
        monkeypatch.setenv("LSQ_SELL_WIDE", "1")
    pr = lsq.synthetic.TanhProblem(m, n, sparse=sparse, per_col=per_col, seed=7, ctx=ctx)
    monkeypatch.delenv("LSQ_NO_SELL", raising=False)
    monkeypatch.delenv("LSQ_SELL_XMAX", raising=False)
    monkeypatch.delenv("LSQ_SELL_WIDE", raising=False)
    pr.reset()
    okind = lsq._lib.LEVENBERG_MARQUARDT if opt == "lm" else lsq._lib.DOGLEG
    skind = {"lsmr": lsq._lib.LSMR, "cholesky": lsq._lib.CHOLESKY, "qr": lsq._lib.QR}[sol]
    rg = pr.optimize(okind, skind, trace=True, iterations=50)
    A = (O.Mat(csc=(m, n, pr.colptr, pr.rowval, pr.A)) if sparse else O.Mat(dense=pr.A.reshape((m, n), order="F")))
    J = (O.Mat(csc=(m, n, pr.colptr, pr.rowval, np.zeros_like(pr.A))) if sparse else O.Mat(dense=np.zeros((m, n))))
    f, g, ud, keep = O.tanh_model(A, pr.b)
    ro = O.optimize(OPT[opt][1], SOL[sol][1], J, np.zeros(n), f, g, ud=ud, iterations=50)
    assert rg.iterations == ro.iterations and rg.mul_calls == ro.mul_calls
    assert rg.converged == ro.converged and rg.ssr == pytest.approx(ro.ssr, rel=1e-9)
    assert np.array_equal(rg.trace["inner"], ro.trace["inner"])
    for k in range(ro.iterations):
        xr = ro.trace["x"][k]
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= 1e-8 * max(1.0, np.max(np.abs(xr)))
    pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("big", [False, True])
def test_lm_fused_setup_matches_separate_kernels(ctx, big, monkeypatch):
    """LM's damping + projected gradient norm + LSMR's setup as ONE launch (k_lm_lsmr_setup) against the separate
    kernels it replaces (k_lm_damp_grad + k_lsmr_setup; LSQ_LSMR_SEPARATE_SETUP=1).  Same arithmetic per element; the only
    difference is that sum(v~^2) is grouped per 1024 instead of per 256 elements: identical counts, accept pattern, inner
    counts and Delta; ssr and iterates to 1e-12."""
    m, n, per_col = (300000, 2000, 600) if big else (20000, 200, 100)
    runs = []
    for env in ({}, {"LSQ_LSMR_SEPARATE_SETUP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=9, ctx=ctx)
        pr.reset()
        r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, trace=True, iterations=12)
        for k in env:
            monkeypatch.delenv(k)
        runs.append(r)
        pr.close()
    a, b = runs
    assert a.iterations == b.iterations > 3 and a.mul_calls == b.mul_calls and a.f_calls == b.f_calls
    assert np.array_equal(a.trace["inner"], b.trace["inner"]) and np.array_equal(a.trace["accept"], b.trace["accept"])
    assert np.array_equal(a.trace["delta"], b.trace["delta"])
    assert np.allclose(a.trace["ssr"], b.trace["ssr"], rtol=1e-12, atol=0)
    assert np.allclose(a.trace["gnorm"], b.trace["gnorm"], rtol=1e-12, atol=0)
    assert np.max(np.abs(np.array(a.trace["x"]) - np.array(b.trace["x"]))) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,per_col", [(300000, 2000, 600), (1_000_000, 10_000, 1000)], ids=["300000x2000", "C4"])
def test_lm_pair_tail_matches_the_two_passes(ctx, m, n, per_col, monkeypatch):
    """Round 4: the predicted residual |J dx - f|^2 (levenberg_marquardt.jl:114-117) and f!(x_trial) + sum(abs2, .) (:107,
    :111) of the built-in model as ONE pass over A (k_sell_rows_pair: every value/index pair feeds both gather vectors)
    against the two launches it replaces (LSQ_NO_PAIR_TAIL=1).  Every row's two sums are the same left-to-right sums, so the
    trial residual is identical bit for bit; the two sums of squares are associated differently (per lane in slice order
    instead of per row in window order): identical counts, accept pattern, inner counts; ssr / iterates to 1e-12."""
    runs = []
    for env in ({}, {"LSQ_NO_PAIR_TAIL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=13, ctx=ctx)
        pr.reset()
        r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, trace=True, iterations=8, x_tol=0, f_tol=0, g_tol=0)
        fc = pr.fcur.get()
        for k in env:
            monkeypatch.delenv(k)
        runs.append((r, fc))
        pr.close()
    (a, fa), (b, fb) = runs
    assert a.iterations == b.iterations == 8 and a.mul_calls == b.mul_calls and a.f_calls == b.f_calls
    assert np.array_equal(a.trace["inner"], b.trace["inner"]) and np.array_equal(a.trace["accept"], b.trace["accept"])
    assert np.allclose(a.trace["ssr"], b.trace["ssr"], rtol=1e-12, atol=0)
    assert np.max(np.abs(np.array(a.trace["x"]) - np.array(b.trace["x"]))) <= 1e-12
    assert np.max(np.abs(fa - fb)) <= 1e-12 * max(1.0, np.max(np.abs(fb)))


@pytest.mark.gpu
@pytest.mark.parametrize("delta,noise", [(None, 1e-3), (1e6, 0.5), (1e-3, 0.5)], ids=["default", "rejections", "small_radius"])
def test_lm_speculative_gradient_pass_changes_nothing(ctx, delta, noise, monkeypatch):
    """Round 4: the next Jacobian's gradient + colsumabs2 pass is queued behind the tail of the current iteration, guarded by
    the acceptance test taken on the device (k_sell_rows_pair writes the skip word), and adopted by the host when its own test
    agrees (levenberg_marquardt.jl:118-122).  Same kernels on the same data in either order: the run must be BIT-IDENTICAL to
    the run without speculation (LSQ_NO_SPEC_GRADIENT=1) -- also through rejected steps (a huge initial radius on a noisy
    problem: the Gauss-Newton-like first steps are refused) and the handle's colsumabs2 must be the current one afterwards."""
    m, n, per_col = 300000, 2000, 600
    runs = []
    for env in ({}, {"LSQ_NO_SPEC_GRADIENT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=17, ctx=ctx)
        if noise != 1e-3:      # a noisier right-hand side: larger residual, more curvature, steps get refused
            rng = np.random.default_rng(3)
            pr.close()
            cp, rv, nz = lsq.synthetic.sparse_inputs(m, n, per_col, 17)
            xt_true = lsq.synthetic.uniform(n, 17 + 101, -3.0, 3.0)
            bvec = lsq.synthetic.csc_matvec(m, cp, rv, nz, np.tanh(xt_true)) + noise * rng.standard_normal(m)
            pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=17, ctx=ctx, inputs=(cp, rv, nz), b=bvec)
        pr.reset()
        r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, trace=True, iterations=14, delta=delta, x_tol=0, f_tol=0, g_tol=0)

        class _J:
            h = pr.J
            ctx_ = ctx
        cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), _J).get()
        for k in env:
            monkeypatch.delenv(k)
        runs.append((r, cs, pr.fcur.get()))
        pr.close()
    (a, csa, fa), (b, csb, fb) = runs
    assert a.iterations == b.iterations == 14 and a.mul_calls == b.mul_calls and a.f_calls == b.f_calls and a.g_calls == b.g_calls
    assert np.array_equal(a.trace["accept"], b.trace["accept"]) and np.array_equal(a.trace["inner"], b.trace["inner"])
    if delta == 1e6:
        assert not np.all(a.trace["accept"]), "this case is meant to contain rejected steps"
    assert np.array_equal(a.trace["ssr"], b.trace["ssr"]) and np.array_equal(a.trace["gnorm"], b.trace["gnorm"])
    assert np.array_equal(np.array(a.trace["x"]), np.array(b.trace["x"]))
    assert np.array_equal(a.minimizer, b.minimizer) and np.array_equal(fa, fb)
    assert np.array_equal(csa, csb)          # the handle's colsumabs2 cache: current, not a speculative leftover


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["lm", "dogleg"])
def test_tanh_model_column_scaled_vs_multiplied_out(ctx, opt, monkeypatch):
    """The built-in model keeps J = A diag(1 - tanh(x)^2) as a COLUMN-SCALED handle on big sparse patterns (nothing is
    multiplied out, g! writes n factors).  With LSQ_NO_COLSCALE=1 the same model multiplies J out into both sliced copies
    after every accepted step, as rounds 1-2 did.  Same algorithm, entries used as A_ij*s_j on the fly instead of the
    stored fl(A_ij*s_j): identical iteration counts, accept pattern and inner counts; iterates to 1e-10."""
    m, n, per_col = 300000, 2000, 600
    okind = lsq._lib.LEVENBERG_MARQUARDT if opt == "lm" else lsq._lib.DOGLEG
    runs = []
    for env in ({}, {"LSQ_NO_COLSCALE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=7, ctx=ctx)
        for k in env:
            monkeypatch.delenv(k)
        pr.reset()
        r = pr.optimize(okind, lsq._lib.LSMR, trace=True, iterations=12)
        runs.append((r.iterations, r.ssr, np.array(r.trace["x"]), np.array(r.trace["inner"]), np.array(r.trace["accept"]),
                     r.mul_calls))
        pr.close()
    a, b = runs
    assert a[0] > 3 and a[0] == b[0] and a[5] == b[5]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert a[1] == pytest.approx(b[1], rel=1e-11)
    assert np.max(np.abs(a[2] - b[2])) <= 1e-10 * max(1.0, np.max(np.abs(b[2])))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_device_g_values_stay_consistent(ctx, fused, monkeypatch):
    """After the built-in device g!: column-scaled handle (default on big patterns) -- the stored values read back as A and
    every operation acts on A diag(s); multiplied-out mode (LSQ_NO_COLSCALE=1) -- only the mirrors the products read were
    written and the CSC-ordered nzval is rebuilt on demand.  Either way colsumabs2 / J'u / J v / rowsumabs2 taken afterwards
    must agree with J(x) = A diag(1 - tanh(x)^2)."""
    m, n, per_col = 300000, 2000, 600
    if not fused:
        monkeypatch.setenv("LSQ_NO_COLSCALE", "1")
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=5, ctx=ctx)
    monkeypatch.delenv("LSQ_NO_COLSCALE", raising=False)
    x0 = lsq.synthetic.uniform(n, 3)
    pr.reset(x0)
    L = lsq.lib()
    assert L.lsq_model_g()(pr.J, pr.x.ptr, pr.model) == 0
    sfac = 1.0 - np.tanh(x0) ** 2
    cols = np.repeat(np.arange(n), np.diff(pr.colptr))
    want = pr.A * sfac[cols]
    got = np.empty_like(pr.A)
    lsq._lib.check(L.lsq_mat_get_values(pr.J, got.ctypes.data_as(lsq._lib.c_dp)))
    if fused:
        assert np.array_equal(got, pr.A)
    else:
        np.testing.assert_allclose(got, want, rtol=4e-16 * 8, atol=0)
    cs = lsq.DeviceVector(ctx, n)
    lsq._lib.check(L.lsq_colsumabs2(pr.J, cs.ptr))
    np.testing.assert_allclose(cs.get(), np.add.reduceat(want * want, pr.colptr[:-1]), rtol=1e-12)
    u = lsq.DeviceVector(ctx, m, lsq.synthetic.normal(m, 9))
    g = lsq.DeviceVector(ctx, n)
    lsq._lib.check(L.lsq_mul(pr.J, 1, 1.0, u.ptr, 0.0, g.ptr))
    ref = np.add.reduceat(want * u.get()[pr.rowval], pr.colptr[:-1])
    np.testing.assert_allclose(g.get(), ref, rtol=1e-10, atol=1e-10 * np.max(np.abs(ref)))
    S = sp.csc_matrix((want, pr.rowval, pr.colptr), shape=(m, n))
    v = lsq.synthetic.normal(n, 4)
    out = lsq.DeviceVector(ctx, m)
    lsq._lib.check(L.lsq_mul(pr.J, 0, 1.0, lsq.DeviceVector(ctx, n, v).ptr, 0.0, out.ptr))
    np.testing.assert_allclose(out.get(), S @ v, rtol=0, atol=1e-12 * (1 + np.max(np.abs(S @ v))))
    rs = lsq.DeviceVector(ctx, m)
    lsq._lib.check(L.lsq_rowsumabs2(pr.J, rs.ptr))
    np.testing.assert_allclose(rs.get(), np.asarray(S.multiply(S).sum(axis=1)).ravel(), rtol=1e-12)
    pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sliced", "wide", "segments", "small", "dense"])
def test_column_scaled_jacobian(ctx, kind, monkeypatch):
    """lsq_mat_set_colscale (include/lsqhip.h): a handle holding V with n factors s acts as J = V diag(s) in every operation of
    the hot path -- products, colsumabs2, rowsumabs2, the damped LSMR solve -- on the sliced layouts (fused, nothing multiplied
    out), on segment-kernel patterns, on small matrices (reference-order kernels) and on dense ones (values multiplied out
    behind the handle).  Checked against the oracle on the multiplied-out matrix; then s changes (colscale_changed), then V
    changes (set_values), then the scale is removed."""
    rng = np.random.default_rng(11)
    if kind in ("sliced", "segments", "wide"):
        m, n = 200000, 1500
        S = rand_csc(m, n, 0.004, 5)
        if kind == "segments":
            monkeypatch.setenv("LSQ_NO_SELL", "1")
        if kind == "wide":     # x passes through LDS in four column windows (what n > 12160 gets)
            monkeypatch.setenv("LSQ_SELL_XMAX", "400")
    elif kind == "small":
        m, n = 300, 20
        S = rand_csc(m, n, 0.3, 6)
    else:
        m, n = 900, 40
        S = rng.standard_normal((m, n))
    J = lsq.DeviceMatrix(ctx, S)
    monkeypatch.delenv("LSQ_NO_SELL", raising=False)
    monkeypatch.delenv("LSQ_SELL_XMAX", raising=False)
    V = S.tocsc() if kind != "dense" else S
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    damp = rng.uniform(0.5, 2.0, n)

    def check_all(Vm, s):
        Jm = (Vm @ sp.diags(s)).tocsc() if kind != "dense" else Vm * s[None, :]
        A = O.Mat.from_scipy(Jm) if kind != "dense" else O.Mat(dense=Jm)
        scale = 1 + (np.abs(Jm).sum(axis=1).max() if kind != "dense" else np.abs(Jm).sum(axis=1).max())
        out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, lsq.DeviceVector(ctx, n, x), 1.5, -0.5).get()
        assert np.max(np.abs(out - O.mul(A, x, 1.5, -0.5, y))) <= 1e-12 * scale
        scale_t = 1 + np.abs(Jm).sum(axis=0).max()
        out = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, lsq.DeviceVector(ctx, m, y), -2.0, 0.25, trans=True).get()
        assert np.max(np.abs(out - O.mulT(A, y, -2.0, 0.25, x))) <= 1e-12 * scale_t
        assert np.allclose(lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get(), O.colsumabs2(A), rtol=1e-13, atol=0)
        assert np.allclose(lsq.rowsumabs2_(lsq.DeviceVector(ctx, m), J).get(), O.rowsumabs2(A), rtol=1e-13, atol=1e-300)
        sv = lsq.AllocatedSolver(J, lsq.LSMR(), for_lm=True)
        xg, nmul = sv.ldiv_(lsq.DeviceVector(ctx, n), lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        st, xo, nmo, _ = O.ldiv(O.LSMR, A, y, damp)
        assert st == O.OK and nmul == nmo
        assert np.max(np.abs(xg.get() - xo)) <= 1e-8 * max(1.0, np.max(np.abs(xo)))
        sv.free()

    s1 = rng.uniform(0.2, 1.5, n)
    ds = lsq.DeviceVector(ctx, n, s1)
    J.set_colscale(ds)
    vals0 = J.values()
    assert np.array_equal(vals0, V.data if kind != "dense" else np.asfortranarray(V).reshape(-1, order="F"))   # still V
    check_all(V, s1)
    s2 = rng.uniform(0.1, 3.0, n)
    ds.set(s2)
    J.colscale_changed()
    check_all(V, s2)
    if kind != "dense":
        V2 = V.copy()
        V2.data = rng.standard_normal(V2.nnz)
        J.set_values(V2.data)
    else:
        V2 = rng.standard_normal((m, n))
        J.set_values(np.asfortranarray(V2).reshape(-1, order="F"))
    check_all(V2, s2)
    J.set_colscale(None)
    check_all(V2, np.ones(n))
    J.free()


@pytest.mark.gpu
@pytest.mark.parametrize("opt,sol,sparse", [("lm", "lsmr", True), ("dogleg", "lsmr", True), ("lm", "cholesky", False),
                                            ("dogleg", "qr", False)])
def test_allocated_workspace_reuse_is_stateless(ctx, opt, sol, sparse):
    """types.jl:141-160: an allocated problem may be optimised repeatedly.  The library keeps the
    optimizer/solver buffers of the last (J, optimizer, solver) in the context; a second solve from
    the same start must be bit-identical to the first, and a different problem in between must
    invalidate the cache."""
    m, n, per_col = (20000, 200, 100) if sparse else (1500, 48, None)
    okind = lsq._lib.LEVENBERG_MARQUARDT if opt == "lm" else lsq._lib.DOGLEG
    skind = {"lsmr": lsq._lib.LSMR, "cholesky": lsq._lib.CHOLESKY, "qr": lsq._lib.QR}[sol]
    pr = lsq.synthetic.TanhProblem(m, n, sparse=sparse, per_col=per_col, seed=11, ctx=ctx)
    other = lsq.synthetic.TanhProblem(m // 2, n, sparse=sparse, per_col=(per_col // 2 if sparse else None), seed=12, ctx=ctx)
    runs = []
    for k in range(3):
        pr.reset()
        r = pr.optimize(okind, skind, trace=True, iterations=30)
        runs.append(r)
        if k == 1:                      # evict the cached workspace
            other.reset()
            other.optimize(okind, skind, iterations=3)
    for r in runs[1:]:
        assert r.iterations == runs[0].iterations and r.mul_calls == runs[0].mul_calls
        assert r.ssr == runs[0].ssr and np.array_equal(r.minimizer, runs[0].minimizer)
        assert np.array_equal(r.trace["inner"], runs[0].trace["inner"])
    other.close()
    pr.close()


@pytest.mark.parametrize("opt,sol,sparse", GRID)
def test_minpack_trajectories(opt, sol, sparse):
    """test/nonlinearsolvers.jl:505-537 on the device (reference-order kernels), trajectory-checked
    against the oracle: identical counts on all 21 instances, for every solver/optimizer pair."""
    lsq.set_exact(True)
    for p in P.minpack_all():
        rg = gpu_run(p, OPT[opt][0], SOL[sol][0](), sparse)
        ro = oracle_run(p, OPT[opt][1], SOL[sol][1], sparse)
        compare(rg, ro, (P.label(p), opt, sol, sparse), xtol=1e-12 if sol == "lsmr" else 1e-5)
    lsq.set_exact(None)


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
def test_minpack_cholesky_trajectories(opt):
    """test/nonlinearsolvers.jl:573-595"""
    lsq.set_exact(True)
    for p in P.minpack_cholesky():
        rg = gpu_run(p, OPT[opt][0], lsq.Cholesky())
        ro = oracle_run(p, OPT[opt][1], O.CHOLESKY)
        assert rg.converged
        compare(rg, ro, (P.label(p), opt, "cholesky"), xtol=1e-5)
    lsq.set_exact(None)


def _count_stable():
    """tests/golden/count_stable.json: runs of the grid whose counts do not depend on the summation order of the
    stdlib reductions (oracle under orc_set_sum_mode 0..5, tests/golden/make_count_stable.py)."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "count_stable.json")) as fh:
        cs = json.load(fh)
    return {(r["problem"], r["optimizer"], r["solver"], r["sparse"]): r["robust"] for r in cs["runs"]}


@pytest.mark.parametrize("opt,sol,sparse", GRID + [("dogleg", "cholesky", False), ("lm", "cholesky", False)])
def test_minpack_fast_kernels(opt, sol, sparse):
    """The SAME grid through the fast kernels (tree reductions, fused epilogues, launch-per-phase
    LSMR with the host mailbox) that large problems use: the reference's outcome pins everywhere; identical counts
    and 1e-5 iterates on the ROBUST set of tests/golden/count_stable.json -- the runs whose counts the oracle keeps
    under every modelled summation order (the stdlib's plausible ones, wave trees, random orders) and under last-bit
    perturbations of every reduction.  On the other runs (LSMR far past the loss of orthogonality on ill-conditioned
    Jacobians) any such change, the fast kernels' included, moves the stop iteration by a few counts: there the drift from
    the oracle is BOUNDED instead (iterations, mul_calls, minimiser)."""
    stable = _count_stable()
    lsq.set_exact(False)
    try:
        probs = P.minpack_cholesky() if sol == "cholesky" else P.minpack_all()
        for p in probs:
            rg = gpu_run(p, OPT[opt][0], SOL[sol][0](), sparse)
            assert rg.ssr <= 1e-3, (P.label(p), rg.ssr)          # test/nonlinearsolvers.jl:532
            if sol == "cholesky":
                assert rg.converged                              # :592
            ro = oracle_run(p, OPT[opt][1], SOL[sol][1], sparse)
            if stable[(P.label(p), opt, sol, sparse)]:
                # iterates: 1e-5 for the direct solvers; 1e-4 for LSMR, whose inner solves are themselves only
                # accurate to atol = btol = 1e-6 on operators with cond ~ 1e6+ (wood(4): 1.3e-5 mid-trajectory)
                compare(rg, ro, (P.label(p), opt, sol, sparse), xtol=1e-4 if sol == "lsmr" else 1e-5)
            else:
                # NOT robust: the oracle's own counts move under reordered sums here, so equality is not the claim -- but the
                # drift is bounded (measured with tools/fast_vs_oracle_nonrobust.py: iteration counts within
                # 0-2 except watson(6) +7 % and watson(9) -20 % under Dogleg+LSMR, minimisers within 2.5e-3 in watson(9)'s
                # flat valley, 5e-5 elsewhere): a regression in the fused LSMR kernels could not hide in these runs
                key = (P.label(p), opt, sol, sparse)
                assert rg.converged == ro.converged, key
                assert abs(rg.iterations - ro.iterations) <= max(3, ro.iterations // 4), (key, rg.iterations, ro.iterations)
                assert abs(rg.mul_calls - ro.mul_calls) <= max(12, ro.mul_calls // 3), (key, rg.mul_calls, ro.mul_calls)
                scale = max(1.0, float(np.max(np.abs(ro.minimizer))))
                assert np.max(np.abs(rg.minimizer - ro.minimizer)) <= (5e-3 if "watson(9)" in key[0] else 2e-4) * scale, key
    finally:
        lsq.set_exact(None)


def test_golden_fixtures():
    """The HIP path against the committed golden vectors (tests/golden/minpack_oracle.json, 162 runs
    of the reference's MINPACK grid): identical iteration / f / g / mul counts and minimisers."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "minpack_oracle.json")) as fh:
        gold = json.load(fh)
    probs = {P.label(p): p for p in P.minpack_all()}
    lsq.set_exact(True)
    for rec in gold["runs"]:
        r = gpu_run(probs[rec["problem"]], OPT[rec["optimizer"]][0], SOL[rec["solver"]][0](), rec["sparse"])
        key = (rec["problem"], rec["optimizer"], rec["solver"], rec["sparse"])
        assert r.iterations == rec["iterations"], key
        assert (r.f_calls, r.g_calls, r.mul_calls) == (rec["f_calls"], rec["g_calls"], rec["mul_calls"]), key
        assert r.converged == rec["converged"], key
        assert np.allclose(r.minimizer, rec["x"], rtol=1e-6, atol=1e-8), key
    lsq.set_exact(None)


@pytest.mark.parametrize("opt,sol,sparse", GRID)
def test_operator_level_loops(opt, sol, sparse):
    """The reference's loops restated over the OPERATOR-level ABI only (loops.py: what the Julia shim
    runs) against the fused loop-level entry point: identical counts, equal minimisers."""
    lsq.set_exact(True)
    for p in P.minpack_all()[:12] + P.minpack_all()[14:]:
        name, f, g, x0 = p
        n = len(x0)
        def mk():
            if sparse:
                m_, n_, colptr, rowval = P.full_csc_pattern(n, n)
                J = sp.csc_matrix((np.zeros(n * n), rowval, colptr), shape=(n, n))
                g_ = lambda Jm, x: g(Jm.data.reshape((n, n), order="F"), x)
            else:
                J, g_ = np.zeros((n, n), order="F"), g
            return lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(n), f_=f, g_=g_, J=J)
        r1 = lsq.optimize_(mk(), OPT[opt][0](SOL[sol][0]()))
        r2 = lsq.optimize_operator_level(mk(), OPT[opt][0](SOL[sol][0]()))
        key = (P.label(p), opt, sol, sparse)
        assert (r1.iterations, r1.f_calls, r1.g_calls, r1.mul_calls) == (r2.iterations, r2.f_calls, r2.g_calls, r2.mul_calls), key
        assert (r1.converged, r1.x_converged, r1.f_converged, r1.g_converged) == \
               (r2.converged, r2.x_converged, r2.f_converged, r2.g_converged), key
        assert np.allclose(r1.minimizer, r2.minimizer, rtol=1e-10, atol=1e-12), key
    lsq.set_exact(None)


# Round 6: every assertion the REFERENCE'S OWN tests hold -- the known-answer trajectories, the factor model's
# outcome pins (test/nonlinearleastsquares.jl:107-108), the bound problems (test/bounds.jl:12-14,23-26,33-36), the README examples
# (test/runtests.jl:19-46) -- through BOTH code paths: kernels = "reference-order" is what these small problems take by default
# (lsq_exact.hip: the reference's summation order, identical counts against the oracle); kernels = "fast" forces the kernels the
# headline runs (tree reductions, fused epilogues, device-resident LSMR, blocked dense solvers: lsq.set_exact(False)), which are
# held to the reference-held assertions themselves -- counts against the oracle only where the problem is count-stable.
KERNELS = ["reference-order", "fast"]


@pytest.fixture(params=KERNELS)
def kernels(request):
    lsq.set_exact(None if request.param == "reference-order" else False)
    yield request.param
    lsq.set_exact(None)


def test_kat_trajectories(kernels):
    """SURVEY 8c KAT-DL / KAT-LM through the HIP path (hand-derived from the reference's formulas: exact in any summation order)."""
    r = gpu_run(P.readme_rosenbrock(), lsq.Dogleg, lsq.QR(), iterations=2)
    assert r.trace["rho"][0] == pytest.approx(-9999.0, rel=1e-12)
    assert r.trace["rho"][1] == pytest.approx(-624.25 / 0.75, rel=1e-12)
    assert list(r.trace["delta"]) == [0.5, 0.25] and np.all(r.trace["x"] == 0)
    r = gpu_run(P.readme_rosenbrock(), lsq.LevenbergMarquardt, lsq.QR(), iterations=1)
    assert r.trace["rho"][0] == pytest.approx(-6886.0523416, rel=1e-9) and r.trace["delta"][0] == 5.0


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
def test_factor_model(opt, kernels):
    """test/nonlinearleastsquares.jl:96-110 (rank-deficient J'J: pins the min-norm QR solve)."""
    name, f, g, x0 = P.factor_dense()
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.ones(9), f_=f, g_=g, J=np.ones((9, 6)))
    r = lsq.optimize_(nls, OPT[opt][0](lsq.QR()), full_trace=True)
    ff, gg = P.wrap_dense(f, g, 9, 6)
    ro = O.optimize(OPT[opt][1], O.QR, O.Mat(dense=np.zeros((9, 6))), x0, ff, gg)
    assert r.converged and r.ssr <= 12                                  # the reference's own pin (:107)
    if kernels == "reference-order":
        assert r.iterations == ro.iterations and np.allclose(r.minimizer, ro.minimizer, rtol=1e-6, atol=1e-8)
    else:   # (a rank-deficient problem: the minimiser is not unique, the objective is)
        assert abs(r.ssr - ro.ssr) <= 1e-8 * max(1.0, ro.ssr)
    name, f, gs, x0, (m, n, colptr, rowval) = P.factor_sparse()
    J = sp.csc_matrix((np.ones(18), rowval, colptr), shape=(9, 6))
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.ones(9), f_=f, g_=lambda Jm, x: gs(Jm.data, x), J=J)
    r = lsq.optimize_(nls, OPT[opt][0](lsq.LSMR()), full_trace=True)
    ro = O.optimize(OPT[opt][1], O.LSMR, O.Mat(csc=(m, n, colptr, rowval, np.zeros(18))), x0, f, gs)
    assert r.converged and r.ssr <= 12                                  # (:108)
    if kernels == "reference-order":
        assert r.iterations == ro.iterations and r.mul_calls == ro.mul_calls
    else:
        assert abs(r.ssr - ro.ssr) <= 1e-8 * max(1.0, ro.ssr)


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
def test_bounds(opt, kernels):
    """test/bounds.jl:7-38"""
    mk = OPT[opt][0]

    def go(p, **kw):
        name, f, g, x0 = p
        nls = lsq.LeastSquaresProblem(x=x0.copy(), f_=f, g_=g, output_length=2)
        return lsq.optimize_(nls, mk(), **kw)

    r = go(P.readme_rosenbrock(), lower=[0.0, 0.0])
    assert r.converged and np.all(r.minimizer >= -1e-8) and np.linalg.norm(r.minimizer - [1, 1]) <= 1e-6
    r = go(P.bound_lower_active(), lower=[1.0, -100.0], x_tol=1e-50, f_tol=1e-50)
    assert r.converged and r.g_converged and np.linalg.norm(r.minimizer - [1, 3]) <= 1e-6
    r = go(P.bound_upper_active(), upper=[2.0, 100.0], x_tol=1e-50, f_tol=1e-50)
    assert r.converged and r.g_converged and np.linalg.norm(r.minimizer - [2, 2]) <= 1e-6
    with pytest.raises(lsq.ArgumentError):
        go(P.readme_rosenbrock(), lower=[1.0, 1.0])


def test_finite_difference_and_defaults(kernels):
    """test/runtests.jl:19-70 + test/nonlinearsolvers.jl:619-628"""
    rosen = lambda x: np.array([1 - x[0], 100 * (x[1] - x[0] ** 2)])
    for o in (lsq.Dogleg(), lsq.LevenbergMarquardt()):
        r = lsq.optimize(rosen, np.zeros(2), o)
        assert r.converged and r.ssr <= 1e-8
    r = lsq.optimize(lambda x: np.sum(x ** 2), np.array([1.0, 1.0]), lsq.Dogleg())  # issue #41
    assert r.converged
    name, f, g, x0 = P.wood()
    r = lsq.optimize_(lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(4), f_=f, g_=g, J=np.ones((4, 4))))
    assert r.optimizer == "Dogleg"
    Js = sp.csc_matrix(np.ones((4, 4)))
    r = lsq.optimize_(lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(4), f_=f,
                                              g_=lambda Jm, x: g(Jm.data.reshape((4, 4), order="F"), x), J=Js))
    assert r.optimizer == "LevenbergMarquardt"
    r = lsq.optimize(rosen, np.zeros(2), lsq.LevenbergMarquardt(), store_trace=True)
    assert len(r.tr) >= 1 and isinstance(r.tr[0], lsq.OptimizationState)
    # output_length defaults to size(J, 1) (runtests.jl:54-61)
    over = lambda o, x: o.__setitem__(slice(None), [x[0] - 1, x[1] - 2, x[2] - 3, x[0] + x[1], x[1] + x[2]])
    p = lsq.LeastSquaresProblem(x=np.zeros(3), f_=over, J=np.zeros((5, 3)))
    assert len(p.y) == 5 and lsq.optimize_(p, lsq.Dogleg()).converged


def test_nonfinite_raises():
    name, f, g, x0 = P.readme_rosenbrock()
    nls = lsq.LeastSquaresProblem(x=np.array([np.nan, 0.0]), f_=f, g_=g, output_length=2)
    with pytest.raises(lsq.IsFiniteException) as e:
        lsq.optimize_(nls, lsq.LevenbergMarquardt())
    assert e.value.indices == [0]


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
@pytest.mark.parametrize("sol", ["qr", "lsmr"])
@pytest.mark.parametrize("bounded", [False, True])
def test_nan_in_jacobian_at_a_later_iteration_raises(opt, sol, bounded):
    """check_isfinite(x) on the REJECTED-step path (utils.jl:70-75, levenberg_marquardt.jl:135, dogleg.jl:189): a
    Jacobian that turns NaN at its 2nd evaluation gives a NaN step, the step is rejected (rho = NaN), the restored
    x = (x - dx) + dx is NaN, and the next iteration throws IsFiniteException -- the same index as the oracle.  With
    bounds the NaN step must survive the box clipping (Julia's min / max propagate NaN; fmin / fmax would not)."""
    name, f, g0, x0 = P.wood()
    calls = {"n": 0}

    def g(Jm, x):
        g0(Jm, x)
        calls["n"] += 1
        if calls["n"] >= 2:
            Jm[1, 2] = np.nan

    kw = dict(lower=[-10.0] * 4, upper=[10.0] * 4) if bounded else {}
    J0 = np.zeros((4, 4))
    if sol == "lsmr":
        J = sp.csc_matrix(np.ones((4, 4)))
        gg = lambda Jm, x: (g(J0, x), Jm.data.__setitem__(slice(None), J0.reshape(-1, order="F")))
    else:
        J, gg = J0.copy(), g
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(4), f_=f, g_=gg, J=J)
    with pytest.raises(lsq.IsFiniteException) as e:
        lsq.optimize_(nls, OPT[opt][0](SOL[sol][0]()), iterations=50, **kw)
    # the oracle on the same problem
    calls["n"] = 0
    Jo = O.Mat(dense=np.zeros((4, 4))) if sol == "qr" else O.Mat(csc=P.full_csc_pattern(4, 4) + (np.zeros(16),))
    go = lambda Jv, x: (g(J0, x), Jv.__setitem__(slice(None), J0.reshape(-1, order="F")))
    ro = O.optimize(OPT[opt][1], SOL[sol][1], Jo, x0, f, go, iterations=50,
                    lower=kw.get("lower"), upper=kw.get("upper"))
    assert ro.status == O.ENONFINITE
    assert e.value.indices == [ro.bad_index]


def test_host_side_g_with_pinned_async_upload(ctx):
    """a HOST-side g! (numpy writes nonzeros(J), as the reference's sparse g! does,
    test/nonlinearleastsquares.jl:47-86) -- the values go up through page-locked memory with lsq_mat_set_values_async
    after every accepted step.  Same problem as the device-side model: same iteration / call counts, same iterates."""
    m, n, pc = 300000, 2000, 600
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=7, ctx=ctx)
    pr.reset()
    rd = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=12)
    A, colptr, rowval, b = pr.A, pr.colptr, pr.rowval, pr.b
    pr.close()
    S = sp.csc_matrix((A, rowval, colptr), shape=(m, n))
    cols = np.repeat(np.arange(n), np.diff(colptr))

    def f_(out, x):
        out[:] = S @ np.tanh(x) - b

    def g_(J, x):
        np.multiply(A, (1.0 - np.tanh(x) ** 2)[cols], out=J.data)

    J = sp.csc_matrix((np.zeros_like(A), rowval, colptr), shape=(m, n))
    data_before = J.data
    nls = lsq.LeastSquaresProblem(x=np.zeros(n), y=np.zeros(m), f_=f_, g_=g_, J=J)
    rh = lsq.optimize_(nls, lsq.LevenbergMarquardt(lsq.LSMR()), iterations=12, ctx=ctx)
    assert (rh.iterations, rh.f_calls, rh.g_calls, rh.mul_calls) == (rd.iterations, rd.f_calls, rd.g_calls, rd.mul_calls)
    assert rh.converged == rd.converged and rh.ssr == pytest.approx(rd.ssr, rel=1e-9)
    assert np.max(np.abs(rh.minimizer - rd.minimizer)) <= 1e-8
    # the Jacobian handed back lives in ordinary memory again and holds g!(x_last accepted)
    assert nls.J.data is data_before and np.all(np.isfinite(nls.J.data)) and np.any(nls.J.data != 0.0)
    # the pieces on their own: pinned upload == blocking upload
    Jd = lsq.DeviceMatrix(ctx, S)
    pin = lsq.PinnedBuffer(ctx, len(A))
    pin.array[:] = 2.0 * A
    Jd.set_values_async(pin)
    Jd.upload_wait()
    assert np.array_equal(Jd.values(), 2.0 * A)
    xv = lsq.DeviceVector(ctx, n, np.ones(n))
    out = lsq.mul_(lsq.DeviceVector(ctx, m), Jd, xv, 1.0, 0.0).get()
    assert np.max(np.abs(out - 2.0 * (S @ np.ones(n)))) <= 1e-10
    pin.free()
    Jd.free()


# --------------------------------------------------------------- reference-held vectors: NIST StRD
import nist_cases as NC  # noqa: E402

NIST_KEYS = [NC.config_key(o, s_, st, jac) for (o, s_, st) in NC.CONFIGS for jac in ("central", "analytic")]


@pytest.mark.parametrize("exact", [True, False], ids=["reference_order", "fast_kernels"])
@pytest.mark.parametrize("key", NIST_KEYS)
def test_nist_certified_values(key, exact):
    """test/nonlinearfitting.jl:1457-1470 through the HIP path -- and through ALL of it: both optimizers x {QR, Cholesky, LSMR
    on the dense J, LSMR on the same J as a fixed-pattern CSC} from every column of `parameters`, the reference's tolerances,
    its default central-difference Jacobian and the analytic one, with the reference-order kernels (lsq_exact.hip) and with
    the fast kernels.  Every run is classified with its evidence by the SAME code that classifies the oracle's runs
    (tests/nist_cases.py: hit / slow_in_basin / plateau / stationary / stalled_far / rank_deficient; the reference's own
    assert, no NaN, is part of it) and the class must be the one tests/golden/nist_outcomes.json records for the oracle.
    Where the oracle's own class is decided by round-off (`order_dependent`: it changes under the oracle's summation-order
    and rounding-noise modes) any of the classes seen there is accepted -- three knife-edge starts; LSMR with the
    reference-order kernels repeats the oracle's arithmetic bit for bit and gets no such allowance."""
    import nist
    opt, solver, storage, jac = key.split("/")
    fx = NC.load_outcomes()
    run = NC.hip_runner(lsq)
    lsq.set_exact(exact)
    try:
        got = {}
        for p in nist.problems():
            for si in range(len(p.starts)):
                got["%s/%d" % (p.name, si)] = NC.classify(p, si, opt, solver, storage, jac, run, fx["plateaus"])[0]
    finally:
        lsq.set_exact(None)
    assert len(got) == 33
    loose = {} if (exact and solver == "lsmr") else fx["order_dependent"].get(key, {})
    for start, v in fx["classes"][key].items():
        assert got[start] == v["class"] or got[start] in loose.get(start, ()), (key, start, got[start], v["class"])
    if solver == "qr":      # the reference's configuration: what its `println("strd ...")` would show
        assert sum(c == "hit" for c in got.values()) >= 31


# --------------------------------------------------------------- three-launch LSMR iteration (round 5)
@pytest.mark.parametrize("m,n,per_col,opt", [(300000, 2000, 600, "lm"), (300000, 2000, 600, "dogleg"), (1_000_000, 10_000, 1000, "lm")],
                         ids=["300000x2000-lm", "300000x2000-dogleg", "C4-lm"])
def test_lsmr_three_launch_iteration(ctx, m, n, per_col, opt, monkeypatch):
    """Round 5: K3 (k_lsmr_update) folded into the next J*v launch (k_lsmr_fused, lsq_lsmr3.h: a few workgroups at the front of
    the grid take the scalar chain, ||x|| and the stop decision of lsmr.jl:205-231 and the n-vector updates of :152-156 beside
    the product; the product workgroups gather the UNNORMALISED v~ and apply 1/alpha to the finished dot products) against the
    four-launch iteration it replaces (LSQ_LSMR_FOUR_LAUNCHES=1).  Same recurrence, other associations (sum(v~^2) over 157
    instead of 313 partials, (J w)/alpha instead of J (w/alpha), ||x||^2 by one workgroup): identical iteration counts, LSMR
    inner counts per outer iteration and accept pattern; ssr and iterates to 1e-11; run-to-run identical bits."""
    okind = lsq._lib.LEVENBERG_MARQUARDT if opt == "lm" else lsq._lib.DOGLEG
    runs = []
    for env in ({}, {}, {"LSQ_LSMR_FOUR_LAUNCHES": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=per_col, seed=23, ctx=ctx)
        pr.reset()
        # (6 iterations: the useful trajectory -- past convergence the accept decisions are round-off, see compare_until_roundoff)
        r = pr.optimize(okind, lsq._lib.LSMR, trace=True, iterations=6, x_tol=0, f_tol=0, g_tol=0)
        for k in env:
            monkeypatch.delenv(k)
        runs.append((r, pr.fcur.get()))
        pr.close()
    (a, fa), (b, fb), (c4, fc) = runs
    for other in (b, c4):
        assert a.iterations == other.iterations == 6 and a.mul_calls == other.mul_calls and a.f_calls == other.f_calls
        assert np.array_equal(a.trace["inner"], other.trace["inner"]), (a.trace["inner"], other.trace["inner"])
        assert np.array_equal(a.trace["accept"], other.trace["accept"])
    assert np.sum(a.trace["inner"]) // 2 >= 8          # (multi-iteration solves: the decisions were exercised)
    assert np.array_equal(np.array(a.trace["x"]), np.array(b.trace["x"])) and np.array_equal(a.trace["ssr"], b.trace["ssr"])
    assert np.array_equal(fa, fb)
    np.testing.assert_allclose(a.trace["ssr"], c4.trace["ssr"], rtol=1e-11, atol=0)
    assert np.max(np.abs(np.array(a.trace["x"]) - np.array(c4.trace["x"]))) <= 1e-11
