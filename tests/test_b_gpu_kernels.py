"""GPU parity, tier 2 -- kernels and single linear solves against the oracle / LAPACK (products on every layout, BLAS-1,
ldiv! of LSMR / Cholesky / QR on every launch path, certificates, preconditioners, operators, the serialised comparator).
Tolerances: tests/gpu_common.py.  Runs after tests/test_a_gpu_contract.py."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle import oracle as O
from gpu_common import GRID, OPT, SOL, compare, gpu_run, lsq, oracle_run, rand_csc

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("window_rows", [None, 1024, 96])
@pytest.mark.parametrize("plan", ["stream", "wave", "block", "ldswin", None])
@pytest.mark.parametrize("m,n,density", [(3000, 200, 0.01), (500, 40, 0.5), (64, 3000, 0.02)])
def test_sparse_products(ctx, plan, m, n, density, window_rows):
    """All three launch plans, for the CSR rows, the CSC columns and (window_rows: forced small so
    that small test matrices are cut into several windows) the row-window-blocked CSC of J'*y."""
    for k in ("LSQ_PLAN_CSC", "LSQ_PLAN_CSR", "LSQ_PLAN_BCSC"):
        if plan and not (plan == "ldswin" and k != "LSQ_PLAN_BCSC"):
            os.environ[k] = plan
        else:
            os.environ.pop(k, None)
    if window_rows:
        os.environ["LSQ_WINDOW_ROWS"] = str(window_rows)
    try:
        S = rand_csc(m, n, density, m + n)
        # ragged extremes: an empty column/row and one very long row
        S = S.tolil()
        S[5, :] = np.random.default_rng(0).standard_normal(n)
        S[:, 1] = 0
        S[3, :] = 0
        S = S.tocsc()
        S.sort_indices()
        S.eliminate_zeros()
        J = lsq.DeviceMatrix(ctx, S)
    finally:
        for k in ("LSQ_PLAN_CSC", "LSQ_PLAN_CSR", "LSQ_PLAN_BCSC", "LSQ_WINDOW_ROWS"):
            os.environ.pop(k, None)
    A = O.Mat.from_scipy(S)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    dx, dy = lsq.DeviceVector(ctx, n, x), lsq.DeviceVector(ctx, m, y)
    scale = 1 + np.abs(S).sum(axis=1).max()
    out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, dx, 1.5, -0.5).get()
    assert np.max(np.abs(out - O.mul(A, x, 1.5, -0.5, y))) <= 1e-12 * scale
    out = lsq.mul_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J, dx, 1.0, 0.0).get()
    assert np.max(np.abs(out - O.mul(A, x))) <= 1e-12 * scale  # beta == 0 overwrites NaN
    scale_t = 1 + np.abs(S).sum(axis=0).max()
    out = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, dy, -2.0, 0.25, trans=True).get()
    assert np.max(np.abs(out - O.mulT(A, y, -2.0, 0.25, x))) <= 1e-12 * scale_t
    cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get()
    assert np.allclose(cs, O.colsumabs2(A), rtol=1e-13, atol=0)
    assert cs[1] == 0.0


@pytest.mark.parametrize("rows,grows", [(None, None), (64, 64), (192, 256)])
@pytest.mark.parametrize("m,n,density", [(3000, 200, 0.01), (700, 40, 0.5), (64, 3000, 0.02), (9000, 6000, 0.001),
                                         (20000, 300, 0.004)])
def test_sliced_layouts(ctx, m, n, density, rows, grows):
    """The sliced (SELL) layouts of J*x and J'*y, forced onto small ragged matrices (empty rows and
    columns, one full row, several blocks / gather windows / column blocks; m = 20000 with 64-row blocks
    gives more blocks than CUs, i.e. workgroups that loop over several blocks) and compared with the
    oracle and, bit for bit, with a sequential evaluation of sampled rows."""
    S = rand_csc(m, n, density, m + 3 * n).tolil()
    S[5, :] = np.random.default_rng(0).standard_normal(n)
    S[:, 1] = 0
    S[3, :] = 0
    S = S.tocsc()
    S.sort_indices()
    S.eliminate_zeros()
    os.environ["LSQ_SELL_FORCE"] = "1"
    if rows:
        os.environ["LSQ_SELL_ROWS"], os.environ["LSQ_SELL_GROWS"] = str(rows), str(grows)
    try:
        J = lsq.DeviceMatrix(ctx, S)
    finally:
        for k in ("LSQ_SELL_FORCE", "LSQ_SELL_ROWS", "LSQ_SELL_GROWS"):
            os.environ.pop(k, None)
    lsq.set_exact(False)
    try:
        A = O.Mat.from_scipy(S)
        rng = np.random.default_rng(1)
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        dx, dy = lsq.DeviceVector(ctx, n, x), lsq.DeviceVector(ctx, m, y)
        scale = 1 + np.abs(S).sum(axis=1).max()
        out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, dx, 1.5, -0.5).get()
        assert np.max(np.abs(out - O.mul(A, x, 1.5, -0.5, y))) <= 1e-12 * scale
        # every output is ONE lane's left-to-right sum over the row's entries: the reference's order
        # (SparseArrays mul!), so sampled rows -- among them the full row 5 -- must match bit for bit
        Sr = S.tocsr()
        # (rows averaging >= 48 entries keep the wave-per-row plan, whose tree sums differ in the last bits)
        for i in ([5, 3] + list(rng.integers(0, m, 40))) if S.nnz / m < 48 else []:
            dot = 0.0
            for k in range(Sr.indptr[i], Sr.indptr[i + 1]):
                dot += Sr.data[k] * x[Sr.indices[k]]
            assert out[i] == 1.5 * dot + -0.5 * y[i]
        out = lsq.mul_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J, dx, 1.0, 0.0).get()
        assert np.max(np.abs(out - O.mul(A, x))) <= 1e-12 * scale
        scale_t = 1 + np.abs(S).sum(axis=0).max()
        out = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, dy, -2.0, 0.25, trans=True).get()
        assert np.max(np.abs(out - O.mulT(A, y, -2.0, 0.25, x))) <= 1e-12 * scale_t
        cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get()
        assert np.allclose(cs, O.colsumabs2(A), rtol=1e-13, atol=0) and cs[1] == 0.0
        # values written after creation reach the sliced copies, and read back unchanged
        J.set_values(2.0 * S.data)
        out = lsq.mul_(lsq.DeviceVector(ctx, m), J, dx, 1.0, 0.0).get()
        assert np.max(np.abs(out - 2.0 * O.mul(A, x))) <= 1e-12 * scale
        assert np.array_equal(J.values(), 2.0 * S.data)
    finally:
        lsq.set_exact(None)


@pytest.mark.parametrize("m,n,density,xmax,rows", [(3000, 200, 0.01, 64, None), (64, 3000, 0.02, 100, None),
                                                   (9000, 6000, 0.001, 1000, 128), (20000, 300, 0.004, 77, 64),
                                                   (5000, 400, 0.3, 128, 4096)])
def test_sliced_rows_column_windows(ctx, m, n, density, xmax, rows):
    """J*x of patterns wider than the LDS copy of x (n > 12160 in production; here the window is narrowed with
    LSQ_SELL_XMAX): the row block's workgroup walks the column windows and continues every row's running sum, so the rows
    must still match a sequential left-to-right evaluation BIT FOR BIT; rowsumabs2 likewise; column-scaled handles; values
    written after creation; rows without entries in some or all windows."""
    S = rand_csc(m, n, density, 2 * m + n).tolil()
    S[5, :] = np.random.default_rng(0).standard_normal(n)
    S[:, 1] = 0
    S[3, :] = 0
    S[7, : n // 2] = 0          # entries only in the upper windows
    S[8, n // 3:] = 0           # ... only in the lower ones
    S = S.tocsc()
    S.sort_indices()
    S.eliminate_zeros()
    os.environ["LSQ_SELL_FORCE"], os.environ["LSQ_SELL_XMAX"] = "1", str(xmax)
    if rows:
        os.environ["LSQ_SELL_ROWS"] = str(rows)
    try:
        J = lsq.DeviceMatrix(ctx, S)
    finally:
        for k in ("LSQ_SELL_FORCE", "LSQ_SELL_ROWS", "LSQ_SELL_XMAX"):
            os.environ.pop(k, None)
    lsq.set_exact(False)
    try:
        A = O.Mat.from_scipy(S)
        rng = np.random.default_rng(1)
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        dx = lsq.DeviceVector(ctx, n, x)
        scale = 1 + np.abs(S).sum(axis=1).max()
        Sr = S.tocsr()

        def seq_rows(vals, xx, idx):
            out = {}
            for i in idx:
                dot = 0.0
                for k in range(Sr.indptr[i], Sr.indptr[i + 1]):
                    dot += vals[k] * xx[Sr.indices[k]]
                out[i] = dot
            return out
        sample = [5, 3, 7, 8] + list(rng.integers(0, m, 60))
        out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, dx, 1.5, -0.5).get()
        assert np.max(np.abs(out - O.mul(A, x, 1.5, -0.5, y))) <= 1e-12 * scale
        if S.nnz / m < 48:
            for i, dot in seq_rows(Sr.data, x, sample).items():
                assert out[i] == 1.5 * dot + -0.5 * y[i], i
        out = lsq.mul_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J, dx, 1.0, 0.0).get()
        assert np.max(np.abs(out - O.mul(A, x))) <= 1e-12 * scale and out[3] == 0.0
        # the adjoint product and colsumabs2 of the same handle
        dy = lsq.DeviceVector(ctx, m, y)
        outt = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, dy, -2.0, 0.25, trans=True).get()
        assert np.max(np.abs(outt - O.mulT(A, y, -2.0, 0.25, x))) <= 1e-12 * (1 + np.abs(S).sum(axis=0).max())
        assert np.allclose(lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get(), O.colsumabs2(A), rtol=1e-13, atol=0)
        # rowsumabs2: left-to-right sums of squares across the windows
        rs = lsq.rowsumabs2_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J).get()
        ref = O.rowsumabs2(A)
        assert np.allclose(rs, ref, rtol=1e-13, atol=0) and rs[3] == 0.0
        if S.nnz / m < 48:
            assert np.array_equal(rs, ref)
        # values written after creation reach the sliced copy and read back unchanged
        J.set_values(2.0 * S.data)
        out = lsq.mul_(lsq.DeviceVector(ctx, m), J, dx, 1.0, 0.0).get()
        assert np.max(np.abs(out - 2.0 * O.mul(A, x))) <= 1e-12 * scale
        assert np.array_equal(J.values(), 2.0 * S.data)
    finally:
        lsq.set_exact(None)


def test_sliced_layouts_random_patterns(ctx):
    """Forty random shapes / densities / block sizes (every other one with x cut into column windows), with columns and
    rows emptied at random and duplicate-free ragged rows: J*x, J'*y, colsumabs2 and rowsumabs2 of the sliced layouts against scipy."""
    rng = np.random.default_rng(20260928)
    lsq.set_exact(False)
    try:
        for case in range(40):
            m = int(rng.integers(1, 4000))
            n = int(rng.integers(1, 600))
            density = float(rng.choice([0.0005, 0.005, 0.05, 0.3]))
            S = sp.random(m, n, density=density, format="lil", random_state=rng, data_rvs=rng.standard_normal)
            for _ in range(int(rng.integers(0, 4))):
                S[:, int(rng.integers(0, n))] = 0
                S[int(rng.integers(0, m)), :] = 0
            if rng.random() < 0.5:
                S[int(rng.integers(0, m)), :] = rng.standard_normal(n)     # one full row
            S = S.tocsc()
            S.sort_indices()
            S.eliminate_zeros()
            os.environ["LSQ_SELL_FORCE"] = "1"
            os.environ["LSQ_SELL_ROWS"] = str(int(rng.choice([64, 128, 4096])))
            os.environ["LSQ_SELL_GROWS"] = str(int(rng.choice([64, 512, 8192])))
            if case % 2:      # every other case: x in column windows narrower than n (k_sell_rows_wide)
                os.environ["LSQ_SELL_XMAX"] = str(int(rng.integers(64, 300)))
            try:
                J = lsq.DeviceMatrix(ctx, S)
            finally:
                for k in ("LSQ_SELL_FORCE", "LSQ_SELL_ROWS", "LSQ_SELL_GROWS", "LSQ_SELL_XMAX"):
                    os.environ.pop(k, None)
            x, y = rng.standard_normal(n), rng.standard_normal(m)
            dx, dy = lsq.DeviceVector(ctx, n, x), lsq.DeviceVector(ctx, m, y)
            out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, dx, 0.75, 1.25).get()
            ref = 0.75 * (S @ x) + 1.25 * y
            assert np.max(np.abs(out - ref)) <= 1e-12 * (1 + np.abs(S).sum(axis=1).max() + np.abs(y).max()), (case, m, n, density)
            out = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, dy, 1.0, -1.0, trans=True).get()
            ref = S.T @ y - x
            assert np.max(np.abs(out - ref)) <= 1e-12 * (1 + np.abs(S).sum(axis=0).max() + np.abs(x).max()), (case, m, n, density)
            cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get()
            assert np.allclose(cs, np.asarray(S.multiply(S).sum(axis=0)).ravel(), rtol=1e-13, atol=0), (case, m, n, density)
            rs = lsq.rowsumabs2_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J).get()
            assert np.allclose(rs, np.asarray(S.multiply(S).sum(axis=1)).ravel(), rtol=1e-13, atol=0), (case, m, n, density)
            J.free()
    finally:
        lsq.set_exact(None)


@pytest.mark.parametrize("kind", ["dense", "csr", "sliced"])
def test_rowsumabs2(ctx, kind):
    """rowsumabs2! (utils.jl:153-161, what colsumabs2! of an adjoint Jacobian computes) against the oracle:
    dense, CSR mirror and sliced rows."""
    m, n = (700, 90) if kind == "dense" else (5000, 300)
    if kind == "dense":
        A = np.random.default_rng(3).standard_normal((m, n))
        J, ref = lsq.DeviceMatrix(ctx, A), O.rowsumabs2(O.Mat(dense=A))
    else:
        S = rand_csc(m, n, 0.02, 77).tolil()
        S[7, :] = 0
        S[11, :] = np.random.default_rng(4).standard_normal(n)
        S = S.tocsc()
        S.sort_indices()
        S.eliminate_zeros()
        if kind == "sliced":
            os.environ["LSQ_SELL_FORCE"] = "1"
        try:
            J = lsq.DeviceMatrix(ctx, S)
        finally:
            os.environ.pop("LSQ_SELL_FORCE", None)
        ref = O.rowsumabs2(O.Mat.from_scipy(S))
    out = lsq.rowsumabs2_(lsq.DeviceVector(ctx, m, np.full(m, np.nan)), J).get()
    assert np.allclose(out, ref, rtol=1e-13, atol=0)
    if kind != "dense":
        assert out[7] == 0.0
    if kind in ("dense", "sliced"):      # left-to-right sums: the reference's order, bit for bit
        assert np.array_equal(out, ref)


def test_empty_and_tiny_sparse(ctx):
    S = sp.csc_matrix((5, 3))
    J = lsq.DeviceMatrix(ctx, S)
    out = lsq.mul_(lsq.DeviceVector(ctx, 5, np.ones(5)), J, lsq.DeviceVector(ctx, 3, np.ones(3)), 1.0, 2.0).get()
    assert np.all(out == 2.0)
    S = sp.csc_matrix(np.array([[2.0]]))
    J = lsq.DeviceMatrix(ctx, S)
    assert lsq.mul_(lsq.DeviceVector(ctx, 1), J, lsq.DeviceVector(ctx, 1, [3.0])).get()[0] == 6.0


@pytest.mark.parametrize("m,n", [(300, 17), (7, 40), (1025, 129), (20000, 12), (100001, 3), (8192, 255)])
def test_dense_products(ctx, m, n):
    """(the last three shapes take the window-blocked J'y / colsumabs2 of matrices with few columns)"""
    rng = np.random.default_rng(m * n)
    D = rng.standard_normal((m, n))
    J = lsq.DeviceMatrix(ctx, D)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    out = lsq.mul_(lsq.DeviceVector(ctx, m, y), J, lsq.DeviceVector(ctx, n, x), 0.5, 2.0).get()
    assert np.allclose(out, 0.5 * D @ x + 2 * y, rtol=1e-12, atol=1e-12)
    out = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, lsq.DeviceVector(ctx, m, y), 1.0, 0.0, trans=True).get()
    assert np.allclose(out, D.T @ y, rtol=1e-12, atol=1e-12)
    assert np.allclose(lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get(), (D * D).sum(0), rtol=1e-13)
    out2 = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, lsq.DeviceVector(ctx, m, y), -1.5, 0.25, trans=True).get()
    assert np.allclose(out2, -1.5 * D.T @ y + 0.25 * x, rtol=1e-12, atol=1e-12)
    again = lsq.mul_(lsq.DeviceVector(ctx, n, x), J, lsq.DeviceVector(ctx, m, y), -1.5, 0.25, trans=True).get()
    assert np.array_equal(out2, again)   # fixed-order window sums


def test_blas1(ctx):
    rng = np.random.default_rng(5)
    for n in (1, 63, 1000, 100003):
        x, y, w = rng.standard_normal(n), rng.standard_normal(n), rng.random(n)
        dx, dy, dw = (lsq.DeviceVector(ctx, n, v) for v in (x, y, w))
        assert lsq.sumsq(dx) == pytest.approx(np.sum(x * x), rel=1e-13)
        assert lsq.norm(dx) == pytest.approx(np.linalg.norm(x), rel=1e-13)
        assert lsq.wdot(dx, dy, dw) == pytest.approx(O.wdot(x, y, w), rel=1e-12, abs=1e-12)
        assert lsq.maxabs(dx) == np.max(np.abs(x))
        lo, hi = x - (rng.random(n) < 0.3) * 0.0 - 1.0 * (rng.random(n) < 0.5), x + 1.0
        lo[::2] = x[::2]  # half the coordinates sit on their lower bound
        dlo, dhi = lsq.DeviceVector(ctx, n, lo), lsq.DeviceVector(ctx, n, hi)
        assert lsq.maxabs_projected_gradient(dy, dx, dlo, dhi) == O.maxabs_projected_gradient(y, x, lo, hi)
    # run-to-run determinism of the two-stage reduction
    big = lsq.DeviceVector(ctx, 1 << 20, rng.standard_normal(1 << 20))
    assert len({lsq.sumsq(big) for _ in range(5)}) == 1


# ------------------------------------------------------------------------------- ldiv!
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("damped", [True, False])
def test_ldiv_lsmr(ctx, sparse, damped):
    m, n = 400, 60
    S = rand_csc(m, n, 0.1, 21)
    rng = np.random.default_rng(22)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.05
    Jh = S if sparse else S.toarray()
    A = O.Mat.from_scipy(S) if sparse else O.Mat(dense=S.toarray())
    J = lsq.DeviceMatrix(ctx, Jh)
    sv = lsq.AllocatedSolver(J, lsq.LSMR(), for_lm=damped)
    dy, dxo = lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n)
    if damped:
        dd = lsq.DeviceVector(ctx, n, damp)
        _, nmul = sv.ldiv_(dxo, dy, dd)
        st, xr, nmul_r, dafter = O.ldiv(O.LSMR, A, y, damp)
        assert np.allclose(dd.get(), dafter, rtol=1e-15)  # damp <- sqrt(damp) (iterative_lsmr.jl:252)
    else:
        _, nmul = sv.ldiv_(dxo, dy)
        st, xr, nmul_r = O.ldiv(O.LSMR, A, y)
    assert nmul == nmul_r and nmul > 0
    assert np.allclose(dxo.get(), xr, rtol=1e-8, atol=1e-10)
    assert np.array_equal(dy.get(), y)  # y preserved
    # zero right-hand side: ||A'b|| == 0 early exit, mvps == 0 (lsmr.jl:115)
    dz = lsq.DeviceVector(ctx, m)
    if damped:
        _, nmul0 = sv.ldiv_(dxo, dz, lsq.DeviceVector(ctx, n, damp))
    else:
        _, nmul0 = sv.ldiv_(dxo, dz)
    assert nmul0 == 0 and np.all(dxo.get() == 0)


@pytest.mark.parametrize("n", [1, 9, 70, 129, 200, 512, 700])
def test_ldiv_cholesky(ctx, n):
    rng = np.random.default_rng(30 + n)
    m = 3 * n + 5
    D = rng.standard_normal((m, n))
    y = rng.standard_normal(m)
    damp = rng.random(n)
    J = lsq.DeviceMatrix(ctx, D)
    dy, dxo = lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
    _, nmul = sv.ldiv_(dxo, dy, lsq.DeviceVector(ctx, n, damp))
    st, xr, _, _ = O.ldiv(O.CHOLESKY, O.Mat(dense=D), y, damp)
    assert nmul == 1 and np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12)
    # n >= 128 (and at most one 64 x 64 upper tile per CU): the whole factorisation is one launch (k_chol_chain; k_chol_tiles when tiles + 1 > CUs)
    assert sv.info()["chol_path"] == ("blocked-one-launch" if n >= 128 else "blocked" if n >= 32 else "one-workgroup")
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=False)  # pivoted (Dogleg)
    _, nmul = sv.ldiv_(dxo, dy)
    st, xr, _ = O.ldiv(O.CHOLESKY, O.Mat(dense=D), y)
    assert np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12)


def test_cholesky_one_launch_is_repeatable(ctx):
    """k_chol_chain / k_chol_tiles: workgroups hand tiles to each other inside ONE launch (epoch-tagged flags, bounded waits).  300 solves on
    one solver: every result equals the first bit for bit and no wait ever gave up (the path would fall back to
    'blocked' and stay there).  tools/chol_stress.py runs the same check for thousands of solves."""
    rng = np.random.default_rng(5)
    m, n = 3000, 700                       # 11 x 11 tiles, ragged last tile
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    J = lsq.DeviceMatrix(ctx, A)
    y, damp, x = lsq.DeviceVector(ctx, m, rng.standard_normal(m)), lsq.DeviceVector(ctx, n, rng.random(n) + 0.01), lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
    sv.ldiv_(x, y, damp)
    ref = x.get().copy()
    assert np.allclose(ref, O.ldiv(O.CHOLESKY, O.Mat(dense=A), y.get(), damp.get())[1], rtol=1e-9, atol=1e-12)
    for _ in range(300):
        sv.ldiv_(x, y, damp)
        assert sv.info()["chol_path"] == "blocked-one-launch"
        assert np.array_equal(x.get(), ref)


@pytest.mark.parametrize("cond,certified", [(1e1, True), (1e4, True), (1e7, False)])
def test_ldiv_cholesky_dogleg_certificate(ctx, cond, certified):
    """Dogleg's Cholesky() is the pivoted cholesky!(Symmetric(J'J), Val(true)) (dense_cholesky.jl:33).  The blocked
    unpivoted factorisation may replace it only when 1/||inv(U)||_F^2 > 16 n eps max diag(J'J) proves that no
    pivot of dpstrf can fail; otherwise the pivoted kernel runs.  Singular values of J graded down to 1/cond
    (cond(J'J) = cond^2): certified at 1e1 and 1e4, not at 1e7 -- the oracle's pivoted solve is matched in
    every case; a duplicated column must still raise RankDeficientException through the fallback."""
    m, n = 400, 96
    rng = np.random.default_rng(int(np.log10(cond)) + 5)
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    D = (U * np.logspace(0, -np.log10(cond), n)) @ V.T
    y = rng.standard_normal(m)
    J = lsq.DeviceMatrix(ctx, D)
    dxo = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=False)
    sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    assert sv.info()["chol_path"] == ("blocked-certified" if certified else "one-workgroup")
    st, xr, _ = O.ldiv(O.CHOLESKY, O.Mat(dense=D), y)
    tol = max(1e-9, 100 * cond * cond * np.finfo(float).eps)
    assert np.linalg.norm(dxo.get() - xr) <= tol * np.linalg.norm(xr)
    # an exactly zero column gives an exactly zero pivot: dpstrf (tol = 0) stops, and so must the fallback
    D2 = D.copy()
    D2[:, 40] = 0.0
    assert O.ldiv(O.CHOLESKY, O.Mat(dense=D2), y)[0] != 0
    sv2 = lsq.AllocatedSolver(lsq.DeviceMatrix(ctx, D2), lsq.Cholesky(), for_lm=False)
    with pytest.raises(lsq.RankDeficientException):
        sv2.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    assert sv2.info()["chol_path"] == "one-workgroup"
    # a duplicated column leaves a pivot of rounding noise: with tol = 0 the reference only stops if the noise
    # happens to be non-positive -- the certificate must refuse either way, the outcome is the pivoted kernel's
    D3 = D.copy()
    D3[:, 40] = D3[:, 7]
    sv3 = lsq.AllocatedSolver(lsq.DeviceMatrix(ctx, D3), lsq.Cholesky(), for_lm=False)
    try:
        sv3.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    except lsq.RankDeficientException:
        pass
    assert sv3.info()["chol_path"] == "one-workgroup"


def test_cholesky_failures(ctx):
    rng = np.random.default_rng(33)
    D = rng.standard_normal((30, 6))
    D[:, 4] = D[:, 2]
    J = lsq.DeviceMatrix(ctx, D)
    dy, dxo = lsq.DeviceVector(ctx, 30, rng.standard_normal(30)), lsq.DeviceVector(ctx, 6)
    with pytest.raises(lsq.RankDeficientException):
        lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=False).ldiv_(dxo, dy)
    D[:, 4] = 0.0
    J = lsq.DeviceMatrix(ctx, D)
    with pytest.raises(lsq.PosDefException):
        lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True).ldiv_(dxo, dy, lsq.DeviceVector(ctx, 6))
    # the same through the one-launch blocked factorisation: a zero column in the third 64-block, no damping there
    D = rng.standard_normal((900, 256))
    D[:, 150] = 0.0
    J = lsq.DeviceMatrix(ctx, D)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
    with pytest.raises(lsq.PosDefException, match="151"):
        sv.ldiv_(lsq.DeviceVector(ctx, 256), lsq.DeviceVector(ctx, 900, rng.standard_normal(900)), lsq.DeviceVector(ctx, 256))
    assert sv.info()["chol_path"] == "blocked-one-launch"
    # ... in the first block of the chain, and in the ragged last tile (k_chol_chain: the chain workgroup reports the
    # position and releases every flag it owns, nobody is left waiting); the solver is usable again afterwards
    for n, col in ((200, 3), (200, 197), (512, 448)):
        D = rng.standard_normal((3 * n, n))
        D[:, col] = 0.0
        sv = lsq.AllocatedSolver(lsq.DeviceMatrix(ctx, D), lsq.Cholesky(), for_lm=True)
        y = rng.standard_normal(3 * n)
        with pytest.raises(lsq.PosDefException, match=str(col + 1)):
            sv.ldiv_(lsq.DeviceVector(ctx, n), lsq.DeviceVector(ctx, 3 * n, y), lsq.DeviceVector(ctx, n))
        dx = lsq.DeviceVector(ctx, n)
        sv.ldiv_(dx, lsq.DeviceVector(ctx, 3 * n, y), lsq.DeviceVector(ctx, n, np.ones(n)))
        assert np.allclose(dx.get(), O.ldiv(O.CHOLESKY, O.Mat(dense=D), y, np.ones(n))[1], rtol=1e-9, atol=1e-12)
        assert sv.info()["chol_path"] == "blocked-one-launch" and sv.stats()["chol_one_launch"]["giveups"] == 0


@pytest.mark.parametrize("m,n,rank", [(40, 10, 10), (12, 12, 12), (30, 12, 7), (9, 6, 5), (20, 8, 1),
                                      (6, 10, 6), (6, 10, 4), (300, 65, 65), (1100, 130, 130), (900, 100, 37),
                                      (80, 900, 80)])
def test_ldiv_qr(ctx, m, n, rank):
    rng = np.random.default_rng(100 + m + n + rank)
    A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))
    y = rng.standard_normal(m)
    J = lsq.DeviceMatrix(ctx, A)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    dxo = lsq.DeviceVector(ctx, n)
    _, nmul = sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    xr, rk, *_ = O.qr_solve(A, y)
    assert nmul == 1 and sv.info()["qr_rank"] == rk == rank
    assert np.allclose(dxo.get(), xr, rtol=1e-8, atol=1e-10)
    if rank == min(m, n) and m >= n:
        damp = rng.random(n) + 0.01
        svd = lsq.AllocatedSolver(J, lsq.QR(), for_lm=True)
        svd.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        st, xr, _, _ = O.ldiv(O.QR, O.Mat(dense=A), y, damp)
        assert np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("m,n,rank", [(300, 65, 65), (1100, 130, 130), (900, 100, 37), (700, 200, 1), (640, 128, 128),
                                      (2000, 321, 321), (2000, 321, 300), (500, 500, 500),
                                      (100, 20, 20), (100, 20, 7), (64, 64, 64), (40, 33, 33), (70, 17, 16), (33, 16, 16)])
def test_ldiv_qr_two_stage(ctx, m, n, rank, monkeypatch):
    """The two-stage factorisation (unpivoted blocked Householder with MFMA trailing updates, then either the
    full-rank certificate or the pivoted sweep on R) forced onto small problems: panel tails (n not a
    multiple of 64), rank-deficient inputs (rank decision and minimum-norm completion come from the
    pivoted stage: the certificate must refuse them), square and stacked [J; sqrt(damp)] operands.
    Same expectations as the one-stage path."""
    rng = np.random.default_rng(300 + m + n + rank)
    A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))
    y = rng.standard_normal(m)
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    xr, rk, *_ = O.qr_solve(A, y)
    sols = {}
    for envs in (("LSQ_QR_TWO_STAGE",), ("LSQ_QR_TWO_STAGE", "LSQ_QR_ALWAYS_PIVOT"), ("LSQ_QR_ONE_STAGE",)):
        for env in envs:
            monkeypatch.setenv(env, "1")
        sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
        _, nmul = sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        info = sv.info()
        assert nmul == 1 and info["qr_rank"] == rk == rank, envs
        want = ("one-stage" if "LSQ_QR_ONE_STAGE" in envs else
                "two-stage-certified" if rank == n and "LSQ_QR_ALWAYS_PIVOT" not in envs else "two-stage-pivoted")
        assert info["qr_path"] == want, (envs, info)
        sols[envs] = dxo.get()
        assert np.allclose(sols[envs], xr, rtol=1e-8, atol=1e-10), envs
        if rank == n:
            damp = rng.random(n) + 0.01
            svd = lsq.AllocatedSolver(J, lsq.QR(), for_lm=True)
            svd.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
            st, xd, _, _ = O.ldiv(O.QR, O.Mat(dense=A), y, damp)
            assert np.allclose(dxo.get(), xd, rtol=1e-9, atol=1e-12)
        for env in envs:
            monkeypatch.delenv(env)
    ref = sols[("LSQ_QR_ONE_STAGE",)]
    for k, v in sols.items():
        assert np.allclose(v, ref, rtol=1e-8, atol=1e-10), k


@pytest.mark.parametrize("m,n", [(2049, 321), (1500, 700), (777, 513), (4099, 200), (1030, 1030), (12000, 200), (16500, 200)])
def test_qr_wave_private_kernels_and_lookahead(ctx, m, n, monkeypatch):
    """Round 5's trailing kernels of the blocked QR (k_qr1_vtb_w, k_qr1_update_w: wave-private tiles, no LDS), round 6's panel forms
    (Q1 form / three passes, Neumann product / modified LU, Gram partials from the update / from pass 0, group sums) and the
    look-ahead of the next panel against the LDS-staged kernels of rounds 2-4 and against the oracle (dense_qr.jl:30-88), on
    shapes that exercise every ragged edge: rows not a multiple of 32 / 64, trailing columns not a multiple of 16, odd leading
    dimensions (8-byte aligned fragments), the right-hand side as the last column of the last tile, the stacked damped
    operand.  The wave-private kernels sum the 64-term dot products in a different order (k permuted, two halves): agreement
    to 1e-11 relative, not bit for bit; look-ahead on / off must agree bit for bit (same kernels, same order)."""
    rng = np.random.default_rng(9000 + m + n)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    xr, rk, *_ = O.qr_solve(A, y)
    st, xd, _, _ = O.ldiv(O.QR, O.Mat(dense=A), y, damp)
    monkeypatch.setenv("LSQ_QR_TWO_STAGE", "1")
    # round 6: the default is the Q1 form of the panel (no pass 2, Gram partials of the next panel formed by the update, Neumann
    # product for tall panels' reflector kernel, look-ahead for panels of more slabs than CUs only); every switch that selects another
    # form is exercised here
    LA = {"LSQ_QR_LOOKAHEAD": "1", "LSQ_QR_LOOKAHEAD_MINCOLS": "0"}
    combos = {"default": {},
              "no_lookahead": {"LSQ_QR_LOOKAHEAD": "0"},
              "q1_lookahead_everywhere": dict(LA),
              "q1_redundant_factor_ahead": dict(LA, LSQ_QR_AHEAD_REDUNDANT="1"),
              "q1_top_lu": {"LSQ_QR_TOP_LU": "1"},
              "q1_no_fused_gram": {"LSQ_QR_NO_FUSED_GRAM": "1"},
              "q1_no_vtb_lds": {"LSQ_QR_VTB_LDS": "0"},
              "q1_group_sums": {"LSQ_QR_HIER": "1"},
              "update_grid_per_row_group": {"LSQ_QR_UPDATE_FLAT": "0"},
              "update_two_runs_per_cu": {"LSQ_QR_UPDATE_FLAT": "2"},
              "operands_in_matrix_order": {"LSQ_QR_NO_SWIZZLE": "1"},
              "three_pass": {"LSQ_QR_CQR_PASS2": "1", "LSQ_QR_LOOKAHEAD": "0"},
              "three_pass_lookahead_everywhere": dict(LA, LSQ_QR_CQR_PASS2="1"),
              "lds_update": {"LSQ_QR_UPDATE_W": "0"},
              "lds_vtb": {"LSQ_QR_VTB_W": "0"},
              "lds_both": {"LSQ_QR_UPDATE_W": "0", "LSQ_QR_VTB_W": "0"}}
    got, gotd = {}, {}
    for name, env in combos.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
        sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        info = sv.info()
        assert info["qr_path"] == "two-stage-certified" and info["qr_panel"] == "cholqr2" and info["qr_rank"] == rk == n, (name, info)
        got[name] = dxo.get()
        svd = lsq.AllocatedSolver(J, lsq.QR(), for_lm=True)
        svd.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        gotd[name] = dxo.get()
        for k in env:
            monkeypatch.delenv(k)
    scale, scaled = np.max(np.abs(xr)), np.max(np.abs(xd))
    for name in combos:
        assert np.max(np.abs(got[name] - xr)) <= 1e-9 * scale, name
        assert np.max(np.abs(gotd[name] - xd)) <= 1e-9 * scaled, name
        assert np.max(np.abs(got[name] - got["lds_both"])) <= 1e-11 * scale, name
        assert np.max(np.abs(gotd[name] - gotd["lds_both"])) <= 1e-11 * scaled, name
    # same kernels, same order => same bits: the default has no look-ahead for panels of at most 256 slabs of 64 rows (one per CU;
    # at 16500 x 200 the first panel after the leading one is factored ahead, the next is not: the rule's transition); the Gram
    # partials formed by the update are the ones pass 0 forms; the LDS reservation changes placement only.  (With look-ahead
    # pass 1 multiplies by the explicit inverse that one workgroup formed instead of substituting: agreement to 1e-11, above.)
    pairs = [("q1_no_fused_gram", "default"), ("q1_no_vtb_lds", "default"),
             # (which workgroup takes a tile changes nothing about the tile)
             ("update_grid_per_row_group", "default"), ("update_two_runs_per_cu", "default"),
             # (W2 and Q1 read from their fragment-order copies or from the matrices themselves: the same numbers)
             ("operands_in_matrix_order", "default")]
    tall = m - 64 > 256 * 64
    if m + n - 64 <= 256 * 64:
        pairs.append(("default", "no_lookahead"))
    for a, b in pairs:
        assert np.array_equal(got[a], got[b]) and np.array_equal(gotd[a], gotd[b]), (a, b)
    if tall:
        assert not np.array_equal(got["default"], got["no_lookahead"])      # (the rule did select the other path)


@pytest.mark.parametrize("m", [3000, 6000, 12000, 18000, 22000])
@pytest.mark.parametrize("coop", ["1", "0"])
def test_ldiv_qr_panel_variants(ctx, m, coop, monkeypatch):
    """Stage 1's panel steps pick their kernel by the number of active rows: one workgroup per column
    (<= 2048 rows, or LSQ_QR1_COOP=0), 2 / 4 / 8 row slabs per column with the in-kernel exchange of partial
    sums (<= 4096 / 8192 / 20480 rows), the looping kernel beyond.  Every variant must reproduce the oracle's
    pivoted-QR solve (n = 130: two full panels and a ragged one; the later pivot columns of a launch travel
    through the side panel)."""
    n = 130
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    xr, rk, *_ = O.qr_solve(A, y)
    monkeypatch.setenv("LSQ_QR_TWO_STAGE", "1")
    monkeypatch.setenv("LSQ_QR1_COOP", coop)
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    for pivot in (False, True):
        if pivot:
            monkeypatch.setenv("LSQ_QR_ALWAYS_PIVOT", "1")
        sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
        sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        info = sv.info()
        assert info["qr_rank"] == rk == n
        assert info["qr_path"] == ("two-stage-pivoted" if pivot else "two-stage-certified")
        assert np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12), (m, coop, pivot)
        # repeated solves reuse the exchange slots (epoch advances)
        sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        assert np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12), (m, coop, pivot, "second solve")


@pytest.mark.parametrize("tsqr", ["default", "off"])
@pytest.mark.parametrize("m,n", [(30000, 20), (100000, 20), (300000, 12), (1200000, 8), (2200000, 6), (50000, 70),
                                 (40000, 31), (33000, 2), (70000, 17), (90000, 25)])
def test_ldiv_qr_tall_thin(ctx, m, n, tsqr, monkeypatch):
    """Tall, thin operands (the usual shape of a fitting problem: many residuals, few parameters).  Default: from
    32768 rows and n <= 32 TSQR levels (every wavefront factors its own slab of rows in registers, without LDS or
    barriers; levels repeat until the stacked triangles are short enough for the panel machinery) -- one pass
    over the matrix.  LSQ_QR_NO_TSQR=1 (and n > 32, or
    fewer rows): 16 / 64 / 256 row slabs per column with the generalised exchange, the right-hand side riding
    through the last panel's steps as one more target column.  Oracle = the reference's pivoted-QR solve;
    repeated solve bit-identical (fixed-order reductions and exchanges)."""
    if tsqr == "off":
        monkeypatch.setenv("LSQ_QR_NO_TSQR", "1")
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    xr, rk, *_ = O.qr_solve(A, y)
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    info = sv.info()
    assert info["qr_rank"] == rk == n and info["qr_path"] == "two-stage-certified"
    x1 = dxo.get()
    assert np.allclose(x1, xr, rtol=1e-9, atol=1e-12)
    sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    assert np.array_equal(dxo.get(), x1)
    # LM's stacked operand [J; sqrt(damp)] through the same kernels
    damp = rng.random(n) + 0.01
    svd = lsq.AllocatedSolver(J, lsq.QR(), for_lm=True)
    svd.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
    st, xd, _, _ = O.ldiv(O.QR, O.Mat(dense=A), y, damp)
    assert np.allclose(dxo.get(), xd, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("for_lm", [False, True])
def test_qr_cholqr_panel_and_its_fallback(ctx, for_lm):
    """Stage 1 of the blocked QR factors its 64-column panels by CholeskyQR2 + a basis-kernel block reflector
    (lsq_qr_cholqr.hip).  (i) On ordinary operands it is the path taken and the solve agrees with LAPACK and with the
    column-by-column Householder panel (LSQ_QR1_NO_CHOLQR=1) to rounding.  (ii) A panel too ill-conditioned for it (cond
    1e10: the Gram matrix is numerically singular) raises the device flag; the solve is repeated with the Householder
    panel and the answer is still the backward-stable one.  (iii) Exactly rank-deficient operands end in the pivoted
    sweep as before."""
    rng = np.random.default_rng(11)
    m, n = 3000, 192

    def solve(A, y, damp=None, env=None):
        if env:
            os.environ[env] = "1"
        try:
            J = lsq.DeviceMatrix(ctx, A)
            sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=damp is not None)
            x = lsq.DeviceVector(ctx, A.shape[1])
            if damp is None:
                sv.ldiv_(x, lsq.DeviceVector(ctx, A.shape[0], y))
            else:
                sv.ldiv_(x, lsq.DeviceVector(ctx, A.shape[0], y), lsq.DeviceVector(ctx, A.shape[1], damp))
            out, info = x.get(), sv.info()
            sv.free()
            J.free()
            return out, info
        finally:
            if env:
                del os.environ[env]

    y = rng.standard_normal(m)
    damp = (rng.random(n) + 0.05) if for_lm else None

    def reference(A):
        if damp is None:
            return np.linalg.lstsq(A, y, rcond=None)[0]
        return np.linalg.lstsq(np.vstack([A, np.diag(np.sqrt(damp))]), np.concatenate([y, np.zeros(n)]), rcond=None)[0]

    # (i) well conditioned
    A = rng.standard_normal((m, n))
    x, info = solve(A, y, damp)
    assert info["qr_panel"] == "cholqr2" and info["qr_path"] == "two-stage-certified"
    xh, infoh = solve(A, y, damp, env="LSQ_QR1_NO_CHOLQR")
    assert infoh["qr_panel"] == "householder-steps"
    ref = reference(A)
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref) and np.linalg.norm(x - xh) <= 1e-12 * np.linalg.norm(ref)
    # moderately ill conditioned (cond 1e5): still CholeskyQR2 (second pass by Cholesky, not by the expansion)
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A5 = U @ np.diag(np.logspace(0, -5, n)) @ V.T
    x, info = solve(A5, y, damp)
    ref = reference(A5)
    assert info["qr_panel"] == "cholqr2"
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    # (ii) a panel with two columns parallel to 1e-9: its Gram matrix is numerically singular -> breakdown -> Householder
    # panel, and the answer is as good as LAPACK's own (cond * eps)
    A10 = rng.standard_normal((m, n))
    A10[:, 1] = A10[:, 0] + 1e-9 * rng.standard_normal(m)
    x, info = solve(A10, y, damp)
    ref = reference(A10)
    if not for_lm:          # (with damping the stacked operand is well conditioned again: nothing to fall back from)
        assert info["qr_panel"] == "householder-steps"
        res = lambda v: np.linalg.norm(A10 @ v - y)
        assert res(x) <= res(ref) * (1 + 1e-10)          # (x itself is determined to cond * eps ~ 1e-6 only)
    else:
        assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    # (iii) rank deficient: minimum-norm solution through the pivoted sweep
    if not for_lm:
        Ad = rng.standard_normal((m, 40)) @ rng.standard_normal((40, n))
        x, info = solve(Ad, y)
        ref = np.linalg.lstsq(Ad, y, rcond=None)[0]
        assert info["qr_path"] == "two-stage-pivoted"
        assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref)


@pytest.mark.parametrize("cond,certified", [(1e2, True), (1e6, True), (1e11, True), (1e12, False)])
def test_ldiv_qr_certificate_decision(ctx, cond, certified, monkeypatch):
    """The full-rank certificate (||R||_F ||inv(R)||_F * rcond * 16 <= 1) may only skip the pivoted sweep when
    xGELSY's rank decision (dense_qr.jl:37,83 -> geqp3 + laic1, rcond = min(m,n) eps) is certain to be n.
    Singular values graded from 1 down to 1/cond (n = 192: 1/rcond = 2.3e13, the certificate needs a
    Frobenius bound <= 1.5e12): inside the threshold the certificate fires; at cond 1e12 the reference still
    finds rank n but the bound (about 4e12) cannot prove it, so the pivoted sweep must run -- and in every
    case rank and solution are the oracle's."""
    m, n = 700, 192
    rng = np.random.default_rng(int(np.log10(cond)) + 77)
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    sv_ = np.logspace(0, -np.log10(cond), n)
    A = (U * sv_) @ V.T
    y = rng.standard_normal(m)
    xr, rk, *_ = O.qr_solve(A, y)
    monkeypatch.setenv("LSQ_QR_TWO_STAGE", "1")
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    info = sv.info()
    assert info["qr_path"] == ("two-stage-certified" if certified else "two-stage-pivoted"), info
    assert info["qr_rank"] == rk
    if certified:
        assert rk == n
    # forward error of a backward-stable solve: cond * eps (relative to the solution's size)
    tol = max(1e-9, 50 * min(cond, 1e13) * np.finfo(float).eps)
    assert np.linalg.norm(dxo.get() - xr) <= tol * np.linalg.norm(xr)


def test_ldiv_qr_certificate_wide_triangle(ctx):
    """n beyond the single-workgroup substitution (n > 2048): the certified path solves with the explicit
    inverse; the pivoted path is the reference order.  Both must satisfy the normal equations
    J'(Jx - y) = 0 to rounding and agree with each other (size-independent property; the oracle's
    unblocked sweep would take minutes here)."""
    m, n = 2304, 2112
    rng = np.random.default_rng(2112)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
    info = sv.info()
    assert info["qr_path"] == "two-stage-certified" and info["qr_rank"] == n
    x = dxo.get()
    g = A.T @ (A @ x - y)
    assert np.linalg.norm(g) <= 1e-10 * np.linalg.norm(A.T @ y)
    xl = np.linalg.lstsq(A, y, rcond=None)[0]
    assert np.linalg.norm(x - xl) <= 1e-9 * np.linalg.norm(xl)


def test_lsmr_custom_preconditioner(ctx):
    """LSMR(preconditioner!, P) (types.jl:82-86): a host callback that restates the default Jacobi rule
    (iterative_lsmr.jl:129-141) must reproduce the built-in solver bit for bit; the identity preconditioner
    must reach the same least-squares solution; both through ldiv! and through optimize!."""
    lsq.set_exact(False)
    try:
        m, n = 3000, 120
        S = rand_csc(m, n, 0.05, 314)
        rng = np.random.default_rng(15)
        y = rng.standard_normal(m)
        damp = rng.random(n) + 0.05
        J = lsq.DeviceMatrix(ctx, S)
        calls = []

        def jacobi(P, Jm, dmp):
            cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), Jm).get()
            if dmp is not None:
                cs = cs + dmp.get()
            calls.append(dmp is not None)
            P.set(np.where(cs > 0, 1.0 / np.sqrt(cs), 0.0))

        def identity(P, Jm, dmp):
            P.set(np.ones(n))

        for damped in (True, False):
            ref = lsq.DeviceVector(ctx, n)
            sv0 = lsq.AllocatedSolver(J, lsq.LSMR(), for_lm=damped)
            args = (lsq.DeviceVector(ctx, m, y),) + ((lsq.DeviceVector(ctx, n, damp),) if damped else ())
            _, nm0 = sv0.ldiv_(ref, *args)
            out = lsq.DeviceVector(ctx, n)
            sv1 = lsq.AllocatedSolver(J, lsq.LSMR(jacobi), for_lm=damped)
            args = (lsq.DeviceVector(ctx, m, y),) + ((lsq.DeviceVector(ctx, n, damp),) if damped else ())
            _, nm1 = sv1.ldiv_(out, *args)
            assert nm1 == nm0 and np.array_equal(out.get(), ref.get())
            sv2 = lsq.AllocatedSolver(J, lsq.LSMR(identity), for_lm=damped)
            args = (lsq.DeviceVector(ctx, m, y),) + ((lsq.DeviceVector(ctx, n, damp),) if damped else ())
            sv2.ldiv_(out, *args)
            if not damped:   # Dogleg's tolerances (atol = btol = 1e-6): same minimiser up to the stopping rule
                assert np.allclose(out.get(), ref.get(), rtol=1e-3, atol=1e-4)
        assert calls == [True, False]
        # only LSMR takes a preconditioner (types.jl:82-86)
        svc = lsq.AllocatedSolver(lsq.DeviceMatrix(ctx, np.eye(4)), lsq.Cholesky(), for_lm=True)
        cb = lsq._lib.PRECOND_CALLBACK(lambda *a: 0)
        assert lsq.lib().lsq_solver_set_preconditioner(svc.h, cb, None) == lsq._lib.EARG
        # whole loop: optimize! with LevenbergMarquardt(LSMR(jacobi)) == default
        p = list(P.minpack_all())[0]
        name, f, g, x0 = p[:4]
        nn = len(x0)
        res = []
        for solver in (lsq.LSMR(), lsq.LSMR(lambda Pv, Jm, dmp: Pv.set(
                (lambda cs: np.where(cs > 0, 1.0 / np.sqrt(cs), 0.0))(lsq.colsumabs2_(lsq.DeviceVector(ctx, nn), Jm).get()
                                                                     + (dmp.get() if dmp is not None else 0.0))))):
            nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(nn), f_=f, g_=g, J=np.zeros((nn, nn), order="F"))
            res.append(lsq.optimize_(nls, lsq.LevenbergMarquardt(solver)))
        assert res[0].iterations == res[1].iterations and res[0].ssr == pytest.approx(res[1].ssr, rel=1e-9)
    finally:
        lsq.set_exact(None)


def test_lsmr_general_preconditioner(ctx):
    """LSMR(preconditioner!, P) with ANY P that supports ldiv! (README.md:47 of the reference; types.jl:82-86), through the
    operator-level recurrence of lsq_lsmr_general.hip:
    (1) the default Jacobi rule restated as a general P (an object with ldiv) must walk the built-in solver's iterations:
        identical mvps, solution to 1e-10 -- damped (LM) and undamped (Dogleg);
    (2) a genuinely NON-diagonal, symmetric P -- the symmetric square root of 4 x 4 diagonal blocks of J'J + diag(damp) --
        against scipy's LSMR (the same Fong-Saunders algorithm) run on the explicitly preconditioned operator
        [J; sqrt(damp)] inv(P) with the reference's tolerances: equal iteration counts, x to 1e-8;
    (3) the whole loop: optimize! with LevenbergMarquardt(LSMR(update, P)) for the Jacobi-as-general P equals the default run."""
    import scipy.sparse.linalg as spla
    lsq.set_exact(False)
    try:
        m, n = 3000, 120
        S = rand_csc(m, n, 0.05, 314)
        rng = np.random.default_rng(15)
        y = rng.standard_normal(m)
        damp = rng.random(n) + 0.05
        J = lsq.DeviceMatrix(ctx, S)
        Sd = S.toarray()

        class JacobiP:
            def __init__(self):
                self.d = np.ones(n)

            def ldiv(self, out, x):            # ldiv!(out, ::InverseDiagonal, x) = x .* stored inverse
                out.set(x.get() * self.d)

        def jacobi_update(Pobj, Jm, dmp):
            cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), Jm).get() + (dmp.get() if dmp is not None else 0.0)
            Pobj.d = np.where(cs > 0, 1.0 / np.sqrt(cs), 0.0)

        class BlockP:
            """P = blockdiag(sqrtm(B_k)), B_k the 4 x 4 diagonal blocks of J'J + diag(damp): symmetric positive definite."""

            def __init__(self):
                self.Pinv = np.eye(n)

            def ldiv(self, out, x):
                out.set(self.Pinv @ x.get())

        def block_update(Pobj, Jm, dmp):
            G = Sd.T @ Sd + (np.diag(dmp.get()) if dmp is not None else 0.0)
            Pinv = np.zeros((n, n))
            for k in range(0, n, 4):
                w, V = np.linalg.eigh(G[k:k + 4, k:k + 4])
                Pinv[k:k + 4, k:k + 4] = (V / np.sqrt(w)) @ V.T
            Pobj.Pinv = Pinv

        for damped in (True, False):
            args = lambda: (lsq.DeviceVector(ctx, m, y),) + ((lsq.DeviceVector(ctx, n, damp),) if damped else ())  # noqa: E731
            ref = lsq.DeviceVector(ctx, n)
            _, nm0 = lsq.AllocatedSolver(J, lsq.LSMR(), for_lm=damped).ldiv_(ref, *args())
            out = lsq.DeviceVector(ctx, n)
            _, nm1 = lsq.AllocatedSolver(J, lsq.LSMR(jacobi_update, JacobiP()), for_lm=damped).ldiv_(out, *args())
            assert nm1 == nm0 > 0
            assert np.max(np.abs(out.get() - ref.get())) <= 1e-10 * max(1.0, np.max(np.abs(ref.get())))
            # (2) non-diagonal P vs scipy on the explicit operator
            Pb = BlockP()
            _, nm2 = lsq.AllocatedSolver(J, lsq.LSMR(block_update, Pb), for_lm=damped).ldiv_(out, *args())
            A = np.vstack([Sd, np.diag(np.sqrt(damp))]) if damped else Sd
            b = np.concatenate([y, np.zeros(n)]) if damped else y
            z, istop, itn = spla.lsmr(A @ Pb.Pinv, b, atol=1e-6, btol=0.5 if damped else 1e-6, conlim=1e8,
                                      maxiter=max(A.shape))[:3]
            assert nm2 == 2 * itn, (nm2, itn, istop)
            xs = Pb.Pinv @ z
            assert np.max(np.abs(out.get() - xs)) <= 1e-8 * max(1.0, np.max(np.abs(xs)))
        # only LSMR takes one
        svc = lsq.AllocatedSolver(lsq.DeviceMatrix(ctx, np.eye(4)), lsq.Cholesky(), for_lm=True)
        assert lsq.lib().lsq_solver_set_general_preconditioner(svc.h, lsq._lib.PRECOND_UPDATE_CALLBACK(lambda *a: 0),
                                                               lsq._lib.PRECOND_LDIV_CALLBACK(lambda *a: 0), None) == lsq._lib.EARG
        # (3) whole loop
        p = list(P.minpack_all())[0]
        name, f, g, x0 = p[:4]
        nn = len(x0)

        class JP:
            d = np.ones(nn)

            def ldiv(self, o, x):
                o.set(x.get() * self.d)

        def jp_update(Pobj, Jm, dmp):
            cs = lsq.colsumabs2_(lsq.DeviceVector(ctx, nn), Jm).get() + (dmp.get() if dmp is not None else 0.0)
            Pobj.d = np.where(cs > 0, 1.0 / np.sqrt(cs), 0.0)

        res = []
        for solver in (lsq.LSMR(), lsq.LSMR(jp_update, JP())):
            nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(nn), f_=f, g_=g, J=np.zeros((nn, nn), order="F"))
            res.append(lsq.optimize_(nls, lsq.LevenbergMarquardt(solver)))
        assert res[0].iterations == res[1].iterations and res[0].mul_calls == res[1].mul_calls
        assert res[0].ssr == pytest.approx(res[1].ssr, rel=1e-9, abs=1e-20)
    finally:
        lsq.set_exact(None)


def test_allocated_problem_is_reusable(ctx):
    """LeastSquaresProblemAllocated (types.jl:141-160; exported): allocate once, optimize! repeatedly -- same
    results as fresh problems, from the same and from a different start."""
    p = list(P.minpack_all())[2]
    name, f, g, x0 = p[:4]
    n = len(x0)
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(n), f_=f, g_=g, J=np.zeros((n, n), order="F"))
    nlsa = lsq.LeastSquaresProblemAllocated(nls, lsq.Dogleg(lsq.QR()))
    assert isinstance(nlsa.optimizer, lsq.Dogleg) and isinstance(nlsa.solver, lsq.QR)
    r1 = lsq.optimize_(nlsa)
    fresh = lsq.optimize_(lsq.LeastSquaresProblem(x=x0.copy(), y=np.zeros(n), f_=f, g_=g, J=np.zeros((n, n), order="F")),
                          lsq.Dogleg(lsq.QR()))
    assert r1.iterations == fresh.iterations and r1.ssr == fresh.ssr and np.array_equal(r1.minimizer, fresh.minimizer)
    x1 = x0 * 1.5 + 0.1
    nlsa.x[:] = x1
    r2 = lsq.optimize_(nlsa)
    fresh2 = lsq.optimize_(lsq.LeastSquaresProblem(x=x1.copy(), y=np.zeros(n), f_=f, g_=g, J=np.zeros((n, n), order="F")),
                           lsq.Dogleg(lsq.QR()))
    assert r2.iterations == fresh2.iterations and r2.ssr == fresh2.ssr
    with pytest.raises(TypeError):
        lsq.optimize_(nlsa, lsq.LevenbergMarquardt())
    nlsa.free()


def test_matrix_free_operator(ctx):
    """A custom Jacobian type (README.md:37-47 of the reference): host callbacks for mul!, the adjoint's mul!
    and colsumabs2!.  Wrapping a stored matrix in such an operator must give the SAME LSMR solve (the library
    runs its fused epilogues over the callback's product) and the same LM trajectory on the tanh model."""
    lsq.set_exact(False)
    try:
        m, n = 4000, 150
        S = rand_csc(m, n, 0.04, 99)
        J = lsq.DeviceMatrix(ctx, S)
        count = {"mul": 0, "mulT": 0, "cs": 0}

        def op_mul(trans, x, out):
            count["mulT" if trans else "mul"] += 1
            lsq.mul_(out, J, x, 1.0, 0.0, trans=trans)

        def op_cs(out):
            count["cs"] += 1
            lsq.colsumabs2_(out, J)

        Op = lsq.DeviceOperator(ctx, m, n, op_mul, op_cs)
        rng = np.random.default_rng(8)
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        dx, dy = lsq.DeviceVector(ctx, n, x), lsq.DeviceVector(ctx, m, y)
        for trans, a, b in ((False, 1.5, -0.5), (True, -2.0, 0.25)):
            v0 = lsq.DeviceVector(ctx, n if trans else m, x if trans else y)
            v1 = lsq.DeviceVector(ctx, n if trans else m, x if trans else y)
            lsq.mul_(v0, J, dy if trans else dx, a, b, trans=trans)
            lsq.mul_(v1, Op, dy if trans else dx, a, b, trans=trans)
            assert np.array_equal(v0.get(), v1.get())
        assert np.array_equal(lsq.colsumabs2_(lsq.DeviceVector(ctx, n), Op).get(),
                              lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get())
        damp = rng.random(n) + 0.1
        for damped in (True, False):
            outs = []
            for M in (J, Op):
                sv = lsq.AllocatedSolver(M, lsq.LSMR(), for_lm=damped)
                out = lsq.DeviceVector(ctx, n)
                args = (lsq.DeviceVector(ctx, m, y),) + ((lsq.DeviceVector(ctx, n, damp),) if damped else ())
                _, nm = sv.ldiv_(out, *args)
                outs.append((nm, out.get()))
            # (the norms are block partial sums, and the blocks of the epilogue-only kernel differ from those of the
            #  fused product kernels: same iteration count, last-bit differences in the iterate)
            assert outs[0][0] == outs[1][0] and np.allclose(outs[0][1], outs[1][1], rtol=1e-11, atol=1e-13)
        assert count["mul"] > 2 and count["mulT"] > 2 and count["cs"] >= 1
        with pytest.raises(lsq.LsqError):
            lsq.AllocatedSolver(Op, lsq.QR(), for_lm=False)
        # optimize!: r(x) = A tanh(x) - b with J(x) = A diag(1 - tanh(x)^2) kept matrix-free
        A = rand_csc(m, n, 0.04, 100)
        Ad = lsq.DeviceMatrix(ctx, A)
        b = A @ np.tanh(rng.uniform(-1, 1, n)) + 1e-3 * rng.standard_normal(m)
        state = {"s": np.ones(n)}

        def f_(out, xx):
            out[:] = A @ np.tanh(xx) - b

        def g_stored(Jm, xx):
            sfac = 1 - np.tanh(xx) ** 2
            Jm.data[:] = A.data * np.repeat(sfac, np.diff(A.indptr))

        def g_free(Jop, xx):
            state["s"] = 1 - np.tanh(xx) ** 2
            Jop.refresh()

        def free_mul(trans, xv, out):
            sd = lsq.DeviceVector(ctx, n, state["s"])
            if not trans:    # J x = A (s .* x)
                t = lsq.DeviceVector(ctx, n, xv.get() * state["s"])
                lsq.mul_(out, Ad, t, 1.0, 0.0)
            else:            # J'y = s .* (A'y)
                lsq.mul_(out, Ad, xv, 1.0, 0.0, trans=True)
                out.set(out.get() * state["s"])
            sd.free()

        def free_cs(out):
            out.set(lsq.colsumabs2_(lsq.DeviceVector(ctx, n), Ad).get() * state["s"] ** 2)

        Jfree = lsq.DeviceOperator(ctx, m, n, free_mul, free_cs)
        r_free = lsq.optimize_(lsq.LeastSquaresProblem(x=np.zeros(n), y=np.zeros(m), f_=f_, g_=g_free, J=Jfree),
                               lsq.LevenbergMarquardt(lsq.LSMR()), iterations=30)
        Jst = A.copy()
        r_st = lsq.optimize_(lsq.LeastSquaresProblem(x=np.zeros(n), y=np.zeros(m), f_=f_, g_=g_stored, J=Jst),
                             lsq.LevenbergMarquardt(lsq.LSMR()), iterations=30)
        assert r_free.converged and r_st.converged and r_free.iterations == r_st.iterations
        assert r_free.ssr == pytest.approx(r_st.ssr, rel=1e-9)
        assert np.max(np.abs(r_free.minimizer - r_st.minimizer)) <= 1e-7
    finally:
        lsq.set_exact(None)


# ------------------------------------------------------- serialised comparator (hazard audit)
def _bits_both_ways(run):
    """run() in the normal mode and with every launch waited for (lsq_debug_set(serial=1): no side-stream concurrency, no
    look-ahead, no speculation; same kernels): the two must agree bit for bit."""
    prev = lsq.debug_get()          # (a suite run under LSQ_DEBUG_SERIAL / LSQ_DEBUG_LAUNCH_JITTER goes on in ITS mode afterwards,
    try:                            #  and "normal" means serial = 0 here whatever the environment says: ADVICE r4)
        lsq.debug_set(None, 0)
        normal = run()
        lsq.debug_set(None, 1)
        serial = run()
        lsq.debug_set(None, 0)
        again = run()
    finally:
        lsq.debug_set(prev[0], prev[1])
    return normal, serial, again


def _same(a, b, what):
    for k, (u, v) in enumerate(zip(a, b)):
        if isinstance(u, np.ndarray):
            assert np.array_equal(u, v), (what, k, float(np.abs(u - v).max()))
        else:
            assert u == v, (what, k, u, v)


@pytest.mark.parametrize("config", ["C2", "C3", "C4"])
def test_serial_mode_is_bit_identical(ctx, config):
    """C2-, C3- and C4-shaped solves (BASELINE configs; C3 at a quarter of its rows to keep the comparator short) in the
    normal mode and serialised: identical bits, counts and paths.  Whatever orders two kernels only by luck (queue
    timing, side streams, flags between queued launches) shows up as a difference here."""
    L = lsq._lib
    if config == "C4":
        pr = lsq.synthetic.TanhProblem(1_000_000, 10_000, sparse=True, per_col=1000, seed=77, ctx=ctx)

        def run():
            pr.reset()
            r = pr.optimize(L.LEVENBERG_MARQUARDT, L.LSMR, iterations=5, trace=True)
            return [r.minimizer, r.iterations, r.mul_calls, r.f_calls, r.g_calls, r.ssr, np.array(r.trace["inner"])]
    elif config == "C2":
        pr = lsq.synthetic.TanhProblem(4096, 512, sparse=False, seed=78, ctx=ctx)

        def run():
            pr.reset()
            r = pr.optimize(L.LEVENBERG_MARQUARDT, L.CHOLESKY, iterations=5, trace=True)
            return [r.minimizer, r.iterations, r.mul_calls, r.f_calls, r.g_calls, r.ssr]
    else:
        pr = lsq.synthetic.TanhProblem(4096, 2048, sparse=False, seed=79, ctx=ctx)

        def run():
            pr.reset()
            r = pr.optimize(L.DOGLEG, L.QR, iterations=3, trace=True)
            return [r.minimizer, r.iterations, r.mul_calls, r.f_calls, r.g_calls, r.ssr]
    normal, serial, again = _bits_both_ways(run)
    _same(normal, again, config + ": normal mode twice")
    _same(normal, serial, config + ": normal vs serialised")
    pr.close()


@pytest.mark.parametrize("m,n,solver,for_lm", [(4096, 512, "chol", True), (16384, 2048, "qr", False), (18432 - 2048, 2048, "qr", True),
                                               (2049, 129, "qr", True), (9000, 200, "qr", False), (3000, 500, "chol", True)])
def test_serial_mode_single_solves(ctx, m, n, solver, for_lm):
    """One ldiv! of the dense solvers at the C2 / C3 operand sizes (and the stacked LM operand of C3, and the shape of the
    round-3 failure): normal, serialised (1: same kernels -> same bits) and without the in-kernel exchanges (2: the
    multi-launch fallbacks -> equal to round-off), all against LAPACK."""
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.05
    J = lsq.DeviceMatrix(ctx, A)
    x = lsq.DeviceVector(ctx, n)
    dy = lsq.DeviceVector(ctx, m, y)
    if for_lm:
        G = A.T @ A + np.diag(damp)
        ref = np.linalg.solve(G, A.T @ y)
    else:
        ref = np.linalg.lstsq(A, y, rcond=None)[0]

    def run():
        sv = lsq.AllocatedSolver(J, lsq.QR() if solver == "qr" else lsq.Cholesky(), for_lm=for_lm)
        if for_lm:
            sv.ldiv_(x, dy, lsq.DeviceVector(ctx, n, damp))
        else:
            sv.ldiv_(x, dy)
        got, info = x.get(), sv.info()
        sv.free()
        return got, info
    (x0, i0), (x1, i1), (x2, i2) = _bits_both_ways(run)
    assert np.linalg.norm(x0 - ref) <= 1e-9 * np.linalg.norm(ref), (i0,)
    assert np.array_equal(x0, x2), ("normal mode twice", float(np.abs(x0 - x2).max()), i0)
    assert np.array_equal(x0, x1), ("normal vs serialised", float(np.abs(x0 - x1).max()), i0, i1)
    prev = lsq.debug_get()
    try:
        lsq.debug_set(None, 2)
        x3, i3 = run()
    finally:
        lsq.debug_set(None, prev[1])
    assert np.linalg.norm(x3 - ref) <= 1e-9 * np.linalg.norm(ref), (i3,)
    assert np.linalg.norm(x3 - x0) <= 1e-11 * np.linalg.norm(x0), (i0, i3)
    J.free()


# ------------------------------------------------------- partitioned device
@pytest.fixture(scope="module")
def ctx_cpx(ctx):
    """A context whose launch heuristics see 32 compute units -- one CPX partition of an MI355X (LSQ_DEBUG_NUM_CUS, read by
    lsq_ctx_create): no slab exchange in the QR panel, launch-per-panel Cholesky when the tiles do not fit, fewer workgroups
    and other split factors everywhere.  The device underneath is whatever the box has (>= 32 CUs), so every co-residency
    assumption made for 32 CUs holds."""
    os.environ["LSQ_DEBUG_NUM_CUS"] = "32"
    try:
        c = lsq.Context(ctx.device)
    finally:
        del os.environ["LSQ_DEBUG_NUM_CUS"]
    assert c.device_info()["num_cus"] == 32
    yield c
    c.close()


@pytest.mark.parametrize("m,n", [(2049, 129), (4096, 512), (700, 200), (8193, 65), (3000, 700), (300, 17)])
def test_partitioned_device_dense_solvers(ctx_cpx, m, n):
    """The dense solvers on a 32-CU view of the device (dense_qr.jl:30-88, dense_cholesky.jl:29-59): all four variants against
    LAPACK (full-rank operands) AND against the oracle's own ldiv!, with the paths that were taken reported -- the one-launch
    Cholesky cannot apply at n = 512 / 700 (36 / 66 tiles + chain > 32 CUs), the QR panel runs without slabs."""
    rng = np.random.default_rng(31 * m + n)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.05
    J = lsq.DeviceMatrix(ctx_cpx, A)
    x = lsq.DeviceVector(ctx_cpx, n)
    Jo = O.Mat(dense=A)
    small = m * n <= 3e5                     # (the oracle's scalar QR at the larger shapes: LAPACK stands in)
    for solver, okind in ((lsq.QR(), O.QR), (lsq.Cholesky(), O.CHOLESKY)):
        for for_lm in (False, True):
            sv = lsq.AllocatedSolver(J, solver, for_lm=for_lm)
            if for_lm:
                sv.ldiv_(x, lsq.DeviceVector(ctx_cpx, m, y), lsq.DeviceVector(ctx_cpx, n, damp))
                ref = np.linalg.solve(A.T @ A + np.diag(damp), A.T @ y)
            else:
                sv.ldiv_(x, lsq.DeviceVector(ctx_cpx, m, y))
                ref = np.linalg.lstsq(A, y, rcond=None)[0]
            got, info = x.get(), sv.info()
            assert np.linalg.norm(got - ref) <= 1e-9 * np.linalg.norm(ref), (type(solver).__name__, for_lm, info)
            if small or okind == O.CHOLESKY:
                ro = O.ldiv(okind, Jo, y, damp if for_lm else None)
                assert ro[0] == 0 and np.linalg.norm(got - ro[1]) <= 1e-9 * np.linalg.norm(ro[1]), (type(solver).__name__, for_lm, info)
            if isinstance(solver, lsq.Cholesky) and for_lm and n >= 512:
                assert info.get("chol_path") != "blocked-one-launch", info   # (one resident workgroup per tile needs more than 32 CUs)
            sv.free()
    J.free()
    assert sum(ctx_cpx.fallback_stats().values()) == 0


@pytest.mark.parametrize("opt,sol,sparse", [("lm", "lsmr", True), ("dogleg", "lsmr", True), ("lm", "cholesky", False),
                                            ("dogleg", "qr", False)])
def test_partitioned_device_trajectories(ctx_cpx, opt, sol, sparse):
    """Whole trust-region runs on the 32-CU view against the oracle (levenberg_marquardt.jl:39-144, dogleg.jl:41-203): the sliced
    layouts are built for 32 CUs (other block sizes, other partial-sum groupings), the dense factorisations take their
    partitioned-device branches -- same counts, same inner counts, iterates to 1e-8."""
    m, n, per_col = (300000, 2000, 600) if sparse else (1500, 48, None)
    pr = lsq.synthetic.TanhProblem(m, n, sparse=sparse, per_col=per_col, seed=7, ctx=ctx_cpx)
    pr.reset()
    okind = lsq._lib.LEVENBERG_MARQUARDT if opt == "lm" else lsq._lib.DOGLEG
    skind = {"lsmr": lsq._lib.LSMR, "cholesky": lsq._lib.CHOLESKY, "qr": lsq._lib.QR}[sol]
    rg = pr.optimize(okind, skind, trace=True, iterations=50)
    A = (O.Mat(csc=(m, n, pr.colptr, pr.rowval, pr.A)) if sparse else O.Mat(dense=pr.A.reshape((m, n), order="F")))
    J = (O.Mat(csc=(m, n, pr.colptr, pr.rowval, np.zeros_like(pr.A))) if sparse else O.Mat(dense=np.zeros((m, n))))
    f, g, ud, keep = O.tanh_model(A, pr.b)
    ro = O.optimize(O.LM if opt == "lm" else O.DOGLEG, {"lsmr": O.LSMR, "cholesky": O.CHOLESKY, "qr": O.QR}[sol], J, np.zeros(n),
                    f, g, ud=ud, iterations=50)
    assert rg.iterations == ro.iterations and rg.mul_calls == ro.mul_calls and rg.converged == ro.converged
    assert np.array_equal(rg.trace["inner"], ro.trace["inner"]) and np.array_equal(rg.trace["accept"], ro.trace["accept"])
    for k in range(ro.iterations):
        xr = ro.trace["x"][k]
        assert np.max(np.abs(rg.trace["x"][k] - xr)) <= 1e-8 * max(1.0, np.max(np.abs(xr)))
    pr.close()
