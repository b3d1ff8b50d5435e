"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
inputs.  Tolerances (fp64; summation order differs: tree reductions vs the serial loops of the
reference):
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
import os

import numpy as np
import pytest
import scipy.sparse as sp

import problems as P
from oracle import oracle as O

pytestmark = pytest.mark.gpu

lsq = pytest.importorskip("lsq_amd")


@pytest.fixture(scope="module")
def ctx():
    return lsq.default_context()


def rand_csc(m, n, density, seed):
    rng = np.random.default_rng(seed)
    S = sp.random(m, n, density=density, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    S.sort_indices()
    return S


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


@pytest.mark.parametrize("m,n,rank,solver", [(700, 200, 1, "qr"), (3000, 130, 130, "qr"), (9000, 200, 200, "qr"),
                                             (700, 200, 200, "chol"), (3000, 500, 500, "chol")])
def test_dense_solves_are_repeatable(ctx, m, n, rank, solver, monkeypatch):
    """Every reduction of the dense factorisations runs in a fixed order (block reductions, slab exchanges,
    split-K slices, pipelined block solves), so repeated solves must agree BIT FOR BIT -- which also makes this
    the detector for races between workgroups or between the rounds of a panel launch (an unsynchronised
    hand-off shows up as a result that changes from run to run)."""
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n)) if rank < n else rng.standard_normal((m, n))
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    cases = [()] if solver == "chol" else [("LSQ_QR_TWO_STAGE",), ("LSQ_QR_TWO_STAGE", "LSQ_QR_ALWAYS_PIVOT")]
    for envs in cases:
        for env in envs:
            monkeypatch.setenv(env, "1")
        first = None
        for rep in range(12):
            if solver == "chol":
                sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
                sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
            else:
                sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
                sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
            x = dxo.get()
            if first is None:
                first = x
            assert np.array_equal(x, first), (envs, rep, np.abs(x - first).max())
        for env in envs:
            monkeypatch.delenv(env)


def test_dense_shape_sweep(ctx):
    """The dense launch sequence depends on the shape in many ways (one-workgroup / two-stage / TSQR levels, 1-256
    row slabs, ragged panels, pair kernel vs MFMA tiles for J'J): shapes around every threshold, all four
    solver variants, against LAPACK through numpy (full-rank operands: the solution is unique, so any
    stable method is the oracle here; tools/dense_fuzz.py is the long version)."""
    rng = np.random.default_rng(2026)
    ms = [16, 17, 63, 65, 129, 257, 1000, 2047, 2049, 4097, 8193, 16385, 20481, 32769, 50000, 131073]
    ns = [1, 2, 3, 8, 9, 13, 15, 16, 17, 21, 25, 29, 32, 33, 63, 64, 65, 100, 129, 200]
    shapes = set()
    while len(shapes) < 70:
        n = int(rng.choice(ns))
        m = max(int(rng.choice(ms)), n + int(rng.integers(0, 40)))
        if m * n <= 6e6:
            shapes.add((m, n))
    for m, n in sorted(shapes):
        A = rng.standard_normal((m, n)) / np.sqrt(m)
        y = rng.standard_normal(m)
        damp = rng.random(n) + 0.05
        J = lsq.DeviceMatrix(ctx, A)
        x = lsq.DeviceVector(ctx, n)
        ref0 = np.linalg.lstsq(A, y, rcond=None)[0]
        refd = np.linalg.solve(A.T @ A + np.diag(damp), A.T @ y)
        cond = np.linalg.cond(A) if m * n <= 2e5 else 10.0
        for solver in (lsq.QR(), lsq.Cholesky()):
            for for_lm in (False, True):
                sv = lsq.AllocatedSolver(J, solver, for_lm=for_lm)
                if for_lm:
                    sv.ldiv_(x, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
                else:
                    sv.ldiv_(x, lsq.DeviceVector(ctx, m, y))
                ref = refd if for_lm else ref0
                tol = 1e-9 * max(1.0, cond * cond if (isinstance(solver, lsq.Cholesky) and not for_lm) else cond)
                err = np.linalg.norm(x.get() - ref) / np.linalg.norm(ref)
                assert np.isfinite(err) and err <= tol, (m, n, type(solver).__name__, for_lm, err, sv.info())
                sv.free()
        J.free()


def test_dense_exchange_timeout_falls_back(ctx, monkeypatch):
    """The in-kernel exchanges (row slabs of the QR panel steps, pipelined block solves) wait with a bound; when a wait
    gives up, the same synchronisation that carries the solver's decision reports it and the solve is repeated
    without exchanges -- from then on for that solver.  LSQ_TEST_EXCHANGE_TIMEOUT makes the library pretend (and
    spoil the result the way a real timeout would): answers must still be the oracle's."""
    monkeypatch.setenv("LSQ_TEST_EXCHANGE_TIMEOUT", "1")
    rng = np.random.default_rng(99)
    m, n = 9000, 200                       # slabs in the panel steps, 4 blocks in the solves
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y = rng.standard_normal(m)
    damp = rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    dxo = lsq.DeviceVector(ctx, n)
    xr, rk, *_ = O.qr_solve(A, y)
    sv = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    for k in range(2):                     # first solve: timeout + retry; second: exchanges paused
        sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        assert sv.info()["qr_rank"] == rk == n
        assert np.allclose(dxo.get(), xr, rtol=1e-9, atol=1e-12)
        st = sv.stats()["qr_exchange"]     # counted once, and paused (16 solves) rather than switched off for good
        assert st["giveups"] == 1 and st["paused"] == 16 - (k + 1), st
    for for_lm in (True, False):
        svc = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=for_lm)
        for _ in range(2):
            if for_lm:
                svc.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
                xc = O.ldiv(O.CHOLESKY, O.Mat(dense=A), y, damp)[1]
                assert svc.info()["chol_path"] == "blocked"      # the one-launch factorisation gave up and is paused
                assert svc.stats()["chol_one_launch"]["giveups"] == 1 and svc.stats()["chol_one_launch"]["paused"] > 0
                assert svc.stats()["tri_pipeline"]["giveups"] == 1
            else:
                svc.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
                xc = O.ldiv(O.CHOLESKY, O.Mat(dense=A), y)[1]
            assert np.allclose(dxo.get(), xc, rtol=1e-9, atol=1e-12), for_lm


def test_fast_paths_are_rearmed_after_a_pause(ctx, monkeypatch):
    """A give-up pauses a co-residency fast path for 16 solves, then it is armed again (no one-way latches): with
    the fault injector on for the first solve only, solve 1 falls back, solves 2..17 run the launch-per-panel path, solve 18
    is the one-launch factorisation again -- and every answer is the oracle's."""
    rng = np.random.default_rng(5)
    m, n = 4096, 512
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y, damp = rng.standard_normal(m), rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    sv = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
    xc = O.ldiv(O.CHOLESKY, O.Mat(dense=A), y, damp)[1]
    dxo = lsq.DeviceVector(ctx, n)
    paths = []
    for k in range(19):
        if k == 0:
            monkeypatch.setenv("LSQ_TEST_EXCHANGE_TIMEOUT", "1")
        sv.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        monkeypatch.delenv("LSQ_TEST_EXCHANGE_TIMEOUT", raising=False)
        assert np.allclose(dxo.get(), xc, rtol=1e-9, atol=1e-12), k
        paths.append(sv.info()["chol_path"])
    assert paths[0] == "blocked" and set(paths[1:16]) == {"blocked"}
    assert paths[-1] == "blocked-one-launch", paths
    st = sv.stats()
    assert st["chol_one_launch"] == {"giveups": 1, "paused": 0}
    assert ctx.fallback_stats()["chol_one_launch"] >= 1


def test_dense_solves_next_to_a_busy_neighbour(ctx):
    """The situation a sharded run creates (an RCCL kernel, or any other tenant, holding CUs while the solvers' one-launch /
    pipelined paths assume their workgroups are co-resident): C2-sized Cholesky and a QR solve while a second stream keeps
    224 workgroups x 96 KB of LDS busy for 30 ms at a time.  Results must be the oracle's whatever the fast paths decide;
    how often they gave up is reported (and bounded: a give-up pauses the path)."""
    rng = np.random.default_rng(6)
    m, n = 4096, 512
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    y, damp = rng.standard_normal(m), rng.random(n) + 0.01
    J = lsq.DeviceMatrix(ctx, A)
    svc = lsq.AllocatedSolver(J, lsq.Cholesky(), for_lm=True)
    svq = lsq.AllocatedSolver(J, lsq.QR(), for_lm=False)
    xc = O.ldiv(O.CHOLESKY, O.Mat(dense=A), y, damp)[1]
    xq = O.qr_solve(A, y)[0]
    dxo = lsq.DeviceVector(ctx, n)
    before = ctx.fallback_stats()
    for k in range(6):
        ctx.occupy(224, 96 * 1024, 30.0)
        svc.ldiv_(dxo, lsq.DeviceVector(ctx, m, y), lsq.DeviceVector(ctx, n, damp))
        assert np.allclose(dxo.get(), xc, rtol=1e-9, atol=1e-12), k
        svq.ldiv_(dxo, lsq.DeviceVector(ctx, m, y))
        assert np.allclose(dxo.get(), xq, rtol=1e-9, atol=1e-12), k
        ctx.occupy_wait()
    after = ctx.fallback_stats()
    fired = {k: after[k] - before[k] for k in after}
    print("fallbacks fired next to a busy neighbour:", fired, svc.stats(), svq.stats())
    assert all(v <= 2 for v in fired.values()), fired          # a path that gave up is paused, not retried every solve


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


def test_kat_trajectories():
    """SURVEY 8c KAT-DL / KAT-LM through the HIP path."""
    r = gpu_run(P.readme_rosenbrock(), lsq.Dogleg, lsq.QR(), iterations=2)
    assert r.trace["rho"][0] == pytest.approx(-9999.0, rel=1e-12)
    assert r.trace["rho"][1] == pytest.approx(-624.25 / 0.75, rel=1e-12)
    assert list(r.trace["delta"]) == [0.5, 0.25] and np.all(r.trace["x"] == 0)
    r = gpu_run(P.readme_rosenbrock(), lsq.LevenbergMarquardt, lsq.QR(), iterations=1)
    assert r.trace["rho"][0] == pytest.approx(-6886.0523416, rel=1e-9) and r.trace["delta"][0] == 5.0


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
def test_factor_model(opt):
    """test/nonlinearleastsquares.jl:96-110 (rank-deficient J'J: pins the min-norm QR solve)."""
    name, f, g, x0 = P.factor_dense()
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.ones(9), f_=f, g_=g, J=np.ones((9, 6)))
    r = lsq.optimize_(nls, OPT[opt][0](lsq.QR()), full_trace=True)
    ff, gg = P.wrap_dense(f, g, 9, 6)
    ro = O.optimize(OPT[opt][1], O.QR, O.Mat(dense=np.zeros((9, 6))), x0, ff, gg)
    assert r.converged and r.ssr <= 12
    assert r.iterations == ro.iterations and np.allclose(r.minimizer, ro.minimizer, rtol=1e-6, atol=1e-8)
    name, f, gs, x0, (m, n, colptr, rowval) = P.factor_sparse()
    J = sp.csc_matrix((np.ones(18), rowval, colptr), shape=(9, 6))
    nls = lsq.LeastSquaresProblem(x=x0.copy(), y=np.ones(9), f_=f, g_=lambda Jm, x: gs(Jm.data, x), J=J)
    r = lsq.optimize_(nls, OPT[opt][0](lsq.LSMR()), full_trace=True)
    ro = O.optimize(OPT[opt][1], O.LSMR, O.Mat(csc=(m, n, colptr, rowval, np.zeros(18))), x0, f, gs)
    assert r.converged and r.ssr <= 12
    assert r.iterations == ro.iterations and r.mul_calls == ro.mul_calls


@pytest.mark.parametrize("opt", ["dogleg", "lm"])
def test_bounds(opt):
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


def test_finite_difference_and_defaults():
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
