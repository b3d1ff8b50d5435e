"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/lsqhip.h declares, fails loudly without a GPU, the default-selection rules of
types.jl:114-127 hold, and the synthetic generator is deterministic.  No compute calls."""
import numpy as np
import pytest
import scipy.sparse as sp

import lsq_amd as lsq


def test_library_exports_every_declared_symbol():
    L = lsq.lib()
    names = lsq.declared_symbols()
    assert len(names) >= 45
    for s in names:
        assert hasattr(L, s), s
        assert s in L._signatures, "no ctypes prototype for %s" % s
    assert L.lsq_version() >= 100


def test_product_does_not_import_the_oracle():
    import os
    import re
    root = os.path.dirname(lsq._lib.__file__)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|lsq_oracle|orc_", txt, re.M), f


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lsq.HipError):
        lsq.Context(0)


def test_default_selection_rules():
    dense, sparse = np.zeros((3, 3)), sp.csc_matrix(np.ones((3, 3)))
    assert isinstance(lsq.default_solver(None, dense), lsq.QR)
    assert isinstance(lsq.default_solver(None, sparse), lsq.LSMR)
    with pytest.raises(lsq.ArgumentError):
        lsq.default_solver(lsq.QR(), sparse)
    assert isinstance(lsq.default_optimizer(None, lsq.LSMR()), lsq.LevenbergMarquardt)
    assert isinstance(lsq.default_optimizer(None, lsq.QR()), lsq.Dogleg)
    assert isinstance(lsq.default_optimizer(lsq.Dogleg(), lsq.LSMR()), lsq.Dogleg)
    assert isinstance(lsq.default_optimizer(lsq.LevenbergMarquardt(), lsq.Cholesky()).solver, lsq.Cholesky)


def test_problem_constructor_checks():
    f = lambda o, x: None
    with pytest.raises(lsq.DimensionMismatch):
        lsq.LeastSquaresProblem(x=np.zeros(2), y=np.zeros(3), f_=f, J=np.zeros((3, 3)))
    with pytest.raises(lsq.DimensionMismatch):
        lsq.LeastSquaresProblem(x=np.zeros(3), y=np.zeros(2), f_=f, J=np.zeros((3, 3)))
    with pytest.raises(ValueError):
        lsq.LeastSquaresProblem(x=np.zeros(3), f_=f)
    p = lsq.LeastSquaresProblem(x=np.zeros(3), f_=f, J=np.zeros((5, 3)))
    assert len(p.y) == 5  # output_length defaults to size(J, 1) (test/runtests.jl:54-61)
    with pytest.raises(NotImplementedError):
        lsq.LSMR(preconditioner=lambda *a: None, P=1)


def test_synthetic_generator_is_deterministic_and_well_formed():
    m, n, pc = 5000, 40, 100
    a = lsq.synthetic.sparse_inputs(m, n, pc, 11)
    b = lsq.synthetic.sparse_inputs(m, n, pc, 11)
    c = lsq.synthetic.sparse_inputs(m, n, pc, 12)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert not np.array_equal(a[1], c[1])
    colptr, rowval, nzval = a
    assert colptr[-1] == n * pc and np.all(np.diff(colptr) == pc)
    for j in range(n):
        r = rowval[colptr[j]:colptr[j + 1]]
        assert np.all(np.diff(r) > 0) and r[0] >= 0 and r[-1] < m  # sorted, distinct, in range
    assert abs(nzval.std() * np.sqrt(pc) - 1.0) < 0.05
    d = lsq.synthetic.dense_inputs(200, 10, 5)
    assert abs(d.std() * np.sqrt(200) - 1.0) < 0.1
    u = lsq.synthetic.uniform(1000, 3)
    assert u.min() >= -1 and u.max() <= 1 and abs(u.mean()) < 0.1
