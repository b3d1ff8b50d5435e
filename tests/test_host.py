"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/lsqhip.h declares, fails loudly without a GPU, the default-selection rules of
types.jl:114-127 hold, and the synthetic generator is deterministic.  No compute calls."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import lsq_amd as lsq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    with pytest.raises(TypeError):
        lsq.LSMR(preconditioner=lambda *a: None, P=1)        # a general P must provide ldiv(out, x)


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


# ------------------------------------------------------------------ the Julia shim of INTEGRATION.md
def test_julia_shim_matches_the_header():
    """Every ccall in INTEGRATION.md against include/lsqhip.h: symbol declared and exported, return type, arity, number of
    values passed, and the C type class of every argument; the option / result structs field by field."""
    import julia_shim_lint as JL
    protos = JL.header_prototypes()
    assert set(protos) == set(lsq.declared_symbols())              # the lint's header parser sees what the loader sees
    calls = JL.ccalls()
    assert len(calls) >= 40
    L = lsq.lib()
    for name, ret, types, nvalues, line in calls:
        where = "INTEGRATION.md julia line %d: ccall(:%s)" % (line, name)
        assert name in protos and hasattr(L, name), where
        cret, cparams = protos[name]
        assert ret in JL.JULIA_CLASS and cret in JL.JULIA_CLASS[ret], (where, "return", ret, cret)
        assert len(types) == len(cparams), (where, "arity", len(types), len(cparams))
        assert nvalues == len(types), (where, "values passed", nvalues, len(types))
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert jt in JL.JULIA_CLASS, (where, "unknown Julia type", jt)
            assert ct in JL.JULIA_CLASS[jt], (where, "argument %d" % (k + 1), jt, ct)
    # the hot-path entry points are all bound
    bound = {c[0] for c in calls}
    for need in ("lsq_mul", "lsq_colsumabs2", "lsq_rowsumabs2", "lsq_ldiv", "lsq_ldiv_damped", "lsq_solver_create",
                 "lsq_solver_destroy", "lsq_csc_create", "lsq_dense_create", "lsq_mat_set_values", "lsq_mat_set_values_async",
                 "lsq_optimize", "lsq_options_default", "lsq_wdot", "lsq_amax_projected", "lsq_box_clip", "lsq_first_nonfinite"):
        assert need in bound, need
    for jname, cname in (("LsqOptions", "lsq_options"), ("LsqResult", "lsq_result")):
        jf, cf = JL.julia_struct(jname), JL.header_struct(cname)
        assert [f for f, _ in jf] == [f for f, _ in cf], (jname, jf, cf)
        for (f, jt), (_, ct) in zip(jf, cf):
            assert ct in JL.JULIA_CLASS[jt], (jname, f, jt, ct)
    # the RCCL shim's calls (rlib) against include/lsqrccl.h
    rprotos = JL.header_prototypes(JL.RCCL_HEADER)
    rcalls = JL.ccalls("rlib")
    assert {c[0] for c in rcalls} >= {"lsq_rccl_unique_id", "lsq_rccl_comm_create", "lsq_rccl_allreduce_callback"}
    for name, ret, types, nvalues, line in rcalls:
        cret, cparams = rprotos[name]
        assert cret in JL.JULIA_CLASS[ret] and len(types) == len(cparams) == nvalues, (name, ret, types, cparams)
        for jt, ct in zip(types, cparams):
            assert ct in JL.JULIA_CLASS[jt], (name, jt, ct)
    # the ctypes mirror (what the tests actually drive) has the same layout
    assert [f for f, _ in lsq._lib.Options._fields_] == [f for f, _ in JL.header_struct("lsq_options")]
    assert [f for f, _ in lsq._lib.Result._fields_] == [f for f, _ in JL.header_struct("lsq_result")]


def test_julia_shim_defines_what_the_reference_loops_call():
    """The two load-time defects a reader found in round 2 cannot come back: (a) `J'` needs Base.adjoint on the handle types
    (levenberg_marquardt.jl:102, dogleg.jl:99); (b) AbstractAllocatedSolver methods must not be ambiguous with the reference's
    (iterative_lsmr.jl:173,233; dense_qr.jl:25,50; dense_cholesky.jl:19): argument 1 narrower, argument 2 identical."""
    import re
    import julia_shim_lint as JL
    src = JL.julia_blocks()
    for pattern, why in JL.REQUIRED_METHODS:
        assert re.search(pattern, src), "shim lacks a method matching %r (needed by %s)" % (pattern, why)
    assert not re.search(r"Adjoint\{[^}]*Hip", src), "LinearAlgebra.Adjoint of a non-AbstractMatrix handle"
    methods = JL.allocated_solver_methods()
    assert {a2 for _, a2 in methods} == set(JL.REFERENCE_SOLVER_METHODS), methods
    for a1, a2 in methods:
        assert a1 == "HipProblem", (a1, a2)         # narrower than the reference's `nls::LeastSquaresProblem{...}`
    m = re.search(r"const HipProblem = LeastSquaresProblem\{(.*)\}", src)
    assert m and m.group(1).count(",") == 4          # LeastSquaresProblem{Tx, Ty, Tf, TJ, Tg}: types.jl:7
    assert "<:HipVector,<:HipVector,<:Any,<:HipJacobian" in m.group(1)


def _build_abi_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_demo")
    libdir = os.path.join(ROOT, "leastsquaresoptim.jl_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_demo.c"), "-o", exe, "-L", libdir, "-llsqhip", "-lm",
                           "-Wl,-rpath," + libdir])
    return exe


def test_header_is_plain_c_and_agrees_with_the_library(tmp_path):
    """The boundary is a C ABI: include/lsqhip.h and include/lsqrccl.h compile as C99 (-pedantic -Werror) in a plain C
    program that links liblsqhip.so, reads the reference's default options through it, and sees the same struct sizes as
    the ctypes mirrors the Python host side uses (no device needed)."""
    import ctypes as C
    import subprocess
    import lsq_amd
    lsq_amd.build()
    exe = _build_abi_demo(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    sizes = dict(l.rsplit(" ", 1) for l in out.stdout.strip().splitlines())
    assert int(sizes["sizeof lsq_options"]) == C.sizeof(lsq_amd._lib.Options)
    assert int(sizes["sizeof lsq_result"]) == C.sizeof(lsq_amd._lib.Result)
    assert int(sizes["lsq_version"]) == lsq_amd.lib().lsq_version()


@pytest.mark.gpu
def test_c_program_solves_through_the_abi(tmp_path):
    """The same C program on the device: lsq_ldiv of QR, Cholesky and LSMR on a small dense problem, checked by the
    normal equations -- the boundary used from C, with no Python in the data path."""
    import subprocess
    exe = _build_abi_demo(tmp_path)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.count("max|J'(Jx - y)|") == 3
