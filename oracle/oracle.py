"""ctypes front-end for the CPU oracle (oracle/lsq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package.  Parity pinning statement: see lsq_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

QR, CHOLESKY, LSMR = 0, 1, 2
DOGLEG, LM = 0, 1
OK, EDIM, ENOTPD, ERANK, ENONFINITE, EBOUNDS = range(6)

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class OrcMat(C.Structure):
    _fields_ = [("kind", C.c_int), ("m", C.c_int), ("n", C.c_int), ("val", c_dp),
                ("colptr", c_ip), ("rowval", c_ip)]


class OrcOptions(C.Structure):
    _fields_ = [("x_tol", C.c_double), ("f_tol", C.c_double), ("g_tol", C.c_double),
                ("iterations", C.c_int), ("delta", C.c_double), ("lower", c_dp), ("upper", c_dp),
                ("trace_cap", C.c_int), ("trace_ssr", c_dp), ("trace_gnorm", c_dp),
                ("trace_delta", c_dp), ("trace_rho", c_dp), ("trace_inner", c_ip),
                ("trace_accept", c_ip), ("trace_x", c_dp)]


class OrcResult(C.Structure):
    _fields_ = [("optimizer", C.c_int), ("ssr", C.c_double), ("iterations", C.c_int),
                ("converged", C.c_int), ("x_converged", C.c_int), ("f_converged", C.c_int),
                ("g_converged", C.c_int), ("f_calls", C.c_int), ("g_calls", C.c_int),
                ("mul_calls", C.c_int), ("status", C.c_int), ("bad_index", C.c_int)]


class OrcTanhModel(C.Structure):
    _fields_ = [("A", C.POINTER(OrcMat)), ("b", c_dp), ("t", c_dp), ("J", C.POINTER(OrcMat)),
                ("threads", C.c_int)]


F_CB = C.CFUNCTYPE(None, c_dp, c_dp, C.c_void_p)
G_CB = C.CFUNCTYPE(None, c_dp, c_dp, C.c_void_p)


def build():
    """Compile oracle/liblsq_oracle.so with gcc (building the checker is not using it)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblsq_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_wdot.restype = C.c_double
        L.orc_wnorm.restype = C.c_double
        L.orc_maxabs_projected_gradient.restype = C.c_double
        L.orc_lsmr.restype = C.c_int
        L.orc_lsmr.argtypes = [c_dp, C.POINTER(OrcMat), c_dp, c_dp, c_dp, C.c_double, C.c_double,
                               C.c_double, C.c_int, c_ip, c_dp, c_dp]
        L.orc_pstrf_upper.argtypes = [c_dp, C.c_int, c_ip, c_ip, C.c_double]
        L.orc_qrp_solve.argtypes = [c_dp, C.c_int, C.c_int, c_ip, c_dp, c_dp, C.c_int, C.c_double]
        L.orc_optimize.argtypes = [C.c_int, C.c_int, C.POINTER(OrcMat), c_dp, c_dp, F_CB, G_CB,
                                   C.c_void_p, C.POINTER(OrcOptions), C.POINTER(OrcResult)]
        _LIB = L
    return _LIB


def set_sum_mode(mode):
    """Summation-order model of the stdlib reductions (see lsq_oracle.c): 0 = index order (default)."""
    lib().orc_set_sum_mode(int(mode))


SUM_MODES = {0: "index order", 1: "mapreduce, no SIMD", 2: "4 SIMD lanes", 3: "8 SIMD lanes", 4: "16 SIMD lanes",
             5: "16 SIMD lanes + extended-precision nrm2"}
# mode 6 (64-lane trees at EVERY reduction, sparse products and wdot included) is not a model of the reference but of
# the class of orders the HIP fast path uses; see lsq_oracle.c
FAST_PATH_MODE = 6
RANDOM_ORDER_MODES = list(range(100, 116))       # every reduction adds its terms in a random order
ROUNDING_NOISE_MODES = list(range(1000, 1032))    # index order, every reduction result moved by one rounding error


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Mat:
    """Dense (column-major) or CSC matrix view that keeps its numpy buffers alive."""

    def __init__(self, dense=None, csc=None):
        if dense is not None:
            a = np.asarray(dense, dtype=np.float64)
            self.m, self.n = a.shape
            self.val = np.asfortranarray(a).reshape(-1, order="F").copy()
            self.colptr = self.rowval = None
            self.kind = 0
        else:
            m, n, colptr, rowval, nzval = csc
            self.m, self.n = int(m), int(n)
            self.colptr = np.ascontiguousarray(colptr, dtype=np.int32)
            self.rowval = np.ascontiguousarray(rowval, dtype=np.int32)
            self.val = f64(nzval).copy()
            self.kind = 1
        self.c = OrcMat(self.kind, self.m, self.n, _dp(self.val), _ip(self.colptr), _ip(self.rowval))

    @classmethod
    def from_scipy(cls, S):
        S = S.tocsc()
        S.sort_indices()
        return cls(csc=(S.shape[0], S.shape[1], S.indptr, S.indices, S.data))

    def dense(self):
        if self.kind == 0:
            return self.val.reshape((self.m, self.n), order="F").copy()
        out = np.zeros((self.m, self.n))
        for j in range(self.n):
            for k in range(self.colptr[j], self.colptr[j + 1]):
                out[self.rowval[k], j] = self.val[k]
        return out

    @property
    def ref(self):
        return C.byref(self.c)


def colsumabs2(A):
    v = np.zeros(A.n)
    lib().orc_colsumabs2(_dp(v), A.ref)
    return v


def rowsumabs2(A):
    v = np.zeros(A.m)
    lib().orc_rowsumabs2(_dp(v), A.ref)
    return v


def mul(A, x, alpha=1.0, beta=0.0, y=None):
    y = np.zeros(A.m) if y is None else f64(y).copy()
    x = f64(x)
    lib().orc_mul(_dp(y), A.ref, _dp(x), C.c_double(alpha), C.c_double(beta))
    return y


def mulT(A, y, alpha=1.0, beta=0.0, x=None):
    x = np.zeros(A.n) if x is None else f64(x).copy()
    y = f64(y)
    lib().orc_mulT(_dp(x), A.ref, _dp(y), C.c_double(alpha), C.c_double(beta))
    return x


def wdot(x, y, w):
    x, y, w = f64(x), f64(y), f64(w)
    return lib().orc_wdot(_dp(x), _dp(y), _dp(w), len(x))


def maxabs_projected_gradient(g, x, lower=None, upper=None):
    g, x = f64(g), f64(x)
    lo = None if lower is None else f64(lower)
    hi = None if upper is None else f64(upper)
    return lib().orc_maxabs_projected_gradient(_dp(g), _dp(x), _dp(lo), _dp(hi), len(g))


def lsmr(J, b, diag=None, P=None, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=-1):
    """Raw lsmr! on [J; diag(diag)] diag(P), zero start. Returns dict(x, iter, istop, normr, normAr)."""
    x = np.zeros(J.n)
    by = f64(b).copy()
    d = None if diag is None else f64(diag)
    p = None if P is None else f64(P)
    istop = C.c_int(0)
    nr, nar = C.c_double(0), C.c_double(0)
    it = lib().orc_lsmr(_dp(x), J.ref, _dp(d), _dp(p), _dp(by), atol, btol, conlim, maxiter,
                        C.byref(istop), C.byref(nr), C.byref(nar))
    return dict(x=x, iter=it, istop=istop.value, normr=nr.value, normAr=nar.value)


def ldiv(solver, J, y, damp=None):
    """The L2 boundary: returns (status, x, nmul[, damp_after])."""
    x = np.zeros(J.n)
    y = f64(y)
    nmul = C.c_int(0)
    L = lib()
    if damp is None:
        if solver == LSMR:
            st = L.orc_ldiv_lsmr(_dp(x), J.ref, _dp(y), C.byref(nmul))
        elif solver == CHOLESKY:
            st = L.orc_ldiv_cholesky(_dp(x), J.ref, _dp(y), C.byref(nmul))
        else:
            st = L.orc_ldiv_qr(_dp(x), J.ref, _dp(y), C.byref(nmul), None)
        return st, x, nmul.value
    d = f64(damp).copy()
    if solver == LSMR:
        st = L.orc_ldiv_lsmr_damped(_dp(x), J.ref, _dp(y), _dp(d), C.byref(nmul))
    elif solver == CHOLESKY:
        st = L.orc_ldiv_cholesky_damped(_dp(x), J.ref, _dp(y), _dp(d), C.byref(nmul))
    else:
        st = L.orc_ldiv_qr_damped(_dp(x), J.ref, _dp(y), _dp(d), C.byref(nmul), None)
    return st, x, nmul.value, d


def qr_solve(A, b, rcond=None):
    """geqp3 + rank-revealing min-norm solve on a dense numpy matrix; returns (x, rank, jpvt)."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    a = np.asfortranarray(A).reshape(-1, order="F").copy()
    jp = np.zeros(n, dtype=np.int32)
    tau = np.zeros(max(min(m, n), 1))
    lu = max(m, n)
    u = np.zeros(lu)
    u[:m] = b
    L = lib()
    L.orc_geqp3(_dp(a), m, n, _ip(jp), _dp(tau))
    rc = min(m, n) * np.finfo(float).eps if rcond is None else rcond
    rank = L.orc_qrp_solve(_dp(a), m, n, _ip(jp), _dp(tau), _dp(u), lu, C.c_double(rc))
    return u[:n].copy(), rank, jp, a.reshape((m, n), order="F"), tau


GEQP3_CB = C.CFUNCTYPE(None, c_dp, C.c_int, C.c_int, c_ip, c_dp)
_geqp3_keep = None


def use_lapack_geqp3(on=True):
    """Serve the oracle's geqp3 (inside orc_ldiv_qr / orc_ldiv_qr_damped and so inside optimize(.., QR, ..)) from scipy's
    LAPACK dgeqp3 -- the routine Julia's qr!(A, ColumnNorm()) dispatches to (dense_qr.jl:37,83) -- for shapes where the
    scalar restatement needs minutes (C3: 16384 x 2048).  Everything after the factorisation stays the oracle's."""
    global _geqp3_keep
    if not on:
        lib().orc_set_geqp3_backend(C.cast(None, GEQP3_CB))
        _geqp3_keep = None
        return
    from scipy.linalg import lapack

    def _cb(ap, m, n, jp, taup):
        a = np.ctypeslib.as_array(ap, (m * n,)).reshape((m, n), order="F")
        qr, jpvt, tau, _, info = lapack.dgeqp3(a, overwrite_a=0)
        assert info == 0
        a[:, :] = qr
        np.ctypeslib.as_array(jp, (n,))[:] = jpvt - 1
        k = min(m, n)
        if k:
            np.ctypeslib.as_array(taup, (k,))[:] = tau[:k]

    _geqp3_keep = GEQP3_CB(_cb)
    lib().orc_set_geqp3_backend(_geqp3_keep)


def potrf_upper(A):
    a = np.asfortranarray(np.asarray(A, dtype=np.float64)).reshape(-1, order="F").copy()
    n = A.shape[0]
    info = lib().orc_potrf_upper(_dp(a), n)
    return info, np.triu(a.reshape((n, n), order="F"))


def pstrf_upper(A, tol=0.0):
    a = np.asfortranarray(np.asarray(A, dtype=np.float64)).reshape(-1, order="F").copy()
    n = A.shape[0]
    piv = np.zeros(n, dtype=np.int32)
    rank = C.c_int(0)
    info = lib().orc_pstrf_upper(_dp(a), n, _ip(piv), C.byref(rank), tol)
    return info, np.triu(a.reshape((n, n), order="F")), piv, rank.value


class Result:
    pass


def optimize(optimizer, solver, J, x0, f, g, m=None, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8,
             iterations=1000, delta=-1.0, lower=None, upper=None, trace=True, trace_x=True,
             ud=None):
    """Run the oracle's restatement of optimize!.

    f(out, x) / g(Jval, x) are Python callables on numpy views (Jval is the flat column-major dense
    buffer or nzval), or pre-built ctypes callbacks (F_CB/G_CB) with `ud` for the built-in model.
    """
    n = J.n
    m = J.m if m is None else m
    x = f64(x0).copy()
    fcur = np.zeros(m)

    if isinstance(f, F_CB):
        fcb, gcb = f, g
    else:
        def _f(outp, xp, _):
            f(np.ctypeslib.as_array(outp, (m,)), np.ctypeslib.as_array(xp, (n,)))

        def _g(jp, xp, _):
            g(np.ctypeslib.as_array(jp, (len(J.val),)), np.ctypeslib.as_array(xp, (n,)))

        fcb, gcb = F_CB(_f), G_CB(_g)

    cap = iterations if trace else 0
    tr = dict(ssr=np.zeros(cap), gnorm=np.zeros(cap), delta=np.zeros(cap), rho=np.zeros(cap),
              inner=np.zeros(cap, dtype=np.int32), accept=np.zeros(cap, dtype=np.int32),
              x=np.zeros((cap, n)) if (trace and trace_x) else None)
    lo = None if lower is None or len(lower) == 0 else f64(lower)
    hi = None if upper is None or len(upper) == 0 else f64(upper)
    opt = OrcOptions(x_tol, f_tol, g_tol, iterations, delta, _dp(lo), _dp(hi), cap,
                     _dp(tr["ssr"]) if cap else None, _dp(tr["gnorm"]) if cap else None,
                     _dp(tr["delta"]) if cap else None, _dp(tr["rho"]) if cap else None,
                     _ip(tr["inner"]) if cap else None, _ip(tr["accept"]) if cap else None,
                     _dp(tr["x"]) if tr["x"] is not None else None)
    res = OrcResult()
    lib().orc_optimize(optimizer, solver, J.ref, _dp(x), _dp(fcur), fcb, gcb, ud, C.byref(opt),
                       C.byref(res))
    r = Result()
    r.status = res.status
    r.optimizer = "LevenbergMarquardt" if optimizer == LM else "Dogleg"
    r.minimizer, r.fcur = x, fcur
    r.ssr, r.iterations = res.ssr, res.iterations
    r.converged, r.x_converged = bool(res.converged), bool(res.x_converged)
    r.f_converged, r.g_converged = bool(res.f_converged), bool(res.g_converged)
    r.f_calls, r.g_calls, r.mul_calls = res.f_calls, res.g_calls, res.mul_calls
    r.bad_index = res.bad_index
    k = res.iterations
    r.trace = {key: (v[:k].copy() if v is not None else None) for key, v in tr.items()} if trace else None
    return r


def tanh_model(A, b):
    """Built-in synthetic model callbacks (pure C, no Python in the loop). Returns (f, g, ud, keep)."""
    L = lib()
    t = np.zeros(A.n)
    b = f64(b)
    md = OrcTanhModel(C.pointer(A.c), _dp(b), _dp(t), None, 1)
    f = C.cast(L.orc_tanh_f, F_CB)
    g = C.cast(L.orc_tanh_g, G_CB)
    return f, g, C.cast(C.pointer(md), C.c_void_p), (md, t, b, A)


_OMP = None


def _omp():
    global _OMP
    if _OMP is None:
        # (thread placement must be decided before the OpenMP runtime starts)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
        path = os.path.join(_HERE, "liblsq_oracle_omp.so")
        if not os.path.exists(path):
            build()
        _OMP = C.CDLL(path)
        _OMP.orc_omp_create.restype = C.c_void_p
        _OMP.orc_omp_create.argtypes = [C.c_int, C.c_int, c_ip, c_ip, c_dp, c_dp, C.c_int]
        _OMP.orc_omp_run.argtypes = [C.c_void_p, c_dp, C.c_int, C.POINTER(C.c_longlong), c_dp]
        _OMP.orc_omp_destroy.argtypes = [C.c_void_p]
    return _OMP


class OmpProblem:
    """The all-cores OpenMP variant of the LM+LSMR path on the sparse tanh model (oracle/lsq_oracle_omp.c): CSR mirror and
    work arrays built once (first-touched in parallel), then `run(x0, iterations)` = that many outer iterations with zero
    tolerances.  A measurement baseline (bench.py), not a parity oracle: its reductions are OpenMP reductions."""

    def __init__(self, m, n, colptr, rowval, nzval, b, threads=0):
        L = _omp()
        self._keep = (np.ascontiguousarray(colptr, dtype=np.int32), np.ascontiguousarray(rowval, dtype=np.int32),
                      f64(nzval), f64(b))
        self.n = n
        self.threads = threads if threads > 0 else L.orc_omp_max_threads()
        self.h = L.orc_omp_create(m, n, _ip(self._keep[0]), _ip(self._keep[1]), _dp(self._keep[2]), _dp(self._keep[3]), threads)

    def run(self, x0, iterations):
        x = f64(x0).copy()
        inner, ssr = C.c_longlong(0), C.c_double(0)
        _omp().orc_omp_run(self.h, _dp(x), iterations, C.byref(inner), C.byref(ssr))
        return x, ssr.value, inner.value

    def close(self):
        if self.h:
            _omp().orc_omp_destroy(self.h)
            self.h = None


def lm_lsmr_omp(m, n, colptr, rowval, nzval, b, x0, iterations, threads=0):
    """One-shot form: returns (x, ssr, inner iterations, threads used)."""
    pr = OmpProblem(m, n, colptr, rowval, nzval, b, threads)
    try:
        x, ssr, inner = pr.run(x0, iterations)
        return x, ssr, inner, pr.threads
    finally:
        pr.close()
