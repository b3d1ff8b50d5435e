"""Synthetic benchmark problems of SURVEY.md 8(d): r(x) = A tanh(x) - b, J = A diag(1 - tanh(x)^2).

Inputs come from the counter-based generator inside liblsqhip.so (host code), so the CPU baseline
and every GPU rank regenerate bit-identical problems from the seed alone.  f!/g! run on the device
(lsq_model_*), so nothing crosses PCIe inside the timed loop.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .api import (DeviceVector, LeastSquaresResult, _run_native, default_context)

BASE_SEED = 20260928


def sparse_inputs(m, n, per_col, seed):
    """CSC pattern/values with exactly per_col sorted distinct rows per column (nnz = n*per_col)."""
    nnz = n * per_col
    colptr = np.zeros(n + 1, dtype=np.int32)
    rowval = np.zeros(nnz, dtype=np.int32)
    nzval = np.zeros(nnz)
    check(lib().lsq_synth_sparse(m, n, per_col, seed, colptr.ctypes.data_as(_lib.c_ip),
                                 rowval.ctypes.data_as(_lib.c_ip), nzval.ctypes.data_as(_lib.c_dp)))
    return colptr, rowval, nzval


def dense_inputs(m, n, seed):
    v = np.zeros(m * n)
    check(lib().lsq_synth_dense(m, n, seed, v.ctypes.data_as(_lib.c_dp)))
    return v


def uniform(n, seed, lo=-1.0, hi=1.0):
    v = np.zeros(n)
    check(lib().lsq_synth_uniform(n, seed, lo, hi, v.ctypes.data_as(_lib.c_dp)))
    return v


def normal(n, seed):
    v = np.zeros(n)
    check(lib().lsq_synth_normal(n, seed, v.ctypes.data_as(_lib.c_dp)))
    return v


def rhs_for(matvec, m, n, seed, noise=1e-3):
    """b = A tanh(x_true) + noise * N(0,1), x_true ~ U(-1,1)."""
    x_true = uniform(n, seed + 101)
    b = matvec(np.tanh(x_true)) + noise * normal(m, seed + 202)
    return x_true, b


def csc_matvec(m, colptr, rowval, nzval, t):
    """Host A*t for building b (numpy, vectorised)."""
    n = len(colptr) - 1
    cols = np.repeat(np.arange(n), np.diff(colptr))
    return np.bincount(rowval, weights=nzval * t[cols], minlength=m)


class _Handle:      # minimal handle wrapper for _run_native
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h


class TanhProblem:
    """Device-resident problem: Jacobian handle + model (A, b) + x / fcur vectors."""

    def __init__(self, m, n, sparse=True, per_col=None, seed=BASE_SEED, ctx=None, inputs=None, b=None):
        self.ctx = ctx or default_context()
        self.m, self.n, self.sparse = m, n, sparse
        L = lib()
        h = C.c_void_p()
        if sparse:
            if inputs is None:
                inputs = sparse_inputs(m, n, per_col, seed)
            colptr, rowval, nzval = inputs
            self.colptr, self.rowval, self.A = colptr, rowval, nzval
            check(L.lsq_csc_create(self.ctx.h, m, n, colptr.ctypes.data_as(_lib.c_ip),
                                   rowval.ctypes.data_as(_lib.c_ip), C.byref(h)))
            self.nnz = len(nzval)
            mv = lambda t: csc_matvec(m, colptr, rowval, nzval, t)
        else:
            if inputs is None:
                inputs = dense_inputs(m, n, seed)
            self.A = inputs
            check(L.lsq_dense_create(self.ctx.h, m, n, C.byref(h)))
            self.nnz = m * n
            Amat = self.A.reshape((m, n), order="F")
            mv = lambda t: Amat @ t
        self.J = h
        if b is None:
            self.x_true, self.b = rhs_for(mv, m, n, seed)
        else:                       # (a row block of a larger problem: the caller owns the right-hand side)
            self.x_true, self.b = None, np.ascontiguousarray(b, dtype=np.float64)
        md = C.c_void_p()
        check(L.lsq_model_tanh_create(self.ctx.h, self.J, self.A.ctypes.data_as(_lib.c_dp),
                                      self.b.ctypes.data_as(_lib.c_dp), C.byref(md)))
        self.model = md
        self.x = DeviceVector(self.ctx, n)
        self.fcur = DeviceVector(self.ctx, m)
        self._Jd = None
        self._fg = None

    def reset(self, x0=None):
        if x0 is None:      # x <- 0 on the device (a fill kernel in stream order: no host buffer, no blocking copy)
            check(lib().lsq_fill(self.ctx.h, self.n, 0.0, self.x.ptr))
        else:
            self.x.set(x0)

    def optimize(self, optimizer_kind, solver_kind, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, iterations=1000,
                 delta=None, trace=False, allreduce=None, fetch_x=True, row_allreduce=None, row_allreduce_user=None,
                 global_rows=0):
        # (the handle wrapper and the two callback pointers are made once per problem: a solve of the bench schedule takes 2 ms,
        #  and tens of microseconds of interpreter work per call showed in its timeline)
        if self._Jd is None:
            L = lib()
            self._Jd = _Handle(self.J)
            self._fg = (L.lsq_model_f(), L.lsq_model_g())
        Jd = self._Jd
        st, res, tr = _run_native(self.ctx, optimizer_kind, solver_kind, Jd, self.x, self.fcur,
                                  self._fg[0], self._fg[1], self.model, x_tol, f_tol, g_tol,
                                  iterations, delta, None, None, trace, self.n, allreduce=allreduce,
                                  row_allreduce=row_allreduce, row_allreduce_user=row_allreduce_user, global_rows=global_rows)
        check(st)
        r = LeastSquaresResult()
        r.optimizer = "LevenbergMarquardt" if res.optimizer == _lib.LEVENBERG_MARQUARDT else "Dogleg"
        r.ssr, r.iterations = float(res.ssr), res.iterations
        r.converged = bool(res.converged)
        r.x_converged, r.f_converged, r.g_converged = bool(res.x_converged), bool(res.f_converged), bool(res.g_converged)
        r.x_tol, r.f_tol, r.g_tol = x_tol, f_tol, g_tol
        r.f_calls, r.g_calls, r.mul_calls = res.f_calls, res.g_calls, res.mul_calls
        r.seconds, r.lsmr_iterations = res.seconds, int(res.lsmr_iterations)
        r.trace = tr
        r.minimizer = self.x.get() if fetch_x else None
        return r

    def close(self):
        L = lib()
        if self.model:
            L.lsq_model_destroy(self.model)
            self.model = None
        if self.J:
            L.lsq_mat_destroy(self.J)
            self.J = None
