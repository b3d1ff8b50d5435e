"""Host-side mirror of the reference's user API for the hot path (types.jl:7-269):

    LeastSquaresProblem(x=..., f_=..., g_=..., J=..., y=..., output_length=...)
    optimize_(nls, Dogleg(QR()) | LevenbergMarquardt(LSMR()) | ..., x_tol=..., lower=..., ...)
    optimize(f, x, optimizer, ...)

(`f_`/`g_`/`optimize_` stand for Julia's `f!`/`g!`/`optimize!`).  The trust-region control runs
on the host, every m-/n-/nnz-length array lives on the MI355X and every arithmetic step is a
hand-written HIP kernel reached through the C ABI of include/lsqhip.h.  The Julia side of the same
boundary is shown in INTEGRATION.md.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (ArgumentError, DimensionMismatch, check, lib)

try:  # scipy is only needed to accept csc_matrix Jacobians
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None


# ------------------------------------------------------------------------------------------------
# solver / optimizer selectors (types.jl:76-98)
# ------------------------------------------------------------------------------------------------
class AbstractSolver:
    pass


class QR(AbstractSolver):
    kind = _lib.QR


class Cholesky(AbstractSolver):
    kind = _lib.CHOLESKY


class LSMR(AbstractSolver):
    """LSMR(preconditioner!, P) (types.jl:82-86).

    * `LSMR(preconditioner=fn)` -- DIAGONAL preconditioners, fused into the device-resident recurrence: `fn(P, J, damp)` is
      the reference's preconditioner!(P, x, J, damp); it must fill the DeviceVector P with the factors the preconditioner
      solve multiplies by (an InverseDiagonal stores the inverse, iterative_lsmr.jl:117-122) for J'J + diag(damp); damp is a
      DeviceVector with the un-rooted damping or None (Dogleg).  The solver owns the storage P.
    * `LSMR(preconditioner=fn, P=obj)` -- ANY preconditioner that supports ldiv! (README.md:47 of the reference): `obj` is the
      caller's own object with a method `obj.ldiv(out, x)` (= ldiv!(out, P, x) on DeviceVectors); `fn(obj, J, damp)` refreshes
      it before every solve (None: P never changes).  Runs the operator-level recurrence (lsq_lsmr_general.hip): the slow
      path, as a user-supplied P is in the reference.
    Either callback runs on the host before / inside every solve.  None = the built-in Jacobi preconditioner
    (iterative_lsmr.jl:129-141)."""
    kind = _lib.LSMR

    def __init__(self, preconditioner=None, P=None):
        if P is not None and not hasattr(P, "ldiv"):
            raise TypeError("LSMR(preconditioner!, P): P must provide ldiv(out, x)  (ldiv!(out, P, x))")
        self.preconditioner = preconditioner
        self.P = P


def _general_precond_trampolines(solver, ctx, J):
    """(update_cb, ldiv_cb) for LSMR(preconditioner!, P) with a general P."""
    P, fn = solver.P, solver.preconditioner

    def _update(_jh, d_damp, _user):
        try:
            if fn is not None:
                fn(P, J, DeviceVector.borrow(ctx, J.n, d_damp) if d_damp else None)
            return 0
        except Exception as e:   # pragma: no cover
            import sys
            print("preconditioner! callback failed:", e, file=sys.stderr)
            return 1

    def _ldiv(d_out, d_in, _user):
        try:
            P.ldiv(DeviceVector.borrow(ctx, J.n, d_out), DeviceVector.borrow(ctx, J.n, d_in))
            return 0
        except Exception as e:   # pragma: no cover
            import sys
            print("preconditioner ldiv callback failed:", e, file=sys.stderr)
            return 1

    return _lib.PRECOND_UPDATE_CALLBACK(_update), _lib.PRECOND_LDIV_CALLBACK(_ldiv)


def _precond_trampoline(fn, ctx, J):
    """ctypes callback (d_P, J_handle, d_damp, user) -> fn(P, J, damp) on borrowed DeviceVector views."""
    def _cb(d_p, _jh, d_damp, _user):
        try:
            P = DeviceVector.borrow(ctx, J.n, d_p)
            damp = DeviceVector.borrow(ctx, J.n, d_damp) if d_damp else None
            fn(P, J, damp)
            return 0
        except Exception as e:   # pragma: no cover
            import sys
            print("preconditioner callback failed:", e, file=sys.stderr)
            return 1
    return _lib.PRECOND_CALLBACK(_cb)


class AbstractOptimizer:
    def __init__(self, solver=None):
        if isinstance(solver, type):
            solver = solver()
        self.solver = solver


class Dogleg(AbstractOptimizer):
    kind = _lib.DOGLEG
    name = "Dogleg"


class LevenbergMarquardt(AbstractOptimizer):
    kind = _lib.LEVENBERG_MARQUARDT
    name = "LevenbergMarquardt"


def _is_sparse(J):
    return _sp is not None and _sp.issparse(J)


def default_solver(solver, J):
    """types.jl:114-121"""
    matrix_free = type(J).__name__ == "DeviceOperator"
    if solver is None:
        return LSMR() if (_is_sparse(J) or matrix_free) else QR()
    if matrix_free and not isinstance(solver, LSMR):
        raise ArgumentError(_lib.EARG, "a matrix-free Jacobian works with LSMR() only (README.md:37-47)")
    if isinstance(solver, QR) and _is_sparse(J):
        raise ArgumentError(_lib.EARG, "solver QR() is not available for sparse Jacobians. "
                                       "Choose between Cholesky() and LSMR()")
    return solver


def default_optimizer(optimizer, solver):
    """types.jl:123-127"""
    if isinstance(optimizer, Dogleg):
        return Dogleg(solver)
    if isinstance(optimizer, LevenbergMarquardt):
        return LevenbergMarquardt(solver)
    if isinstance(solver, LSMR):
        return LevenbergMarquardt(solver)
    return Dogleg(solver)


# ------------------------------------------------------------------------------------------------
# device context and buffers
# ------------------------------------------------------------------------------------------------
class Context:
    """One per device/stream (SURVEY 8b 'Threading'); fails loudly without a HIP device."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib().lsq_ctx_create(int(device), stream, C.byref(h)))
        self.h = h
        self.device = device

    def sync(self):
        check(lib().lsq_ctx_sync(self.h))

    def device_info(self):
        """lsq_ctx_device_info: {'num_cus', 'num_xcds', 'arch'} -- what the launch heuristics see (256 CUs unpartitioned)."""
        a, b = C.c_int(0), C.c_int(0)
        name = C.create_string_buffer(128)
        check(lib().lsq_ctx_device_info(self.h, C.byref(a), C.byref(b), name, 128))
        return {"num_cus": a.value, "num_xcds": b.value, "arch": name.value.decode()}

    @property
    def num_cus(self):
        return self.device_info()["num_cus"]

    def fallback_stats(self):
        """lsq_ctx_fallback_stats: how often the co-residency fast paths of this context's solvers gave up on a bounded wait."""
        g = (C.c_int * 4)()
        check(lib().lsq_ctx_fallback_stats(self.h, g))
        return dict(zip(("chol_one_launch", "tri_pipeline", "qr_exchange", "cholqr_panel"), (int(v) for v in g)))

    def tail_stats(self):
        """lsq_ctx_tail_stats: (LSMR solves whose follow-up kernels were queued behind a guessed last iteration, wrong guesses)."""
        v = (C.c_longlong * 2)()
        check(lib().lsq_ctx_tail_stats(self.h, v))
        return int(v[0]), int(v[1])

    def occupy(self, workgroups, lds_bytes=65536, milliseconds=10.0):
        """A neighbour on the device (lsq_bench_occupy): workgroups that hold LDS and spin, on a stream of their own."""
        check(lib().lsq_bench_occupy(self.h, int(workgroups), int(lds_bytes), float(milliseconds)))

    def occupy_wait(self):
        check(lib().lsq_bench_occupy_wait(self.h))

    def close(self):
        if self.h:
            lib().lsq_ctx_destroy(self.h)
            self.h = None


_DEFAULT_CTX = {}


def default_context(device=0):
    if device not in _DEFAULT_CTX:
        _DEFAULT_CTX[device] = Context(device)
    return _DEFAULT_CTX[device]


class DeviceVector:
    """fp64 vector in HBM (the `HipVector` of SURVEY 8b)."""

    def __init__(self, ctx, n, data=None):
        self.ctx, self.n = ctx, int(n)
        p = C.c_void_p()
        check(lib().lsq_malloc(ctx.h, max(self.n, 1) * 8, C.byref(p)))
        self.ptr = p
        if data is not None:
            self.set(data)
        else:
            check(lib().lsq_fill(ctx.h, self.n, 0.0, self.ptr))

    @classmethod
    def borrow(cls, ctx, n, ptr):
        """A view of device memory owned by someone else (never freed here)."""
        v = cls.__new__(cls)
        v.ctx, v.n, v.ptr, v._borrowed = ctx, int(n), ptr, True
        return v

    def set(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        if a.size != self.n:
            raise DimensionMismatch(_lib.EDIM, "vector has length %d, expected %d" % (a.size, self.n))
        check(lib().lsq_h2d(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.n * 8))

    def get(self):
        out = np.empty(self.n)
        check(lib().lsq_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 8))
        return out

    def free(self):
        if self.ptr and not getattr(self, "_borrowed", False):
            lib().lsq_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(v):
    return None if v is None else v.ptr


class PinnedBuffer:
    """Page-locked host memory (lsq_host_alloc) as a float64 numpy array: what a host-side g! should write the Jacobian
    values into, so that the upload after every g!(J, x) runs asynchronously at the PCIe rate
    (DeviceMatrix.set_values_async).  `array` is only valid until free()."""

    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        p = C.c_void_p()
        check(lib().lsq_host_alloc(ctx.h, max(self.n, 1) * 8, C.byref(p)))
        self.ptr = p
        self.array = np.ctypeslib.as_array(C.cast(p, _lib.c_dp), shape=(max(self.n, 1),))[:self.n]

    def free(self):
        if self.ptr:
            self.array = None
            lib().lsq_host_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceMatrix:
    """Jacobian handle: dense column-major or CSC (+ CSR mirror) -- `HipDense` / `HipCSC`."""

    def __init__(self, ctx, J):
        self.ctx = ctx
        h = C.c_void_p()
        L = lib()
        if _is_sparse(J):
            S = J.tocsc()
            S.sort_indices()
            self.sparse = True
            self.m, self.n = S.shape
            self.colptr = np.ascontiguousarray(S.indptr, dtype=np.int32)
            self.rowval = np.ascontiguousarray(S.indices, dtype=np.int32)
            check(L.lsq_csc_create(ctx.h, self.m, self.n, self.colptr.ctypes.data_as(_lib.c_ip),
                                   self.rowval.ctypes.data_as(_lib.c_ip), C.byref(h)))
            self.h = h
            self.nnz = int(S.nnz)
            self.set_values(S.data)
        else:
            A = np.asarray(J, dtype=np.float64)
            if A.ndim != 2:
                raise DimensionMismatch(_lib.EDIM, "J must be a matrix")
            self.sparse = False
            self.m, self.n = A.shape
            check(L.lsq_dense_create(ctx.h, self.m, self.n, C.byref(h)))
            self.h = h
            self.nnz = self.m * self.n
            self.set_values(np.asfortranarray(A).reshape(-1, order="F"))

    def set_values(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        if v.size != self.nnz:
            raise DimensionMismatch(_lib.EDIM, "expected %d values, got %d" % (self.nnz, v.size))
        check(lib().lsq_mat_set_values(self.h, v.ctypes.data_as(_lib.c_dp)))

    def set_values_async(self, pinned):
        """Upload from a PinnedBuffer without blocking the host (lsq_mat_set_values_async): later uses of J wait for the
        copy on the device; the buffer must not be rewritten before upload_wait() (or a call that reads results back)."""
        if pinned.n != self.nnz:
            raise DimensionMismatch(_lib.EDIM, "expected %d values, got %d" % (self.nnz, pinned.n))
        check(lib().lsq_mat_set_values_async(self.h, C.cast(pinned.ptr, _lib.c_dp)))

    def upload_wait(self):
        check(lib().lsq_mat_upload_wait(self.h))

    def values(self):
        out = np.empty(self.nnz)
        check(lib().lsq_mat_get_values(self.h, out.ctypes.data_as(_lib.c_dp)))
        return out

    def set_colscale(self, s):
        """J = V diag(s) (lsq_mat_set_colscale): the values held now are V, `s` a DeviceVector of n factors that stays alive
        (and may be rewritten, followed by colscale_changed()) as long as the scale is set; None removes it."""
        if s is not None and s.n != self.n:
            raise DimensionMismatch(_lib.EDIM, "column scale has length %d, expected %d" % (s.n, self.n))
        check(lib().lsq_mat_set_colscale(self.h, s.ptr if s is not None else None))
        self._colscale = s

    def colscale_changed(self):
        check(lib().lsq_mat_colscale_changed(self.h))

    def free(self):
        if self.h:
            lib().lsq_mat_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# --- operator interface (README.md:37-47) on device objects -------------------------------------
def mul_(y, J, x, alpha=1.0, beta=0.0, trans=False):
    """mul!(y, J, x, alpha, beta) / mul!(y, J', x, alpha, beta)"""
    check(lib().lsq_mul(J.h, 1 if trans else 0, float(alpha), x.ptr, float(beta), y.ptr))
    return y


def colsumabs2_(out, J):
    check(lib().lsq_colsumabs2(J.h, out.ptr))
    return out


def rowsumabs2_(out, J):
    """rowsumabs2!(out, J) (utils.jl:153-161): what colsumabs2! of an adjoint Jacobian computes."""
    check(lib().lsq_rowsumabs2(J.h, out.ptr))
    return out


# BLAS-1 on device vectors -- exactly what lsmr.jl:30-44 and the optimizer loops ask of a vector type
def axpy_(a, x, y):
    """axpy!(a, x, y): y += a*x"""
    check(lib().lsq_axpy(x.ctx.h, x.n, float(a), x.ptr, y.ptr))
    return y


def rmul_(x, a):
    """rmul!(x, a)"""
    check(lib().lsq_scal(x.ctx.h, x.n, float(a), x.ptr))
    return x


def copyto_(dst, src):
    """copyto!(dst, src)"""
    check(lib().lsq_copy(src.ctx.h, src.n, src.ptr, dst.ptr))
    return dst


def fill_(x, a):
    """fill!(x, a)"""
    check(lib().lsq_fill(x.ctx.h, x.n, float(a), x.ptr))
    return x


def clamp_(x, lo, hi):
    """clamp!(x, lo, hi)"""
    check(lib().lsq_clamp(x.ctx.h, x.n, float(lo), float(hi), x.ptr))
    return x


def ediv_(out, x, y):
    """map!(/, out, x, y)"""
    check(lib().lsq_ediv(x.ctx.h, x.n, x.ptr, y.ptr, out.ptr))
    return out


def box_clip_(dx, x, lower=None, upper=None):
    """the step clipping of levenberg_marquardt.jl:89-98 / dogleg.jl:148-160"""
    check(lib().lsq_box_clip(x.ctx.h, x.n, dx.ptr, x.ptr, _ptr(lower), _ptr(upper)))
    return dx


def vsum(x):
    return _scalar(lib().lsq_sum, x.ctx, x.n, x.ptr)


def first_nonfinite(x):
    """check_isfinite (utils.jl:70-75): first non-finite index or -1"""
    r = C.c_int(0)
    check(lib().lsq_first_nonfinite(x.ctx.h, x.n, x.ptr, C.byref(r)))
    return r.value


def _scalar(fn, ctx, n, *ptrs):
    r = C.c_double(0.0)
    check(fn(ctx.h, n, *ptrs, C.byref(r)))
    return r.value


def sumsq(x):
    return _scalar(lib().lsq_sumsq, x.ctx, x.n, x.ptr)


def norm(x):
    return _scalar(lib().lsq_nrm2, x.ctx, x.n, x.ptr)


def wdot(x, y, w):
    return _scalar(lib().lsq_wdot, x.ctx, x.n, x.ptr, y.ptr, w.ptr)


def wnorm(x, w):
    return float(np.sqrt(wdot(x, x, w)))


def maxabs(x):
    return _scalar(lib().lsq_amax, x.ctx, x.n, x.ptr)


def maxabs_projected_gradient(g, x, lower=None, upper=None):
    return _scalar(lib().lsq_amax_projected, g.ctx, g.n, g.ptr, x.ptr, _ptr(lower), _ptr(upper))


def set_exact(on=None):
    """Reference-order arithmetic for small problems (include/lsqhip.h: lsq_set_exact).
    True / False force it on / off; None restores the default (on unless LSQ_EXACT=0)."""
    check(lib().lsq_set_exact(-1 if on is None else (1 if on else 0)))


def debug_set(launch_jitter_us=None, serial=None):
    """Process-wide debug modes of the library (include/lsqhip.h: lsq_debug_set): random host stalls in front of the
    kernel launches / serialised launches (1: same kernels, bit-identical results; 2: also without the in-kernel
    workgroup exchanges).  None leaves a setting alone."""
    check(lib().lsq_debug_set(-1 if launch_jitter_us is None else int(launch_jitter_us), -1 if serial is None else int(serial)))


def debug_get():
    """(launch_jitter_us, serial, stalls injected so far)"""
    import ctypes as _C
    a, b, c = _C.c_int(0), _C.c_int(0), _C.c_longlong(0)
    check(lib().lsq_debug_get(_C.byref(a), _C.byref(b), _C.byref(c)))
    return a.value, b.value, c.value


class DeviceOperator:
    """A matrix-free Jacobian (README.md:37-47 of the reference: any type with mul!, the adjoint's mul!,
    colsumabs2!, size and eltype works with LSMR).  `mul(trans, x, out)` writes J*x (trans = False, m
    entries) or J'*x (trans = True, n entries) into the DeviceVector `out`; `colsumabs2(out)` the n column
    sums of squares.  Both are called on the host with device vectors; use `refresh()` after the operator
    changed (what g! does for a stored Jacobian).  LSMR only."""
    sparse = True   # (types.jl:114-121: a non-dense Jacobian defaults to LSMR)

    def __init__(self, ctx, m, n, mul, colsumabs2):
        self.ctx, self.m, self.n = ctx, int(m), int(n)
        self.shape = (self.m, self.n)
        self._errors = []

        def _mul(trans, d_x, d_out, _user):
            try:
                t = bool(trans)
                mul(t, DeviceVector.borrow(ctx, self.m if t else self.n, d_x),
                    DeviceVector.borrow(ctx, self.n if t else self.m, d_out))
                ctx.sync()
                return 0
            except Exception as e:
                self._errors.append(e)
                return 1

        def _cs(d_out, _user):
            try:
                colsumabs2(DeviceVector.borrow(ctx, self.n, d_out))
                ctx.sync()
                return 0
            except Exception as e:
                self._errors.append(e)
                return 1

        self._cbs = (_lib.OP_MUL_CALLBACK(_mul), _lib.OP_COLSUM_CALLBACK(_cs))
        h = C.c_void_p()
        check(lib().lsq_op_create(ctx.h, self.m, self.n, self._cbs[0], self._cbs[1], None, C.byref(h)))
        self.h = h

    def refresh(self):
        check(lib().lsq_mat_refresh(self.h))

    def free(self):
        if self.h:
            lib().lsq_mat_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class AllocatedSolver:
    """AbstractAllocatedSolver(nls, optimizer) + ldiv! (the L2 plug point)."""

    def __init__(self, J, solver, for_lm):
        h = C.c_void_p()
        check(lib().lsq_solver_create(J.ctx.h, J.h, solver.kind, 1 if for_lm else 0, C.byref(h)))
        self.h, self.J = h, J
        self._pc = None
        if getattr(solver, "P", None) is not None:
            self._pc = _general_precond_trampolines(solver, J.ctx, J)
            check(lib().lsq_solver_set_general_preconditioner(self.h, self._pc[0], self._pc[1], None))
        elif getattr(solver, "preconditioner", None) is not None:
            self._pc = _precond_trampoline(solver.preconditioner, J.ctx, J)
            check(lib().lsq_solver_set_preconditioner(self.h, self._pc, None))

    def set_row_allreduce(self, hook, global_rows):
        """lsq_solver_set_row_allreduce: J (and y) of the coming ldiv! calls are this rank's ROW BLOCK of one problem with
        `global_rows` residuals; `hook` (rowshard.RcclRowAllreduce / HostStagedRowAllreduce, or None to switch it off) sums
        the replicated n-vectors over the ranks.  LSMR() only."""
        self._row_hook = hook       # (keeps the ctypes callback alive)
        if hook is None:
            check(lib().lsq_solver_set_row_allreduce(self.h, _lib.ROW_ALLREDUCE_CALLBACK(), None, 0))
        else:
            check(lib().lsq_solver_set_row_allreduce(self.h, hook.callback, hook.user, int(global_rows)))

    def ldiv_(self, x, y, damp=None):
        """ldiv!(x, J, y[, damp], A) -> (x, nmul)"""
        n = C.c_int(0)
        if damp is None:
            check(lib().lsq_ldiv(self.h, self.J.h, y.ptr, x.ptr, C.byref(n)))
        else:
            check(lib().lsq_ldiv_damped(self.h, self.J.h, y.ptr, damp.ptr, x.ptr, C.byref(n)))
        return x, n.value

    def info(self):
        it, st, rk = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().lsq_solver_info(self.h, C.byref(it), C.byref(st), C.byref(rk)))
        path = C.c_int(0)
        check(lib().lsq_solver_qr_path(self.h, C.byref(path)))
        cpath = C.c_int(0)
        check(lib().lsq_solver_chol_path(self.h, C.byref(cpath)))
        panel = C.c_int(0)
        check(lib().lsq_solver_qr_panel(self.h, C.byref(panel)))
        return dict(lsmr_iter=it.value, lsmr_istop=st.value, qr_rank=rk.value,
                    qr_panel={0: None, 1: "householder-steps", 2: "cholqr2"}[panel.value],
                    qr_path={0: None, 1: "one-stage", 2: "two-stage-pivoted", 3: "two-stage-certified"}[path.value],
                    chol_path={0: None, 1: "one-workgroup", 2: "blocked", 3: "blocked-certified", 4: "blocked-one-launch"}[cpath.value])

    def stats(self):
        """lsq_solver_stats: give-ups of the co-residency fast paths and how many solves each stays paused."""
        g, p = (C.c_int * 4)(), (C.c_int * 4)()
        check(lib().lsq_solver_stats(self.h, g, p))
        names = ("chol_one_launch", "tri_pipeline", "qr_exchange", "cholqr_panel")
        return {k: {"giveups": g[i], "paused": p[i]} for i, k in enumerate(names)}

    def free(self):
        if self.h:
            lib().lsq_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# problem / result types
# ------------------------------------------------------------------------------------------------
class OptimizationState:
    def __init__(self, iteration, value, g_norm):
        self.iteration, self.value, self.g_norm = iteration, value, g_norm

    def __repr__(self):
        return "%6d   %14e   %14e" % (self.iteration, self.value, self.g_norm)


class LeastSquaresProblem:
    """types.jl:7-68.  J may be a numpy matrix (dense) or a scipy.sparse CSC matrix with a FIXED
    pattern whose `.data` g_ overwrites (test/nonlinearleastsquares.jl:47-86)."""

    def __init__(self, x=None, y=None, f_=None, g_=None, J=None, output_length=0, autodiff="central"):
        if x is None:
            raise ValueError("initial x required")
        if f_ is None:
            raise ValueError("initial f! required")
        self.x = np.array(x, dtype=np.float64)
        if y is None:
            if output_length == 0:
                if J is None:
                    raise ValueError("specify J or output_length")
                output_length = J.shape[0]           # types.jl:46 (size(J, 1))
            y = np.zeros(output_length)
        self.y = np.asarray(y, dtype=np.float64)
        if J is None:
            J = np.zeros((len(self.y), len(self.x)), order="F")
        if type(J).__name__ == "DeviceOperator":
            if g_ is None:
                raise ValueError("a matrix-free Jacobian needs g_ (it updates the operator's own state)")
        elif _is_sparse(J):
            J = J.tocsc()
            J.sort_indices()  # g_ writes J.data in this (canonical CSC) order
        else:
            J = np.asfortranarray(J, dtype=np.float64)
        if len(self.x) != J.shape[1]:
            raise DimensionMismatch(_lib.EDIM, "x must have length size(J, 2)")
        if len(self.y) != J.shape[0]:
            raise DimensionMismatch(_lib.EDIM, "y must have length size(J, 1)")
        self.J = J
        self.f_ = f_
        if g_ is None:
            if autodiff == "central":
                g_ = _central_difference_jacobian(f_, len(self.y))
            elif autodiff == "forward":
                raise NotImplementedError("autodiff=:forward (ForwardDiff) is off the hot path and "
                                          "never exercised by the reference's tests")
            else:
                raise ValueError("Invalid automatic differentiation method.")  # DomainError
        self.g_ = g_


def _central_difference_jacobian(f_, m):
    """FiniteDiff-style central differences (types.jl:55-58) -- host side, off the hot path."""
    eps3 = np.finfo(float).eps ** (1.0 / 3.0)

    def g_(J, x):
        if _is_sparse(J):
            raise ArgumentError(_lib.EARG, "autodiff Jacobians are dense only (types.jl:57)")
        fp, fm = np.zeros(m), np.zeros(m)
        xp = np.array(x, dtype=np.float64)
        for j in range(len(x)):
            h = max(eps3 * abs(x[j]), eps3)
            xj = xp[j]
            xp[j] = xj + h
            f_(fp, xp)
            xp[j] = xj - h
            f_(fm, xp)
            xp[j] = xj
            J[:, j] = (fp - fm) / (2 * h)

    return g_


class LeastSquaresResult:
    """types.jl:220-269"""

    def __repr__(self):
        ok = self.x_converged or self.f_converged or self.g_converged
        return ("Results of Optimization Algorithm\n * Status: %s\n\n * Candidate solution\n"
                "    Final objective value:     %.6e\n\n * Found with\n    Algorithm:     %s\n\n"
                " * Convergence measures\n    |x - x'|               %s %.1e\n"
                "    |f(x) - f(x')| / |f(x)| %s %.1e\n    |g(x)|                 %s %.1e\n\n"
                " * Work counters\n    Iterations:    %d\n    f(x) calls:    %d\n"
                "    J(x) calls:    %d\n    mul! calls:    %d\n" % (
                    "success" if ok else "failure (reached maximum number of iterations)",
                    self.ssr, self.optimizer, "<=" if self.x_converged else "!<=", self.x_tol,
                    "<=" if self.f_converged else "!<=", self.f_tol,
                    "<=" if self.g_converged else "!<=", self.g_tol,
                    self.iterations, self.f_calls, self.g_calls, self.mul_calls))


def converged(r):
    return r.x_converged or r.f_converged or r.g_converged


def _run_native(ctx, optimizer_kind, solver_kind, Jd, dx, dy, fcb, gcb, user, x_tol, f_tol, g_tol,
                iterations, delta, lower, upper, trace, n, allreduce=None, preconditioner=None, row_allreduce=None,
                row_allreduce_user=None, global_rows=0, general_preconditioner=None):
    L = lib()
    opt = _lib.Options()
    L.lsq_options_default(C.byref(opt))
    opt.x_tol, opt.f_tol, opt.g_tol = float(x_tol), float(f_tol), float(g_tol)
    opt.iterations = int(iterations)
    opt.delta = float(delta) if delta is not None else -1.0
    keep = []
    if lower is not None and len(lower):
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        if len(lo) != n:
            raise ArgumentError(_lib.EARG, "Bounds must either be empty or of the same length as "
                                           "the number of parameters.")
        opt.h_lower = lo.ctypes.data_as(_lib.c_dp)
        keep.append(lo)
    if upper is not None and len(upper):
        hi = np.ascontiguousarray(upper, dtype=np.float64)
        if len(hi) != n:
            raise ArgumentError(_lib.EARG, "Bounds must either be empty or of the same length as "
                                           "the number of parameters.")
        opt.h_upper = hi.ctypes.data_as(_lib.c_dp)
        keep.append(hi)
    if allreduce is not None:
        if hasattr(allreduce, "callback"):      # an exchange served in C (sharding.RcclScalarExchange): function + handle
            if hasattr(allreduce, "reset"):
                allreduce.reset()               # (the protocol state is per run: include/lsqrccl.h)
            opt.allreduce = allreduce.callback
            opt.allreduce_user = allreduce.user
        else:                                   # a ctypes callback (sharding.make_allreduce_callback)
            opt.allreduce = allreduce
        keep.append(allreduce)
    if preconditioner is not None:
        opt.preconditioner = preconditioner
        keep.append(preconditioner)
    if general_preconditioner is not None:     # LSMR(preconditioner!, P) with a general P: (update_cb, ldiv_cb)
        opt.precond_update, opt.precond_ldiv = general_preconditioner
        keep.append(general_preconditioner)
    if row_allreduce is not None:       # row-sharded single problem (lsq_options.row_allreduce; rowshard.py)
        opt.row_allreduce = row_allreduce
        opt.row_allreduce_user = row_allreduce_user
        opt.global_rows = int(global_rows)
        keep.append(row_allreduce)
    tr = None
    if trace:
        cap = int(iterations)
        tr = dict(ssr=np.zeros(cap), gnorm=np.zeros(cap), delta=np.zeros(cap), rho=np.zeros(cap),
                  inner=np.zeros(cap, dtype=np.int32), accept=np.zeros(cap, dtype=np.int32),
                  x=np.zeros((cap, n)))
        opt.trace_cap = cap
        opt.trace_ssr = tr["ssr"].ctypes.data_as(_lib.c_dp)
        opt.trace_gnorm = tr["gnorm"].ctypes.data_as(_lib.c_dp)
        opt.trace_delta = tr["delta"].ctypes.data_as(_lib.c_dp)
        opt.trace_rho = tr["rho"].ctypes.data_as(_lib.c_dp)
        opt.trace_inner = tr["inner"].ctypes.data_as(_lib.c_ip)
        opt.trace_accept = tr["accept"].ctypes.data_as(_lib.c_ip)
        opt.trace_x = tr["x"].ctypes.data_as(_lib.c_dp)
    res = _lib.Result()
    st = L.lsq_optimize(ctx.h, optimizer_kind, solver_kind, Jd.h, dx.ptr, dy.ptr, fcb, gcb, user,
                        C.byref(opt), C.byref(res))
    if tr is not None:
        k = res.iterations
        tr = {key: v[:k].copy() for key, v in tr.items()}
    return st, res, tr


def optimize_(nls, optimizer=None, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, iterations=1000, delta=None,
              store_trace=False, show_trace=False, show_every=1, lower=(), upper=(), ctx=None,
              full_trace=False, row_allreduce=None, global_rows=0):
    """optimize!(nls, optimizer; kwargs...)  -- types.jl:207-209 then
    levenberg_marquardt.jl:39-144 / dogleg.jl:41-203.  Mutates nls.x, nls.y, nls.J in place."""
    allocated = nls if isinstance(nls, LeastSquaresProblemAllocated) else None
    if allocated is not None:
        # optimize!(nls::LeastSquaresProblemAllocated; kwargs...): buffers, solver and optimizer were chosen at
        # allocation (types.jl:141-160); nothing is allocated here, on the host or on the device
        if optimizer is not None:
            raise TypeError("an allocated problem carries its optimizer (types.jl:152-157)")
        ctx, optimizer, solver = allocated.ctx, allocated.optimizer, allocated.solver
    else:
        ctx = ctx or default_context()
        solver = default_solver(optimizer.solver if optimizer is not None else None, nls.J)
        optimizer = default_optimizer(optimizer, solver)
    n, m = len(nls.x), len(nls.y)
    is_op = isinstance(nls.J, DeviceOperator)
    if allocated is not None:
        Jd, dx, dy = allocated._Jd, allocated._dx, allocated._dy
        dx.set(nls.x)
        dy.set(nls.y)
    else:
        Jd = nls.J if is_op else DeviceMatrix(ctx, nls.J)
        dx, dy = DeviceVector(ctx, n, nls.x), DeviceVector(ctx, m, nls.y)
    L = lib()
    xh, yh = np.zeros(n), np.zeros(m)
    err = []
    # Host-side g!: the Jacobian values live in PAGE-LOCKED memory for the duration of the solve -- g! writes them there
    # directly (a sparse J gets its .data rebound to the pinned array, a dense J is handed over as a pinned column-major
    # view) and the upload after every g!(J, x) is asynchronous (lsq_mat_set_values_async): no staging copy, no blocked
    # host, PCIe rate instead of the pageable-copy rate.  SURVEY 8f-1; levenberg_marquardt.jl:77-81 is where the upload sits.
    stage = None
    J_user = nls.J
    data_user = None
    J_for_g = None

    def bind_stage():
        # (called inside the try below: whatever happens after the rebinding, the finally hands J.data back)
        nonlocal stage, data_user, J_for_g
        stage = allocated._stage if allocated is not None and allocated._stage is not None else PinnedBuffer(ctx, Jd.nnz)
        if Jd.sparse:
            data_user = nls.J.data
            np.copyto(stage.array, data_user)
            nls.J.data = stage.array
            J_for_g = nls.J
        else:
            J_for_g = stage.array.reshape((m, n), order="F")
            np.copyto(J_for_g, nls.J)

    def fcb(d_out, d_x, _):
        try:
            check(L.lsq_d2h(ctx.h, xh.ctypes.data_as(C.c_void_p), d_x, n * 8))
            nls.f_(yh, xh)
            check(L.lsq_h2d(ctx.h, d_out, yh.ctypes.data_as(C.c_void_p), m * 8))
            return 0
        except Exception as e:  # surfaced after the C call returns
            err.append(e)
            return 1

    def gcb(Jh, d_x, _):
        try:
            check(L.lsq_d2h(ctx.h, xh.ctypes.data_as(C.c_void_p), d_x, n * 8))
            if is_op:   # g! updates the operator's own state; nothing to upload
                nls.g_(nls.J, xh)
                return 0
            Jd.upload_wait()            # (the previous upload has long finished; g! is about to overwrite its source)
            nls.g_(J_for_g, xh)
            if Jd.sparse and nls.J.data is not stage.array:     # g! replaced J.data instead of writing into it
                np.copyto(stage.array, nls.J.data)
                nls.J.data = stage.array
            Jd.set_values_async(stage)
            return 0
        except Exception as e:
            err.append(e)
            return 1

    F, G = _lib.F_CALLBACK(fcb), _lib.G_CALLBACK(gcb)
    tracing = store_trace or show_trace or full_trace
    st = res = tr = None
    try:
        if not is_op:
            bind_stage()
        pc = gpc = None
        if getattr(solver, "P", None) is not None:
            gpc = _general_precond_trampolines(solver, ctx, Jd)
        elif getattr(solver, "preconditioner", None) is not None:
            pc = _precond_trampoline(solver.preconditioner, ctx, Jd)
        # row_allreduce: a rowshard.RcclRowAllreduce / HostStagedRowAllreduce -- nls then holds this rank's ROWS of one
        # larger problem (J, y local; x replicated), SURVEY 8f-4
        st, res, tr = _run_native(ctx, optimizer.kind, solver.kind, Jd, dx, dy, F, G, None, x_tol, f_tol,
                                  g_tol, iterations, delta, lower, upper, tracing, n, preconditioner=pc,
                                  general_preconditioner=gpc,
                                  row_allreduce=row_allreduce.callback if row_allreduce is not None else None,
                                  row_allreduce_user=row_allreduce.user if row_allreduce is not None else None,
                                  global_rows=global_rows)
    finally:
        if stage is not None:       # hand the values back in ordinary memory before the pinned buffer can go away
            Jd.upload_wait()
            if Jd.sparse:
                if data_user is not None:
                    data_user[:] = stage.array
                    nls.J.data = data_user
            elif J_for_g is not None:
                np.copyto(J_user, J_for_g)
            if allocated is None or allocated._stage is None:
                stage.free()
        if st is not None:
            # optimize! mutates nls.x / nls.y in place (levenberg_marquardt.jl:46): when an iteration throws
            # (RankDeficientException, IsFiniteException, ...) they hold that iteration's iterate, as in the reference
            nls.x[:] = dx.get()
            nls.y[:] = dy.get()
    if err:
        raise err[0]
    if st == _lib.ENONFINITE:
        e = _lib.IsFiniteException(st, lib().lsq_last_error().decode())
        e.indices = [res.bad_index]
        raise e
    check(st)
    r = LeastSquaresResult()
    r.optimizer = optimizer.name
    r.minimizer = nls.x
    r.ssr = float(res.ssr)
    r.iterations = res.iterations
    r.converged = bool(res.converged)
    r.x_converged, r.f_converged, r.g_converged = bool(res.x_converged), bool(res.f_converged), bool(res.g_converged)
    r.x_tol, r.f_tol, r.g_tol = float(x_tol), float(f_tol), float(g_tol)
    r.f_calls, r.g_calls, r.mul_calls = res.f_calls, res.g_calls, res.mul_calls
    r.jacobian = nls.J
    r.seconds = res.seconds
    r.trace = tr
    states = []
    if tracing and tr is not None:
        # levenberg_marquardt.jl:70 / dogleg.jl:74: state 0 = (0, ssr(x0), Inf), then one state per iteration
        states.append(OptimizationState(0, float(res.ssr0), float("inf")))
        for k in range(res.iterations):
            states.append(OptimizationState(k + 1, tr["ssr"][k], tr["gnorm"][k]))
        if show_trace:
            print("Iter     Function value   Gradient norm ")
            print("------   --------------   --------------")
            for s_ in states:
                if s_.iteration % show_every == 0:
                    print(s_)
    r.tr = states if store_trace else []
    if not is_op and allocated is None:
        Jd.free()
    return r


class LeastSquaresProblemAllocated:
    """LeastSquaresProblemAllocated(nls, optimizer) (types.jl:141-160, exported by the reference): the problem
    together with its optimizer, its solver and every buffer they need, allocated once; `optimize_(nlsa)` can then
    be called repeatedly (e.g. from different starting points written into `nlsa.x`) without allocating -- the
    device Jacobian handle, the vectors and, inside the library, the loop/solver workspace are all reused."""

    def __init__(self, nls, optimizer=None, ctx=None):
        if not isinstance(nls, LeastSquaresProblem):
            raise TypeError("LeastSquaresProblemAllocated(nls::LeastSquaresProblem, optimizer)")
        self.ctx = ctx or default_context()
        self.solver = default_solver(optimizer.solver if optimizer is not None else None, nls.J)
        self.optimizer = default_optimizer(optimizer, self.solver)
        self.x, self.y, self.f_, self.J, self.g_ = nls.x, nls.y, nls.f_, nls.J, nls.g_
        self._Jd = nls.J if isinstance(nls.J, DeviceOperator) else DeviceMatrix(self.ctx, nls.J)
        self._dx = DeviceVector(self.ctx, len(nls.x), nls.x)
        self._dy = DeviceVector(self.ctx, len(nls.y), nls.y)
        # the page-locked staging buffer of the Jacobian uploads is part of the allocated problem too
        self._stage = None if isinstance(self._Jd, DeviceOperator) else PinnedBuffer(self.ctx, self._Jd.nnz)

    def free(self):
        if self._Jd is not None and not isinstance(self._Jd, DeviceOperator):
            self._Jd.free()
        self._Jd = None
        if getattr(self, "_stage", None) is not None:
            self._stage.free()
            self._stage = None


def optimize(f, x, optimizer, autodiff="central", **kwargs):
    """optimize(f, x, optimizer; kwargs...) -- types.jl:182-184 (x is copied; f returns a vector)."""
    x0 = np.array(x, dtype=np.float64)
    out0 = np.atleast_1d(np.asarray(f(x0), dtype=np.float64))

    def f_(out, xx):
        out[:] = np.atleast_1d(f(xx))

    nls = LeastSquaresProblem(x=x0.copy(), f_=f_, output_length=len(out0), autodiff=autodiff)
    return optimize_(nls, optimizer, **kwargs)
