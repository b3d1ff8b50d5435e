"""The reference's trust-region loops restated over the OPERATOR-LEVEL C ABI only
(`mul!`, `colsumabs2!`, `ldiv!`, `wdot`, `axpy!`, `rmul!`, `clamp!`, `sum(abs2, .)`, `maximum(abs, .)`).

This is what runs when the Julia shim of INTEGRATION.md plugs `HipVector` / `HipCSC` /
`HipAllocatedSolver` into the reference's OWN `optimize!` (levenberg_marquardt.jl:39-144,
dogleg.jl:41-203): every statement below is one statement of those loops, each array operation one
C-ABI call on device memory.  It exists to show that the operator-level boundary is sufficient and
is checked against the fused loop-level entry point `lsq_optimize` (tests/test_a_gpu_contract.py).
It costs one host synchronisation per returned scalar, which is why the loop-level call exists.
"""
import numpy as np

from . import _lib
from .api import (AllocatedSolver, DeviceMatrix, DeviceVector, LeastSquaresResult, axpy_, box_clip_, clamp_,
                  colsumabs2_, copyto_, default_context, default_optimizer, default_solver, ediv_, fill_,
                  first_nonfinite, maxabs, maxabs_projected_gradient, mul_, rmul_, sumsq, vsum, wdot, wnorm)

MIN_DELTA, MAX_DELTA, MIN_STEP_QUALITY = 1e-16, 1e16, 1e-3            # types.jl:107-109
MIN_DIAGONAL, MAX_DIAGONAL = 1e-6, 1e32                               # types.jl:110-111
DECREASE_THRESHOLD, INCREASE_THRESHOLD = 0.25, 0.75                   # dogleg.jl:38-39


def _div(a, b):
    """IEEE division (Julia / C semantics: x/0 is Inf or NaN, never an exception)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


def assess_convergence(dx, x, maxabs_gr, ssr, trial_ssr, xtol, ftol, grtol, step_accepted):
    """utils.jl:7-31"""
    x_c = f_c = g_c = False
    if step_accepted and abs(trial_ssr - ssr) <= ftol * (abs(ssr) + ftol):
        f_c = True
    elif maxabs(dx) <= xtol:
        x_c = True
    elif maxabs_gr <= grtol:
        g_c = True
    return x_c, f_c, g_c, (x_c or f_c or g_c)


class _Problem:
    def __init__(self, nls, ctx):
        self.nls, self.ctx = nls, ctx
        self.n, self.m = len(nls.x), len(nls.y)
        self.J = DeviceMatrix(ctx, nls.J)
        self.x, self.fcur = DeviceVector(ctx, self.n, nls.x), DeviceVector(ctx, self.m, nls.y)
        self._xh, self._yh = np.zeros(self.n), np.zeros(self.m)

    def f_(self, out, x):        # f!(out, x): host callback on device vectors
        self._xh[:] = x.get()
        self.nls.f_(self._yh, self._xh)
        out.set(self._yh)

    def g_(self, x):             # g!(J, x)
        self._xh[:] = x.get()
        self.nls.g_(self.nls.J, self._xh)
        self.J.set_values(self.nls.J.data if self.J.sparse else self.nls.J.reshape(-1, order="F"))


def _bounds(ctx, n, x, lower, upper):
    lo = DeviceVector(ctx, n, lower) if len(lower) else None
    hi = DeviceVector(ctx, n, upper) if len(upper) else None
    xh = x.get()
    if (lo is not None and not np.all(xh >= np.asarray(lower))) or (hi is not None and not np.all(xh <= np.asarray(upper))):
        raise _lib.ArgumentError(_lib.EBOUNDS, "Initial guess must be within bounds.")
    return lo, hi


def _result(name, p, ssr, it, conv, flags, tols, counts):
    r = LeastSquaresResult()
    p.nls.x[:] = p.x.get()
    p.nls.y[:] = p.fcur.get()
    r.optimizer, r.minimizer, r.ssr, r.iterations, r.converged = name, p.nls.x, float(ssr), it, conv
    r.x_converged, r.f_converged, r.g_converged = flags
    r.x_tol, r.f_tol, r.g_tol = tols
    r.f_calls, r.g_calls, r.mul_calls = counts
    r.jacobian, r.tr, r.trace = p.nls.J, [], None
    return r


def levenberg_marquardt(nls, solver, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, iterations=1000, delta=10.0, lower=(),
                        upper=(), ctx=None):
    """levenberg_marquardt.jl:39-144, statement by statement."""
    ctx = ctx or default_context()
    p = _Problem(nls, ctx)
    n, m = p.n, p.m
    dx, dtd = DeviceVector(ctx, n), DeviceVector(ctx, n)
    ftrial, fpredict = DeviceVector(ctx, m), DeviceVector(ctx, m)
    x, fcur, J = p.x, p.fcur, p.J
    lo, hi = _bounds(ctx, n, x, lower, upper)
    A = AllocatedSolver(J, solver, for_lm=True)
    decrease_factor = 2.0
    f_calls = g_calls = mul_calls = 0
    converged = x_c = f_c = g_c = False
    p.f_(fcur, x); f_calls += 1
    ssr = sumsq(fcur)
    maxabs_gr = float("inf")
    need_jacobian = True
    it = 0
    while not converged and it < iterations:
        it += 1
        bad = first_nonfinite(x)
        if bad >= 0:
            e = _lib.IsFiniteException(_lib.ENONFINITE, "non-finite x")
            e.indices = [bad]
            raise e
        if need_jacobian:
            p.g_(x); g_calls += 1
            need_jacobian = False
        colsumabs2_(dtd, J)                                           # :82
        dtd_mean = vsum(dtd) / n                                      # :84
        clamp_(dtd, MIN_DIAGONAL * dtd_mean, MAX_DIAGONAL * dtd_mean)
        rmul_(dtd, 1 / delta)                                         # :86
        dx, lmiter = A.ldiv_(dx, fcur, dtd)                           # :87
        box_clip_(dx, x, lo, hi)                                      # :89-98
        mul_calls += lmiter
        mul_(dtd, J, fcur, 1.0, 0.0, trans=True); mul_calls += 1      # :102
        maxabs_gr = maxabs_projected_gradient(dtd, x, lo, hi)
        axpy_(-1.0, dx, x)                                            # :106
        p.f_(ftrial, x); f_calls += 1
        trial_ssr = sumsq(ftrial)
        mul_(fpredict, J, dx, 1.0, 0.0); mul_calls += 1               # :114
        axpy_(-1.0, fcur, fpredict)
        predicted_ssr = sumsq(fpredict)
        predicted_reduction = abs(ssr - predicted_ssr)
        rho = _div(ssr - trial_ssr, predicted_reduction) if predicted_reduction > 0 else 0.0
        step_accepted = rho > MIN_STEP_QUALITY                        # :122
        x_c, f_c, g_c, converged = assess_convergence(dx, x, maxabs_gr, ssr, trial_ssr, x_tol, f_tol, g_tol, step_accepted)
        if step_accepted:
            copyto_(fcur, ftrial)
            ssr = trial_ssr
            t = 2.0 * rho - 1.0
            delta = min(delta / max(1 / 3, 1.0 - t * t * t), MAX_DELTA)   # :130 (products, not pow(): same bits as the C loop)
            decrease_factor = 2.0
            need_jacobian = True
        else:
            axpy_(1.0, dx, x)                                         # :135
            delta = max(delta / decrease_factor, MIN_DELTA)
            decrease_factor *= 2.0
    return _result("LevenbergMarquardt", p, ssr, it, converged, (x_c, f_c, g_c), (x_tol, f_tol, g_tol),
                   (f_calls, g_calls, mul_calls))


def dogleg(nls, solver, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, iterations=1000, delta=1.0, lower=(), upper=(), ctx=None):
    """dogleg.jl:41-203, statement by statement."""
    ctx = ctx or default_context()
    p = _Problem(nls, ctx)
    n, m = p.n, p.m
    dgn, dgr, dx, dtd = (DeviceVector(ctx, n) for _ in range(4))
    ftrial, fpredict = DeviceVector(ctx, m), DeviceVector(ctx, m)
    x, fcur, J = p.x, p.fcur, p.J
    lo, hi = _bounds(ctx, n, x, lower, upper)
    A = AllocatedSolver(J, solver, for_lm=False)
    reuse = False
    wnorm_dgn = wnorm_dgr = alpha = 0.0
    f_calls = g_calls = mul_calls = 0
    converged = x_c = f_c = g_c = False
    p.f_(fcur, x); f_calls += 1
    ssr = sumsq(fcur)
    maxabs_gr = float("inf")
    it = 0
    while not converged and it < iterations:
        it += 1
        bad = first_nonfinite(x)
        if bad >= 0:
            e = _lib.IsFiniteException(_lib.ENONFINITE, "non-finite x")
            e.indices = [bad]
            raise e
        if not reuse:
            p.g_(x); g_calls += 1
            colsumabs2_(dtd, J)                                       # :85
            clamp_(dtd, MIN_DIAGONAL, MAX_DIAGONAL)                   # :90
            if it == 1:
                wnorm_x = wnorm(x, dtd)
                if wnorm_x > 0:
                    delta *= wnorm_x                                  # :92-97
            mul_(dgr, J, fcur, 1.0, 0.0, trans=True); mul_calls += 1  # :99
            maxabs_gr = maxabs_projected_gradient(dgr, x, lo, hi)
            ediv_(dgr, dgr, dtd)                                      # :105
            wnorm_dgr = wnorm(dgr, dtd)
            mul_(fpredict, J, dgr, 1.0, 0.0); mul_calls += 1          # :109
            alpha = _div(wnorm_dgr * wnorm_dgr, sumsq(fpredict))      # :111
            fill_(dgn, 0.0)
            dgn, ls_iter = A.ldiv_(dgn, fcur)                         # :115
            mul_calls += ls_iter
            wnorm_dgn = wnorm(dgn, dtd)
        if wnorm_dgn <= delta:                                        # :120
            copyto_(dx, dgn)
            wnorm_dx = wnorm_dgn
        elif wnorm_dgr * alpha >= delta:                              # :124
            copyto_(dx, dgr)
            rmul_(dx, _div(delta, wnorm_dgr))
            wnorm_dx = delta
        else:                                                         # :131
            b_dot_a = alpha * wdot(dgr, dgn, dtd)
            a_squared_norm = (alpha * wnorm_dgr) * (alpha * wnorm_dgr)
            b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + wnorm_dgn * wnorm_dgn
            c = b_dot_a - a_squared_norm
            d = float(np.sqrt(c * c + b_minus_a_squared_norm * (delta * delta - a_squared_norm)))
            beta = _div(d - c, b_minus_a_squared_norm) if c <= 0 else _div(delta * delta - a_squared_norm, d + c)
            copyto_(dx, dgn)
            rmul_(dx, beta)
            axpy_(alpha * (1 - beta), dgr, dx)
            wnorm_dx = wnorm(dx, dtd)
        box_clip_(dx, x, lo, hi)                                      # :148-160
        axpy_(-1.0, dx, x)
        p.f_(ftrial, x); f_calls += 1
        trial_ssr = sumsq(ftrial)
        mul_(fpredict, J, dx, 1.0, 0.0); mul_calls += 1               # :171
        axpy_(-1.0, fcur, fpredict)
        predicted_ssr = sumsq(fpredict)
        predicted_reduction = abs(ssr - predicted_ssr)
        rho = _div(ssr - trial_ssr, predicted_reduction) if predicted_reduction > 0 else 0.0
        step_accepted = rho >= MIN_STEP_QUALITY                       # :178
        x_c, f_c, g_c, converged = assess_convergence(dx, x, maxabs_gr, ssr, trial_ssr, x_tol, f_tol, g_tol, step_accepted)
        if step_accepted:
            reuse = False
            copyto_(fcur, ftrial)
            ssr = trial_ssr
        else:
            reuse = True
            axpy_(1.0, dx, x)
        if rho < DECREASE_THRESHOLD:                                  # :193-197
            delta = max(MIN_DELTA, delta * 0.5)
        elif rho > INCREASE_THRESHOLD:
            delta = max(delta, 3.0 * wnorm_dx)
    return _result("Dogleg", p, ssr, it, converged, (x_c, f_c, g_c), (x_tol, f_tol, g_tol),
                   (f_calls, g_calls, mul_calls))


def optimize_operator_level(nls, optimizer=None, **kw):
    """optimize!(nls, optimizer) driven through the operator-level ABI only."""
    solver = default_solver(optimizer.solver if optimizer is not None else None, nls.J)
    optimizer = default_optimizer(optimizer, solver)
    if optimizer.kind == _lib.LEVENBERG_MARQUARDT:
        return levenberg_marquardt(nls, solver, **kw)
    return dogleg(nls, solver, **kw)
