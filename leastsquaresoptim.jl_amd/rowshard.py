"""ONE large problem with the Jacobian split by residual rows across ranks (one GPU each).

Rank p owns rows R_p of J (an m_p x n sparse matrix with the global column space) and the matching slices of the
residual vectors; every n-vector (x, dx, dtd, v, h, hbar, P, ...) is REPLICATED, and every rank runs the same scalar
control flow of LevenbergMarquardt(LSMR()) (levenberg_marquardt.jl:39-144, iterative_lsmr.jl:238-259, lsmr.jl:53-238).
What crosses ranks -- each item ONE sum all-reduce (RCCL over xGMI on the node; `gloo` in the CPU tests):

  per outer iteration   colsumabs2(J) = sum_p colsumabs2(J_p)   and the gradient  J'f = sum_p J_p'f_p      (2 n doubles, one call)
                        trial ssr = sum_p |f_p(x_trial)|^2  and  predicted ssr = sum_p |J_p dx - f_p|^2       (2 doubles, one call)
  per inner iteration   J'u = sum_p J_p'u_p  TOGETHER WITH  |u|^2 = sum_p |u_p|^2                            (n + 1 doubles, one call)

The last line is the point of the layout: u is kept unnormalised (as in the fused device kernels, lsq_lsmr.hip: 1/beta is
folded into the consumer), so the norm that the reference needs BEFORE the adjoint product (lsmr.jl:119-122) can travel
WITH it -- one collective of 80 KB + 8 B per inner iteration at C4's width instead of an 8-byte all-reduce followed by an
80 KB one.  On the device this is where k_combine's epilogue sits: it already adds the per-window partial sums of J'u and
runs the v-update on the result; the row-sharded variant adds the ranks' vectors at the same place.  J*v needs nothing
(rows are local), |v| and |x| are norms of replicated vectors.

ON THE DEVICE the whole loop is `lsq_optimize` itself with `lsq_options.row_allreduce` set (include/lsqhip.h): the rank's
row block goes through the same fused kernels as an unsharded problem, and the collectives above are sum all-reduces of
DEVICE buffers enqueued on the library's stream -- between the adjoint product and its epilogue for the inner iteration
(k_sell_cols -> k_combine into a buffer of n + 1 doubles, sum(u^2) in the last slot -> all-reduce -> EpiV + update): no host
round trip, no staging copy.  `RcclRowAllreduce` is that hook as a direct RCCL call (liblsqrccl.so: ncclAllReduce on the
stream); `HostStagedRowAllreduce` carries it over any torch.distributed group through the host (the world-2 tests: two ranks
cannot share one GPU under RCCL).  `optimize_device` below is the entry point.

Two backends run the SAME host-level driver (the executable specification of where the collectives sit): NumpyBackend (scipy CSR/CSC slices; the executable specification, used by the world-2
gloo test on CPU and compared with the oracle) and HipBackend (a DeviceMatrix per rank, every array operation a C-ABI
call on device memory; the n-vector exchange is staged through page-locked host memory).  Summation order differs from the
unsharded run only in the cross-rank sums.
"""
import numpy as np

MIN_DELTA, MAX_DELTA, MIN_STEP_QUALITY = 1e-16, 1e16, 1e-3            # types.jl:107-109
MIN_DIAGONAL, MAX_DIAGONAL = 1e-6, 1e32                               # types.jl:110-111


def row_slice(m, rank, world):
    """Contiguous, balanced row ranges."""
    lo = (m * rank) // world
    hi = (m * (rank + 1)) // world
    return lo, hi


class Comm:
    """Sum all-reduce of small float64 host arrays over a torch.distributed group (world 1: identity)."""

    def __init__(self, dist=None, group=None):
        self.dist, self.group = dist, group
        self.calls = 0
        self.doubles = 0

    def allreduce(self, a):
        self.calls += 1
        self.doubles += a.size
        if self.dist is None or self.dist.get_world_size(self.group) == 1:
            return a
        import torch
        t = torch.from_numpy(a)
        self.dist.all_reduce(t, group=self.group)
        return a


class NumpyBackend:
    """Local rows as scipy CSR (J*v) / CSC (J'u); f(x) -> local residual slice, g(x) -> local nzval (CSC order)."""

    def __init__(self, Jp_csc, f_local, g_local):
        self.J = Jp_csc.tocsc(copy=True)
        self.J.sort_indices()
        self.mp, self.n = self.J.shape
        self.f_local, self.g_local = f_local, g_local
        self._csr = None

    # n-vectors and local m-vectors are numpy arrays
    def zeros_n(self):
        return np.zeros(self.n)

    def zeros_m(self):
        return np.zeros(self.mp)

    def to_host(self, v):
        return v

    def from_host_n(self, h, out=None):
        if out is None:
            return h.copy()
        out[:] = h
        return out

    def f(self, out, x):
        self.f_local(out, x)

    def g(self, x):
        self.g_local(self.J.data, x)
        self._csr = self.J.tocsr()

    def colsumabs2(self):
        return np.asarray(self.J.multiply(self.J).sum(axis=0)).ravel()

    def mul(self, y, x, alpha, beta):          # y <- alpha J x + beta y
        y *= beta
        y += alpha * (self._csr @ x)

    def mulT_host(self, y):                    # J'y as a host vector
        return self.J.T @ y

    def sumsq(self, v):
        return float(np.dot(v, v))

    def axpy(self, a, x, y):
        y += a * x

    def copy(self, dst, src):
        dst[:] = src

    def residual_sumsq(self, Jdx, f):          # |J dx - f|^2 locally
        r = Jdx - f
        return float(np.dot(r, r))


class HipBackend:
    """Local rows on the device (DeviceMatrix of the rank's row slice); every operation is a C-ABI call on device memory."""

    def __init__(self, lsq, ctx, Jp_csc, f_local, g_local):
        self.lsq, self.ctx = lsq, ctx
        S = Jp_csc.tocsc(copy=True)
        S.sort_indices()
        self.host = S
        self.mp, self.n = S.shape
        self.J = lsq.DeviceMatrix(ctx, S)
        self.f_local, self.g_local = f_local, g_local
        self._xh, self._fh = np.zeros(self.n), np.zeros(self.mp)
        self._stage = lsq.PinnedBuffer(ctx, S.nnz)
        self._tmp_n = lsq.DeviceVector(ctx, self.n)
        self._tmp_m = lsq.DeviceVector(ctx, self.mp)

    def zeros_n(self):
        return self.lsq.DeviceVector(self.ctx, self.n)

    def zeros_m(self):
        return self.lsq.DeviceVector(self.ctx, self.mp)

    def to_host(self, v):
        return v.get()

    def from_host_n(self, h, out=None):
        if out is None:
            return self.lsq.DeviceVector(self.ctx, self.n, h)
        out.set(h)
        return out

    def f(self, out, x):
        self._xh[:] = x.get()
        self.f_local(self._fh, self._xh)
        out.set(self._fh)

    def g(self, x):
        self._xh[:] = x.get()
        self.J.upload_wait()
        self.g_local(self._stage.array, self._xh)
        self.J.set_values_async(self._stage)

    def colsumabs2(self):
        return self.lsq.colsumabs2_(self._tmp_n, self.J).get()

    def mul(self, y, x, alpha, beta):
        self.lsq.mul_(y, self.J, x, alpha, beta)

    def mulT_host(self, y):
        return self.lsq.mul_(self._tmp_n, self.J, y, 1.0, 0.0, trans=True).get()

    def sumsq(self, v):
        return self.lsq.sumsq(v)

    def axpy(self, a, x, y):
        self.lsq.axpy_(a, x, y)

    def copy(self, dst, src):
        self.lsq.copyto_(dst, src)

    def residual_sumsq(self, Jdx, f):
        self.lsq.copyto_(self._tmp_m, Jdx)
        self.lsq.axpy_(-1.0, f, self._tmp_m)
        return self.lsq.sumsq(self._tmp_m)


class Result:
    pass


def _lsmr_damped(B, comm, f, dtd_h, colsum_h, grad_h, ssr, maxiter):
    """ldiv!(x, J, y, damp, ::LSMR) for the row-sharded operator (iterative_lsmr.jl:238-259 + lsmr.jl:53-238).
    Replicated n-vectors live on the HOST here (they are 80 KB at C4 and every rank computes the same values); the local
    m-vector u and the products live with the backend.  Returns (dx (host), iterations)."""
    n = B.n
    atol, btol, ctol = 1e-6, 0.5, 1e-8
    # preconditioner and sqrt(damp) (iterative_lsmr.jl:129-141, :251-252)
    s = colsum_h + dtd_h
    P = np.where(s > 0.0, 1.0 / np.sqrt(np.where(s > 0.0, s, 1.0)), 0.0)
    dg = np.sqrt(dtd_h)
    x = np.zeros(n)
    ux = np.zeros(n)
    u = B.zeros_m()
    B.copy(u, f)                                   # u = b - A*0
    # beta_1 = |f| is the global ssr the loop already holds; A'b = P .* (J'f) is the gradient it already holds
    ny = np.sqrt(ssr)
    beta = np.sqrt(ny * ny + 0.0)
    cu = 1.0                                       # u is kept UNNORMALISED: u_true = u / beta_u
    beta_u = beta
    v = (grad_h / beta) * P if beta > 0 else np.zeros(n)
    alpha = float(np.sqrt(np.dot(v, v)))
    if alpha > 0:
        v = v / alpha
    zetabar, alphabar, rho, rhobar, cbar, sbar = alpha * beta, alpha, 1.0, 1.0, 1.0, 0.0
    h, hbar = v.copy(), np.zeros(n)
    betadd, betad, rhodold, tautildeold, thetatilde, zeta, d = beta, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0
    normA2, maxrbar, minrbar = alpha * alpha, 0.0, 1e100
    normb, normAr = beta, alpha * beta
    it = 0
    if normAr == 0:
        return x, 0
    t_dev = B.zeros_n()
    payload = np.zeros(n + 1)
    while it < maxiter:
        it += 1
        t = v * P                                              # iterative_lsmr.jl:31
        B.from_host_n(t, t_dev)
        # u~ <- J t - (alpha / beta_u) u~    (u~ = beta_u * u_true: lsmr.jl:118 with the 1/beta of :121 folded in)
        B.mul(u, t_dev, 1.0, -alpha / beta_u if beta_u > 0 else -alpha)
        ux = t * dg - (alpha / beta_u if beta_u > 0 else alpha) * ux      # damped rows (iterative_lsmr.jl:92)
        # ---- THE collective of the inner iteration: J'u~ and |u~|^2 in one all-reduce of n + 1 doubles ----
        payload[:n] = B.mulT_host(u)
        payload[n] = B.sumsq(u)
        comm.allreduce(payload)
        nu, nx = np.sqrt(payload[n]), float(np.sqrt(np.dot(ux, ux)))
        beta = float(np.sqrt(nu * nu + nx * nx))               # DampenedVector norm, iterative_lsmr.jl:72
        beta_u = beta
        if beta > 0:
            w = (payload[:n] + ux * dg) / beta                 # A'u with u = u~ / beta (iterative_lsmr.jl:106-107)
            v = w * P - beta * v                               # :41-49, lsmr.jl:122
            alpha = float(np.sqrt(np.dot(v, v)))
            if alpha > 0:
                v = v / alpha
        # rotations and estimates (lsmr.jl:127-196), lambda = 0
        alphahat, chat, shat = alphabar, 1.0, 0.0
        rhoold = rho
        rho = float(np.sqrt(alphahat * alphahat + beta * beta))
        c, sn = alphahat / rho, beta / rho
        thetanew = sn * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar, rhotemp = sbar * rho, cbar * rho
        rhobar = float(np.sqrt((cbar * rho) ** 2 + thetanew ** 2))
        cbar = cbar * rho / rhobar
        sbar = thetanew / rhobar
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        hbar = hbar * (-thetabar * rho / (rhoold * rhobarold)) + h
        x = x + (zeta / (rho * rhobar)) * hbar
        h = h * (-thetanew / rho) + v
        betaacute, betacheck = chat * betadd, -shat * betadd
        betahat = c * betaacute
        betadd = -sn * betaacute
        thetatildeold = thetatilde
        rhotildeold = float(np.sqrt(rhodold ** 2 + thetabar ** 2))
        ctildeold, stildeold = rhodold / rhotildeold, thetabar / rhotildeold
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck ** 2
        normr = float(np.sqrt(d + (betad - taud) ** 2 + betadd ** 2))
        normA2 = normA2 + beta * beta
        normA = float(np.sqrt(normA2))
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if it > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normAr = abs(zetabar)
        normx = float(np.sqrt(np.dot(x, x)))
        test1, test2, test3 = normr / normb, normAr / (normA * normr), 1.0 / condA
        t1 = test1 / (1.0 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if it >= maxiter or 1.0 + test3 <= 1.0 or 1.0 + test2 <= 1.0 or 1.0 + t1 <= 1.0 or test3 <= ctol or test2 <= atol \
                or test1 <= rtol:
            break
    return x * P, it                                           # iterative_lsmr.jl:256-257


def lm_lsmr(B, comm, x0, m_total, iterations=1000, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, delta=10.0):
    """optimize!(nls, LevenbergMarquardt(LSMR())) on the row-sharded problem.  x0: host vector (identical on every rank)."""
    n = B.n
    x_h = np.array(x0, dtype=np.float64)
    x_dev, xt_dev, dx_dev = B.from_host_n(x_h), B.zeros_n(), B.zeros_n()
    fcur, ftrial, fpred = B.zeros_m(), B.zeros_m(), B.zeros_m()
    decrease_factor = 2.0
    B.f(fcur, x_dev)
    ssr = float(comm.allreduce(np.array([B.sumsq(fcur)]))[0])
    f_calls, g_calls, mul_calls, inner_total = 1, 0, 0, 0
    need_jac, converged, it = True, False, 0
    xc = fc = gc = False
    maxabs_gr = np.inf
    colsum = grad = None
    while not converged and it < iterations:
        it += 1
        if not np.all(np.isfinite(x_h)):
            raise FloatingPointError("non-finite x at index %d" % int(np.argmax(~np.isfinite(x_h))))
        if need_jac:
            B.g(x_dev)
            g_calls += 1
            need_jac = False
        # colsumabs2(J) and J'f in ONE all-reduce of 2 n doubles (levenberg_marquardt.jl:82, :102)
        pay = np.concatenate([B.colsumabs2(), B.mulT_host(fcur)])
        comm.allreduce(pay)
        colsum, grad = pay[:n].copy(), pay[n:].copy()
        mul_calls += 1
        mean = float(np.sum(colsum)) / n
        dtd = np.clip(colsum, MIN_DIAGONAL * mean, MAX_DIAGONAL * mean) / delta
        dx_h, lmiter = _lsmr_damped(B, comm, fcur, dtd, colsum, grad, ssr, m_total + n)
        mul_calls += 2 * lmiter
        inner_total += lmiter
        maxabs_gr = float(np.max(np.abs(grad)))
        x_trial = x_h - dx_h
        B.from_host_n(x_trial, xt_dev)
        B.from_host_n(dx_h, dx_dev)
        B.f(ftrial, xt_dev)
        f_calls += 1
        B.mul(fpred, dx_dev, 1.0, 0.0)                         # J dx
        mul_calls += 1
        sc = comm.allreduce(np.array([B.sumsq(ftrial), B.residual_sumsq(fpred, fcur)]))
        trial_ssr, predicted_ssr = float(sc[0]), float(sc[1])
        pr = abs(ssr - predicted_ssr)
        rho = (ssr - trial_ssr) / pr if pr > 0 else 0.0
        accepted = rho > MIN_STEP_QUALITY
        xc = fc = gc = False
        if accepted and abs(trial_ssr - ssr) <= f_tol * (abs(ssr) + f_tol):
            fc = True
        elif float(np.max(np.abs(dx_h))) <= x_tol:
            xc = True
        elif maxabs_gr <= g_tol:
            gc = True
        converged = xc or fc or gc
        if accepted:
            fcur, ftrial = ftrial, fcur
            x_h = x_trial
            x_dev, xt_dev = xt_dev, x_dev
            ssr = trial_ssr
            q = 1.0 - (2.0 * rho - 1.0) ** 3
            delta = min(delta / max(1.0 / 3.0, q), MAX_DELTA)
            decrease_factor = 2.0
            need_jac = True
        else:
            x_h = x_trial + dx_h                                # the reference restores x as (x - dx) + dx (:135)
            B.from_host_n(x_h, x_dev)
            delta = max(delta / decrease_factor, MIN_DELTA)
            decrease_factor *= 2.0
    r = Result()
    r.minimizer, r.ssr, r.iterations, r.converged = x_h, ssr, it, converged
    r.x_converged, r.f_converged, r.g_converged = xc, fc, gc
    r.f_calls, r.g_calls, r.mul_calls, r.lsmr_iterations = f_calls, g_calls, mul_calls, inner_total
    r.allreduce_calls, r.allreduce_doubles = comm.calls, comm.doubles
    return r


# ------------------------------------------------------------------------------------------------------------------------
# device-resident row-sharded runs: lsq_optimize with lsq_options.row_allreduce
# ------------------------------------------------------------------------------------------------------------------------
class RcclRowAllreduce:
    """The row all-reduce as a direct RCCL call on the library's stream (include/lsqrccl.h, liblsqrccl.so).

    One communicator per instance; `dist` (any initialised torch.distributed group, or None for a one-rank world) only
    carries the 128-byte unique id from rank 0 to the others.  The HIP device must be current (lsq.Context(device) /
    torch.cuda.set_device) before construction."""

    def __init__(self, rank=0, world=1, dist=None, librccl=None):
        import ctypes as C
        import os
        from . import _lib
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "liblsqrccl.so")
        if not os.path.exists(path):
            raise RuntimeError("liblsqrccl.so is not built (make -C leastsquaresoptim.jl_amd/csrc)")
        L = C.CDLL(path)
        L.lsq_rccl_last_error.restype = C.c_char_p
        L.lsq_rccl_load.argtypes = [C.c_char_p]
        L.lsq_rccl_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.lsq_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.lsq_rccl_allreduce_callback.restype = C.c_void_p
        L.lsq_rccl_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        if librccl is None:
            # ONE RCCL and ONE HIP runtime per process: if PyTorch is already in the process its bundled librccl (and the HIP
            # runtime both libraries then share) is the one to bind; otherwise ROCm's own, next to the libamdhip64 that
            # liblsqhip.so was linked against.  (Binding PyTorch's copy into a process that runs on ROCm's HIP runtime makes
            # ncclCommInitRank fail with "unhandled cuda error": two runtimes, two notions of the current device.)
            import sys
            cands = []
            if "torch" in sys.modules:
                cands.append(os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so"))
            cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librccl.so")]
            librccl = next((c_ for c_ in cands if os.path.exists(c_)), None)
        if L.lsq_rccl_load(librccl.encode() if librccl else None) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        idbuf = C.create_string_buffer(128)
        if rank == 0 and L.lsq_rccl_unique_id(idbuf) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        if world > 1:
            box = [idbuf.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            idbuf = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        if L.lsq_rccl_comm_create(idbuf, rank, world, C.byref(h)) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        self._L, self.comm, self.rank, self.world = L, h, rank, world
        self.callback = C.cast(L.lsq_rccl_allreduce_callback(), _lib.ROW_ALLREDUCE_CALLBACK)
        self.user = h

    def stats(self):
        import ctypes as C
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._L.lsq_rccl_comm_stats(self.comm, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if self.comm:
            self._L.lsq_rccl_comm_destroy(self.comm)
            self.comm = None


class HostStagedRowAllreduce:
    """The same hook over ANY torch.distributed group (gloo in the tests), staged through the host: drain the stream, copy
    the buffer out, all-reduce, copy it back.  For worlds RCCL cannot form (two ranks on one GPU) -- not for speed."""

    def __init__(self, ctx, dist=None, group=None):
        import ctypes as C
        from . import _lib
        self.calls = self.doubles = 0
        L = _lib.lib()

        def cb(d_buf, count, _stream, _user):
            try:
                h = np.empty(count)
                _lib.check(L.lsq_d2h(ctx.h, h.ctypes.data_as(C.c_void_p), d_buf, count * 8))     # (synchronises the stream)
                if dist is not None and dist.get_world_size(group) > 1:
                    import torch
                    dist.all_reduce(torch.from_numpy(h), group=group)
                _lib.check(L.lsq_h2d(ctx.h, d_buf, h.ctypes.data_as(C.c_void_p), count * 8))
                self.calls += 1
                self.doubles += count
                return 0
            except Exception as e:   # pragma: no cover
                import sys
                print("row all-reduce failed:", e, file=sys.stderr)
                return 1

        self.callback = _lib.ROW_ALLREDUCE_CALLBACK(cb)
        self.user = None

    def stats(self):
        return self.calls, self.doubles

    def close(self):
        pass


def optimize_device(problem, hook, global_rows, iterations=1000, x_tol=1e-8, f_tol=1e-8, g_tol=1e-8, delta=None, trace=False):
    """optimize!(nls, LevenbergMarquardt(LSMR())) for the row block held by `problem` (a synthetic.TanhProblem built on this
    rank's rows of A and b, or anything with the same .optimize) with the cross-rank sums done by `hook`
    (RcclRowAllreduce / HostStagedRowAllreduce).  Every rank gets the same replicated minimizer and scalars."""
    from . import _lib
    return problem.optimize(_lib.LEVENBERG_MARQUARDT, _lib.LSMR, x_tol=x_tol, f_tol=f_tol, g_tol=g_tol, iterations=iterations,
                            delta=delta, trace=trace, row_allreduce=hook.callback, row_allreduce_user=hook.user,
                            global_rows=global_rows)
