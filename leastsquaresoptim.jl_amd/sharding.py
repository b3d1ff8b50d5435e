"""The one global exchange of sharded runs: independent problems, one per rank, and a
single all-reduce per OUTER iteration carrying {sum of ssr, max of gradient norms, all-converged}.

A mixed sum/max/min reduction is packed into ONE SUM all-reduce of world+3 doubles: slot 0 = ssr,
slot 1 = converged count, slot 2 = count of ranks that are LEAVING WITH AN ERROR, slot 3+rank = this
rank's gradient norm (zeros elsewhere), so every rank recovers the max locally.  Backend `nccl` is RCCL
over xGMI on MI355X; `gloo` is used by the CPU tests.  The payload is 8*(world+3) bytes: latency-only,
never inside the LSMR loop.

Error protocol (include/lsqhip.h): the loop calls the hook a last time with converged = -1 when it leaves
with an error; every rank that sees a non-zero abort count gets return code 2 (-> LSQ_ERCCL) and issues
no further collective, so all ranks issue the same number of all-reduces and nobody waits forever.
"""
import datetime
import os
import sys
import time

from . import _lib


NSLOT = 3   # ssr, converged count, abort count; then one gradient-norm slot per rank


def exchange(dist, rank, world, ssr, gnorm, converged, buf, host=None):
    """Returns (sum ssr, max gnorm, all converged).  `buf` is a float64 tensor of world+NSLOT elements on
    the backend's device; `host` an optional (pinned) staging tensor of the same shape."""
    import torch
    if host is None:
        host = torch.zeros(world + NSLOT, dtype=torch.float64)
    host.zero_()
    host[0] = ssr
    host[1] = 1.0 if converged else 0.0
    host[NSLOT + rank] = gnorm
    buf.copy_(host)
    dist.all_reduce(buf)
    host.copy_(buf)
    return float(host[0]), float(host[NSLOT:].max()), bool(float(host[1]) >= world - 0.5)


_DRAINS = []


def drain_all():
    """Completes exchanges that active ranks left in flight (call before tearing the process group down)."""
    for d in _DRAINS:
        d()


def make_allreduce_callback(dist, rank, world, device, group=None):
    """ctypes callback for lsq_options.allreduce (vals = {ssr, maxabs_gr, converged}).

    Called once per outer iteration by the loop kernels' host side (for active ranks: after g!, the
    gradient pass and the damping have been queued).  Every call issues exactly one all-reduce, so the
    collective sequence matches across ranks whatever their state.  What is returned differs:

    * a rank that reports `converged` (a frozen rank, which acts on "all converged") waits for its
      exchange and gets this iteration's global values;
    * an ACTIVE rank never acts on the result inside the loop ("all converged" cannot be true while it
      is not converged itself), so its all-reduce is left in flight and the call returns the values of
      the PREVIOUS exchange: the latency of the collective overlaps the whole iteration.  (Its
      reported global ssr is therefore one iteration old while it is active.)

    The staging buffers are written through numpy views (no tensor indexing ops)."""
    import torch
    on_gpu = str(device).startswith("cuda")
    bufs = [torch.zeros(world + NSLOT, dtype=torch.float64, device=device) for _ in range(2)]
    hosts = [torch.zeros(world + NSLOT, dtype=torch.float64) for _ in range(2)]
    if on_gpu:
        hosts = [h.pin_memory() for h in hosts]
    views = [h.numpy() for h in hosts]   # share memory with the (pinned) staging tensors
    state = {"work": None, "slot": 0}
    done_ev = torch.cuda.Event() if on_gpu else None
    # staging copies and the collective are issued from a stream of their own, never torch's default (null) stream: a copy
    # on the null stream completed ~130 us after it was issued while the LM loop's kernels were queued (measured: the
    # exchange cost 9-11 % of the C4 rate at one rank that way, ~2 % from a side stream)
    xstream = torch.cuda.Stream(device=device) if on_gpu else None

    # a peer that died without its farewell must not hang this rank forever: the wait for an exchange is bounded
    # (LSQ_EXCHANGE_TIMEOUT_S, default 120 s) and a timeout surfaces as a failed hook (-> LSQ_ECALLBACK, no further call)
    timeout_s = float(os.environ.get("LSQ_EXCHANGE_TIMEOUT_S", "120"))

    def _finish(work, k):
        if on_gpu:
            work.wait()      # (RCCL: orders this stream behind the collective; returns at once)
            hosts[k].copy_(bufs[k], non_blocking=True)
            # (polled, not stream.synchronize(): the blocking wait's wake-up costs ~50 us of host time, and this thread
            #  is the one that feeds the LM loop's launches)
            done_ev.record()
            deadline = None
            spins = 0
            while not done_ev.query():
                spins += 1
                if spins & 0xFFFF == 0:
                    now = time.monotonic()
                    deadline = deadline or now + timeout_s
                    if now > deadline:
                        raise TimeoutError("exchange not completed after %.0f s (a peer rank is gone?)" % timeout_s)
        else:
            if not work.wait(datetime.timedelta(seconds=timeout_s)):
                raise TimeoutError("exchange not completed after %.0f s (a peer rank is gone?)" % timeout_s)
            hosts[k].copy_(bufs[k])
        hv = views[k]
        if hv[2] > 0.5:
            state["aborted"] = True
        return float(hv[0]), float(hv[NSLOT:].max()), 1.0 if hv[1] >= world - 0.5 else 0.0

    def _cb(vals, count, _user):
        if xstream is None:
            return _cb_body(vals, count, _user)
        with torch.cuda.stream(xstream):
            return _cb_body(vals, count, _user)

    def _cb_body(vals, count, _user):
        try:
            if state.get("aborted"):
                return 2
            k = state["slot"]
            prev, pk = state["work"], k ^ 1
            # the exchange left in flight an iteration ago is complete by now: look at it BEFORE issuing the next
            # one, so that a rank that learns of an abort issues no further collective
            if prev is not None:
                state["last"] = _finish(prev, pk)
                state["work"] = None
                if state.get("aborted"):
                    return 2
            leaving = vals[2] < -0.5
            conv = vals[2] > 0.5
            hv = views[k]
            hv[:] = 0.0
            hv[0] = 0.0 if leaving else vals[0]
            hv[1] = 1.0 if conv else 0.0
            hv[2] = 1.0 if leaving else 0.0
            hv[NSLOT + rank] = 0.0 if leaving else vals[1]
            bufs[k].copy_(hosts[k], non_blocking=True)
            work = dist.all_reduce(bufs[k], async_op=True, group=group)
            if leaving or conv or state.get("last") is None:
                res = _finish(work, k)           # frozen / leaving / first call: synchronous
            else:
                res = state["last"]              # active rank: this iteration's exchange stays in flight
                state["work"] = work
            state["last"] = res
            state["slot"] = k ^ 1
            if state.get("aborted"):
                return 0 if leaving else 2
            vals[0], vals[1], vals[2] = res
            return 0
        except Exception as e:  # pragma: no cover
            print("allreduce callback failed:", e, file=sys.stderr)
            return 1

    def _drain():
        if state["work"] is not None:
            if xstream is None:
                state["last"] = _finish(state["work"], state["slot"] ^ 1)
            else:
                with torch.cuda.stream(xstream):
                    state["last"] = _finish(state["work"], state["slot"] ^ 1)
            state["work"] = None

    _DRAINS.append(_drain)
    return _lib.ALLREDUCE_CALLBACK(_cb)


# ------------------------------------------------------------------------------------------------------------------------
# the same exchange served in C (liblsqrccl.so: lsq_rccl_xchg_*, include/lsqrccl.h) -- what bench.py --gpus N uses and what a
# Julia host binds with ccall.  The Python hook above stays as its test double: tests/test_sharding.py
# runs both over gloo (world 2) on the same made-up scalars and asserts the same sequence of results and return codes.
# ------------------------------------------------------------------------------------------------------------------------
_RL = None
XCHG_ISSUE_FN = None
XCHG_FINISH_FN = None


def rccl_shim():
    """ctypes handle of liblsqrccl.so with the signatures of include/lsqrccl.h."""
    global _RL, XCHG_ISSUE_FN, XCHG_FINISH_FN
    if _RL is not None:
        return _RL
    import ctypes as C
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "liblsqrccl.so")
    if not os.path.exists(path):
        raise RuntimeError("liblsqrccl.so is not built (make -C leastsquaresoptim.jl_amd/csrc)")
    L = C.CDLL(path)
    vp, i, pvp, pll = C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)
    XCHG_ISSUE_FN = C.CFUNCTYPE(i, C.POINTER(C.c_double), i, i, vp)
    XCHG_FINISH_FN = C.CFUNCTYPE(i, i, vp)
    L.lsq_rccl_last_error.restype = C.c_char_p
    L.lsq_rccl_load.argtypes = [C.c_char_p]
    L.lsq_rccl_unique_id.argtypes = [C.c_char_p]
    L.lsq_rccl_comm_create.argtypes = [C.c_char_p, i, i, pvp]
    L.lsq_rccl_comm_destroy.argtypes = [vp]
    L.lsq_rccl_comm_stats.argtypes = [vp, pll, pll]
    L.lsq_rccl_xchg_create.argtypes = [vp, i, i, pvp]
    L.lsq_rccl_xchg_create_custom.argtypes = [i, i, XCHG_ISSUE_FN, XCHG_FINISH_FN, vp, pvp]
    L.lsq_rccl_xchg_destroy.argtypes = [vp]
    L.lsq_rccl_xchg_drain.argtypes = [vp]
    L.lsq_rccl_xchg_reset.argtypes = [vp]
    L.lsq_rccl_xchg_callback.restype = vp
    L.lsq_rccl_xchg_stats.argtypes = [vp, pll, pll, C.POINTER(i)]
    _RL = L
    return L


def _librccl_path():
    # ONE RCCL and ONE HIP runtime per process (see rowshard.RcclRowAllreduce): PyTorch's bundled copy if PyTorch is loaded
    cands = []
    if "torch" in sys.modules:
        cands.append(os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so"))
    cands.append(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librccl.so"))
    return next((c for c in cands if os.path.exists(c)), None)


class _XchgBase:
    """Common part: .callback / .user go into lsq_options.allreduce / allreduce_user (api._run_native takes the object)."""

    def _bind(self, L, h):
        import ctypes as C
        self._L, self.h = L, h
        self.callback = C.cast(L.lsq_rccl_xchg_callback(), _lib.ALLREDUCE_CALLBACK)
        self.user = h
        _DRAINS.append(self.drain)

    def drain(self):
        if self.h and self._L.lsq_rccl_xchg_drain(self.h) != 0:
            raise RuntimeError("exchange: " + self._L.lsq_rccl_last_error().decode())

    def reset(self):
        """Before another run on the same handle (lsq_rccl_xchg_reset: the protocol state is per run); api._run_native calls it."""
        if self.h and self._L.lsq_rccl_xchg_reset(self.h) != 0:
            raise RuntimeError("exchange: " + self._L.lsq_rccl_last_error().decode())

    def stats(self):
        import ctypes as C
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        self._L.lsq_rccl_xchg_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"collectives": a.value, "synchronous": b.value, "aborted": bool(c.value)}

    def __call__(self, vals, count, user=None):       # (so that tests can drive it like the Python hook)
        return self.callback(vals, count, self.user)

    def close(self):
        if self.h:
            if self.drain in _DRAINS:
                _DRAINS.remove(self.drain)
            self._L.lsq_rccl_xchg_destroy(self.h)
            self.h = None


class RcclScalarExchange(_XchgBase):
    """lsq_options.allreduce as a direct RCCL call from C (lsq_rccl_xchg_create): one communicator of `world` ranks over the
    current HIP device; `dist` (any initialised torch.distributed group, or None when world == 1) only carries the 128-byte
    unique id from rank 0 to the others.  The HIP device must be current before construction."""

    def __init__(self, rank=0, world=1, dist=None, librccl=None):
        import ctypes as C
        L = rccl_shim()
        librccl = librccl or _librccl_path()
        if L.lsq_rccl_load(librccl.encode() if librccl else None) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        idbuf = C.create_string_buffer(128)
        if rank == 0 and L.lsq_rccl_unique_id(idbuf) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        if world > 1:
            box = [idbuf.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            idbuf = C.create_string_buffer(box[0], 128)
        comm = C.c_void_p()
        if L.lsq_rccl_comm_create(idbuf, rank, world, C.byref(comm)) != 0:
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        self.comm, self.rank, self.world = comm, rank, world
        h = C.c_void_p()
        if L.lsq_rccl_xchg_create(comm, rank, world, C.byref(h)) != 0:
            L.lsq_rccl_comm_destroy(comm)
            raise RuntimeError("RCCL: " + L.lsq_rccl_last_error().decode())
        self._bind(L, h)

    def comm_stats(self):
        import ctypes as C
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._L.lsq_rccl_comm_stats(self.comm, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        super().close()
        if self.comm:
            self._L.lsq_rccl_comm_destroy(self.comm)
            self.comm = None


class TorchTransportExchange(_XchgBase):
    """The C protocol (lsq_rccl_xchg_create_custom) over a torch.distributed group as its transport: issue =
    dist.all_reduce(async_op=True) on a tensor that aliases the C side's staging buffer, finish = work.wait().  CPU tests
    (gloo); `world == 1` without a process group completes at once."""

    def __init__(self, dist, rank, world, group=None):
        import ctypes as C
        import numpy as np
        import torch
        L = rccl_shim()
        works = [None, None]
        timeout_s = float(os.environ.get("LSQ_EXCHANGE_TIMEOUT_S", "120"))

        def issue(h_buf, count, slot, _user):
            try:
                if world > 1:
                    t = torch.from_numpy(np.ctypeslib.as_array(h_buf, (count,)))
                    works[slot] = (dist.all_reduce(t, async_op=True, group=group), t)
                return 0
            except Exception as e:  # pragma: no cover
                print("exchange transport (issue) failed:", e, file=sys.stderr)
                return 1

        def finish(slot, _user):
            try:
                w = works[slot]
                works[slot] = None
                if w is not None and not w[0].wait(datetime.timedelta(seconds=timeout_s)):
                    return 1
                return 0
            except Exception as e:  # pragma: no cover
                print("exchange transport (finish) failed:", e, file=sys.stderr)
                return 1

        self._keep = (XCHG_ISSUE_FN(issue), XCHG_FINISH_FN(finish))
        h = C.c_void_p()
        if L.lsq_rccl_xchg_create_custom(rank, world, self._keep[0], self._keep[1], None, C.byref(h)) != 0:
            raise RuntimeError("exchange: " + L.lsq_rccl_last_error().decode())
        self._bind(L, h)
