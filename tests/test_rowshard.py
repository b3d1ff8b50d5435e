"""one problem with J split by residual rows across ranks (leastsquaresoptim.jl_amd/rowshard.py).
CPU: world-2 `gloo`, numpy backend (the executable specification of where the collectives sit) against the unsharded
oracle.  GPU box: two ranks on the one device, every array operation a C-ABI call, against the unsharded device run."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch.distributed as dist
import torch.multiprocessing as mp

import lsq_amd as lsq
from lsq_amd import rowshard as RS
from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(m=6000, n=60, pc=40, seed=0):
    rng = np.random.default_rng(seed)
    rowval = np.concatenate([np.sort(rng.choice(m, pc, replace=False)) for _ in range(n)]).astype(np.int32)
    colptr = (np.arange(n + 1) * pc).astype(np.int32)
    A = rng.standard_normal(n * pc) / np.sqrt(pc)
    S = sp.csc_matrix((A, rowval, colptr), shape=(m, n))
    b = S @ np.tanh(rng.uniform(-1, 1, n)) + 1e-3 * rng.standard_normal(m)
    return m, n, colptr, rowval, A, S, b


def _local(S, b, rank, world):
    """Row slice of the tanh model r(x) = A tanh(x) - b, J = A diag(1 - tanh(x)^2), as callbacks on local arrays."""
    m = S.shape[0]
    lo, hi = RS.row_slice(m, rank, world)
    Ap = S.tocsr()[lo:hi].tocsc()
    Ap.sort_indices()
    cols = np.repeat(np.arange(Ap.shape[1]), np.diff(Ap.indptr))
    bp = b[lo:hi]

    def f(out, x):
        out[:] = Ap @ np.tanh(x) - bp

    def g(vals, x):
        vals[:] = Ap.data * (1.0 - np.tanh(x) ** 2)[cols]

    return Ap, f, g


def _oracle(m, n, colptr, rowval, A, b, iterations):
    Am = O.Mat(csc=(m, n, colptr, rowval, A))
    J = O.Mat(csc=(m, n, colptr, rowval, np.zeros_like(A)))
    f, g, ud, keep = O.tanh_model(Am, b)
    return O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=iterations, trace=True, trace_x=False)


def _worker(rank, world, port, q, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n, colptr, rowval, A, S, b = _problem() if backend == "numpy" else _problem(120000, 400, 300, 3)
    Ap, f, g = _local(S, b, rank, world)
    comm = RS.Comm(dist)
    if backend == "numpy":
        B = RS.NumpyBackend(Ap, f, g)
    else:
        B = RS.HipBackend(lsq, lsq.Context(0), Ap, f, g)
    r = RS.lm_lsmr(B, comm, np.zeros(n), m, iterations=40)
    q.put((rank, r.iterations, r.lsmr_iterations, r.ssr, r.converged, r.f_calls, r.g_calls, r.mul_calls, r.minimizer,
           r.allreduce_calls, r.allreduce_doubles))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(backend):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(2):
        rec = q.get(timeout=600)
        res[rec[0]] = rec[1:]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_single_rank_equals_oracle():
    m, n, colptr, rowval, A, S, b = _problem()
    Ap, f, g = _local(S, b, 0, 1)
    r = RS.lm_lsmr(RS.NumpyBackend(Ap, f, g), RS.Comm(None), np.zeros(n), m, iterations=40)
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    assert (r.iterations, r.f_calls, r.g_calls, r.mul_calls) == (ro.iterations, ro.f_calls, ro.g_calls, ro.mul_calls)
    assert r.lsmr_iterations == int(ro.trace["inner"].sum()) // 2 and r.converged == ro.converged
    assert r.ssr == pytest.approx(ro.ssr, rel=1e-10) and np.max(np.abs(r.minimizer - ro.minimizer)) <= 1e-8


def test_row_sharded_gloo_world2_equals_oracle():
    """Both ranks walk the same trajectory as the unsharded oracle (cross-rank sums differ in the last bits only), and the
    collective count is what the design promises: per outer iteration 1 (colsumabs2 + gradient) + 1 (trial / predicted
    ssr), per inner iteration 1 (J'u with |u|^2 riding along), plus the initial ssr."""
    res = _run_world2("numpy")
    m, n, colptr, rowval, A, S, b = _problem()
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    inner = int(ro.trace["inner"].sum()) // 2
    for rank in (0, 1):
        it, li, ssr, conv, fcalls, gcalls, mulc, x, calls, dbl = res[rank]
        assert (it, fcalls, gcalls, mulc, li, conv) == (ro.iterations, ro.f_calls, ro.g_calls, ro.mul_calls, inner, ro.converged)
        assert ssr == pytest.approx(ro.ssr, rel=1e-10) and np.max(np.abs(x - ro.minimizer)) <= 1e-8
        assert calls == 1 + 2 * it + li and dbl == 1 + it * (2 * n + 2) + li * (n + 1)
    assert np.array_equal(res[0][7], res[1][7])          # replicated n-vectors: bit-identical on every rank


@pytest.mark.gpu
def test_row_sharded_on_device_world2():
    """The same driver with every array operation on the device (two ranks share the box's one GPU; exchange over gloo):
    same trajectory as the unsharded device-resident run of the same problem."""
    res = _run_world2("hip")
    m, n, colptr, rowval, A, S, b = _problem(120000, 400, 300, 3)
    ctx = lsq.Context(0)
    J = sp.csc_matrix((np.zeros_like(A), rowval, colptr), shape=(m, n))
    cols = np.repeat(np.arange(n), np.diff(colptr))
    nls = lsq.LeastSquaresProblem(x=np.zeros(n), y=np.zeros(m), f_=lambda out, x: out.__setitem__(slice(None), S @ np.tanh(x) - b),
                                  g_=lambda Jm, x: np.multiply(A, (1.0 - np.tanh(x) ** 2)[cols], out=Jm.data), J=J)
    r1 = lsq.optimize_(nls, lsq.LevenbergMarquardt(lsq.LSMR()), iterations=40, ctx=ctx)
    for rank in (0, 1):
        it, li, ssr, conv, fcalls, gcalls, mulc, x, calls, dbl = res[rank]
        assert (it, fcalls, gcalls, mulc, conv) == (r1.iterations, r1.f_calls, r1.g_calls, r1.mul_calls, r1.converged)
        assert ssr == pytest.approx(r1.ssr, rel=1e-9) and np.max(np.abs(x - r1.minimizer)) <= 1e-7
