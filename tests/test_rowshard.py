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
        rec = q.get(timeout=240)
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


def _local_rows_csc(S, b, rank, world):
    """This rank's row block of A as the (colptr, rowval, nzval) triple TanhProblem takes, and its slice of b."""
    lo, hi = RS.row_slice(S.shape[0], rank, world)
    Ap = S.tocsr()[lo:hi].tocsc()
    Ap.sort_indices()
    return (hi - lo, Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32), np.ascontiguousarray(Ap.data)), b[lo:hi]


def _check_against_oracle(ro, rec, n, lookahead=2):
    """Same trajectory as the UNSHARDED ORACLE (counts, accept pattern, inner counts; ssr to 1e-10, iterates to 1e-8), and the
    collective count the design promises: 1 (initial ssr) + per outer iteration 1 (gradient) + 1 (trial, predicted ssr) +
    1 per g! (colsumabs2) + per ENQUEUED inner iteration 1 (n + 1 doubles) -- inner iterations are enqueued in chunks of the
    look-ahead so that every rank issues the same number whatever its host's timing."""
    it, inner, ssr, conv, fcalls, gcalls, mulc, x, calls, dbl, tr_inner = rec
    assert (it, fcalls, gcalls, mulc, conv) == (ro.iterations, ro.f_calls, ro.g_calls, ro.mul_calls, ro.converged)
    assert np.array_equal(tr_inner, ro.trace["inner"]) and inner == int(ro.trace["inner"].sum()) // 2
    assert ssr == pytest.approx(ro.ssr, rel=1e-10) and np.max(np.abs(x - ro.minimizer)) <= 1e-8
    enq = sum(-(-(k // 2) // lookahead) * lookahead for k in ro.trace["inner"])
    assert calls == 1 + 2 * it + gcalls + enq, (calls, it, gcalls, enq)
    assert dbl == 1 + it * (n + 2) + gcalls * n + enq * (n + 1)


BIG = (300000, 2000, 600, 4)     # nnz 1.2e6, m > 131072: the sliced layouts and the column-scaled (never multiplied out) J
MID = (120000, 400, 300, 3)      # segment kernels, g! multiplies J out


def _worker_device(rank, world, port, q, mode, shape, device_per_rank=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL between processes needs it on this driver
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # (carries RCCL's unique id only)
    m, n, colptr, rowval, A, S, b = _problem(*shape)
    (mp_, cp, rv, nz), bp = _local_rows_csc(S, b, rank, world)
    ctx = lsq.Context(rank if device_per_rank else 0)
    pr = lsq.synthetic.TanhProblem(mp_, n, sparse=True, ctx=ctx, inputs=(cp, rv, nz), b=bp)
    pr.reset()
    hook = RS.RcclRowAllreduce(rank, world, dist if world > 1 else None) if mode == "rccl" else \
        RS.HostStagedRowAllreduce(ctx, dist if world > 1 else None)
    r = RS.optimize_device(pr, hook, m, iterations=40, trace=True)
    calls, dbl = hook.stats()
    q.put((rank, r.iterations, r.lsmr_iterations, r.ssr, r.converged, r.f_calls, r.g_calls, r.mul_calls, r.minimizer, calls, dbl,
           np.array(r.trace["inner"])))
    pr.close()
    hook.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_device(world, mode, shape, device_per_rank=False):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_device, args=(r, world, port, q, mode, shape, device_per_rank)) for r in range(world)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(world):
        rec = q.get(timeout=240)
        res[rec[0]] = rec[1:]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [BIG, MID], ids=["sliced_colscaled", "segments"])
def test_row_sharded_device_loop_through_rccl_equals_oracle(shape):
    """SURVEY 8f-4 on the device: lsq_optimize with lsq_options.row_allreduce = a DIRECT ncclAllReduce on the library's stream
    (liblsqrccl.so; a one-rank communicator: a GPU box has one device).  The inner iteration's J'u + |u|^2 go through the
    collective between k_combine and the EpiV epilogue -- no host round trip, no staging -- and the run equals the unsharded
    ORACLE (not the product's own unsharded run)."""
    res = _run_device(1, "rccl", shape)
    m, n, colptr, rowval, A, S, b = _problem(*shape)
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    _check_against_oracle(ro, res[0], n)


def _visible_devices():
    """GPUs of the box, from the KFD topology (no HIP runtime is initialised at collection time)."""
    import glob
    count = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split() for l in open(f) if len(l.split()) == 2)
            count += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            pass
    return count or len(glob.glob("/dev/dri/renderD*"))      # (topology not readable: one render node per GPU)


@pytest.mark.gpu
@pytest.mark.skipif(_visible_devices() < 2, reason="needs two GPUs: RCCL cannot put two ranks on one device")
def test_row_sharded_device_loop_through_rccl_world2_equals_oracle():
    """the direct-RCCL hook with MORE than one rank (one device per rank, xGMI between them): every rank must
    issue the same number of ncclAllReduce calls (chunking by the look-ahead, the `reported` gate, the unique id carried by
    the launcher's group), walk the unsharded oracle's trajectory, and end with a bit-identical replicated minimizer.
    Skipped on the one-GPU boxes of this pool; written for the 8-GPU tier."""
    res = _run_device(2, "rccl", BIG, device_per_rank=True)
    m, n, colptr, rowval, A, S, b = _problem(*BIG)
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    for rank in (0, 1):
        _check_against_oracle(ro, res[rank], n)
    assert np.array_equal(res[0][7], res[1][7])
    assert res[0][8:10] == res[1][8:10]                 # hook.stats(): the same collectives, the same payload


@pytest.mark.gpu
def test_row_sharded_device_loop_world2_equals_oracle():
    """Two ranks, each with half of the rows on the device (they share the box's one GPU, which RCCL cannot do: the hook is
    the host-staged gloo one), against the unsharded oracle; the replicated minimizer is bit-identical on both ranks."""
    res = _run_device(2, "gloo", MID)
    m, n, colptr, rowval, A, S, b = _problem(*MID)
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    for rank in (0, 1):
        _check_against_oracle(ro, res[rank], n)
    assert np.array_equal(res[0][7], res[1][7])


@pytest.mark.gpu
def test_row_sharded_host_driver_on_device_world2():
    """The host-level driver (lm_lsmr over the operator-level C ABI, every array operation a call on device memory; exchange
    over gloo) against the unsharded ORACLE."""
    res = _run_world2("hip")
    m, n, colptr, rowval, A, S, b = _problem(120000, 400, 300, 3)
    ro = _oracle(m, n, colptr, rowval, A, b, 40)
    inner = int(ro.trace["inner"].sum()) // 2
    for rank in (0, 1):
        it, li, ssr, conv, fcalls, gcalls, mulc, x, calls, dbl = res[rank]
        assert (it, fcalls, gcalls, mulc, li, conv) == (ro.iterations, ro.f_calls, ro.g_calls, ro.mul_calls, inner, ro.converged)
        assert ssr == pytest.approx(ro.ssr, rel=1e-10) and np.max(np.abs(x - ro.minimizer)) <= 1e-8


def _worker_ldiv(rank, world, port, q, damped):
    """Operator-level row sharding (include/lsqhip.h: lsq_solver_set_row_allreduce + lsq_ldiv / lsq_ldiv_damped): this rank's
    row block of J and y, one ldiv!, then colsumabs2 of the handle (must still be the LOCAL block's) and an unsharded solve
    on the same handle (must not see the ranks' sum)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n, S, y, damp = _ldiv_problem()
    lo, hi = RS.row_slice(m, rank, world)
    Sp = S.tocsr()[lo:hi].tocsc()
    Sp.sort_indices()
    ctx = lsq.Context(0)
    J = lsq.DeviceMatrix(ctx, Sp)
    sv = lsq.AllocatedSolver(J, lsq.LSMR(), for_lm=damped)
    hook = RS.HostStagedRowAllreduce(ctx, dist)
    sv.set_row_allreduce(hook, m)
    x = lsq.DeviceVector(ctx, n)
    dy = lsq.DeviceVector(ctx, hi - lo, y[lo:hi])
    if damped:
        dd = lsq.DeviceVector(ctx, n, damp)
        _, nmul = sv.ldiv_(x, dy, dd)
    else:
        _, nmul = sv.ldiv_(x, dy)
    xs = x.get()
    calls = hook.stats()[0]
    cs_local = lsq.colsumabs2_(lsq.DeviceVector(ctx, n), J).get()
    # the same handle, unsharded afterwards: this block alone
    sv.set_row_allreduce(None, 0)
    if damped:
        _, nmul_own = sv.ldiv_(x, dy, lsq.DeviceVector(ctx, n, damp))
    else:
        _, nmul_own = sv.ldiv_(x, dy)
    q.put((rank, xs, nmul, calls, cs_local, x.get(), nmul_own, sv.info()["lsmr_iter"]))
    dist.barrier()
    dist.destroy_process_group()


def _ldiv_problem():
    rng = np.random.default_rng(5)
    m, n = 9000, 300            # (beyond the reference-order kernels' size: the fast LSMR path)
    S = sp.random(m, n, density=0.02, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    S.sort_indices()
    # rows with very different weights in the two halves: the blocks' own column sums differ a lot from their total
    S = sp.diags(np.where(np.arange(m) < m // 2, 3.0, 0.2)) @ S
    S = S.tocsc()
    S.sort_indices()
    return m, n, S, rng.standard_normal(m), rng.random(n) + 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("damped", [True, False])
def test_row_sharded_operator_level_ldiv_world2_equals_oracle(damped):
    """a row-sharded ldiv! built its default Jacobi preconditioner from the LOCAL block's colsumabs2, so
    the replicated vectors diverged between the ranks.  Two ranks (gloo, host-staged hook), one damped / undamped LSMR solve
    of a row-split J against the ORACLE on the whole J: same mul count, same x on both ranks bit for bit; the handle's own
    colsumabs2 stays the block's; the same solver un-sharded afterwards equals the oracle on the block alone."""
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    ps = [mpc.Process(target=_worker_ldiv, args=(r, 2, port, q, damped)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(2):
        rec = q.get(timeout=240)
        res[rec[0]] = rec[1:]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    m, n, S, y, damp = _ldiv_problem()
    if damped:
        st, xr, nmul_r, _ = O.ldiv(O.LSMR, O.Mat.from_scipy(S), y, damp)
    else:
        st, xr, nmul_r = O.ldiv(O.LSMR, O.Mat.from_scipy(S), y)
    assert np.array_equal(res[0][0], res[1][0])                    # replicated x: the same bits on both ranks
    for rank in (0, 1):
        xs, nmul, calls, cs_local, x_own, nmul_own, it_own = res[rank]
        assert nmul == nmul_r and nmul > 0, (nmul, nmul_r)
        assert np.allclose(xs, xr, rtol=1e-8, atol=1e-10)
        lo, hi = RS.row_slice(m, rank, 2)
        Sp = S.tocsr()[lo:hi].tocsc()
        assert np.allclose(cs_local, np.asarray(Sp.multiply(Sp).sum(axis=0)).ravel(), rtol=1e-12)
        if damped:
            _, xo, nmul_o, _ = O.ldiv(O.LSMR, O.Mat.from_scipy(Sp), y[lo:hi], damp)
        else:
            _, xo, nmul_o = O.ldiv(O.LSMR, O.Mat.from_scipy(Sp), y[lo:hi])
        assert nmul_own == nmul_o and np.allclose(x_own, xo, rtol=1e-8, atol=1e-10)
    assert res[0][2] == res[1][2]                                   # the same number of collectives on both ranks


def test_rccl_shim_exports_its_header():
    """include/lsqrccl.h vs liblsqrccl.so: every declared symbol is exported (no RCCL call without a GPU)."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "lsqrccl.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(lsq_rccl_\w+)\s*\(", txt)))
    assert len(names) == 14, names      # 7 of the row-sharded hook + 7 of the scalar exchange (lsq_rccl_xchg_*; _reset: round 6)
    L = C.CDLL(os.path.join(root, "leastsquaresoptim.jl_amd", "liblsqrccl.so"))
    for nme in names:
        assert hasattr(L, nme), nme
