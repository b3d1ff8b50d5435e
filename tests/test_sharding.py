"""N>1 path on CPU: the single per-outer-iteration exchange of sharded runs over `gloo`
(world_size 2), and -- on the GPU box -- two ranks driving the real LM loop with that exchange
(frozen-rank protocol: a rank that converged keeps taking part until all have)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lsq_amd as lsq


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_exchange(rank, world, port, q):
    _init(rank, world, port)
    buf = torch.zeros(world + 2, dtype=torch.float64)
    out = []
    # rank r contributes ssr = r+1, gnorm = 10*(r+1); only rank 0 has converged in round 0
    out.append(lsq.sharding.exchange(dist, rank, world, rank + 1.0, 10.0 * (rank + 1), rank == 0, buf))
    out.append(lsq.sharding.exchange(dist, rank, world, 0.5, 0.25 if rank else 7.0, True, buf))
    cb = lsq.sharding.make_allreduce_callback(dist, rank, world, "cpu")
    import ctypes as C
    vals = (C.c_double * 3)(2.0 * (rank + 1), 3.0 - rank, 1.0)
    assert cb(vals, 3, None) == 0
    out.append((vals[0], vals[1], vals[2]))
    # active ranks (converged = 0): the first call is synchronous, later ones return the PREVIOUS exchange
    # while their own all-reduce stays in flight; a converged call drains it and is synchronous again
    cb2 = lsq.sharding.make_allreduce_callback(dist, rank, world, "cpu")
    seq = []
    for it, conv in enumerate([0.0, 0.0, 0.0, 1.0]):
        vals = (C.c_double * 3)(10.0 * it + rank, float(it + rank), conv)
        assert cb2(vals, 3, None) == 0
        seq.append((vals[0], vals[1], vals[2]))
    out.append(seq)
    q.put((rank, out))
    dist.destroy_process_group()


def test_exchange_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][0] == (3.0, 20.0, False)          # sum, max, not all converged
        assert res[r][1] == (1.0, 7.0, True)
        assert res[r][2] == (6.0, 3.0, 1.0)
        # sums over ranks {0,1}: iteration it contributes 20*it + 1, max gnorm it + 1
        assert res[r][3] == [(1.0, 1.0, 0.0), (1.0, 1.0, 0.0), (21.0, 2.0, 0.0), (61.0, 4.0, 1.0)]


def _worker_lm(rank, world, port, q):
    _init(rank, world, port)
    ctx = lsq.Context(0)
    # two different problems; the second needs more iterations
    m, n = (4000, 40) if rank == 0 else (6000, 60)
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=100, seed=11 + rank, ctx=ctx)
    pr.reset()
    cb = lsq.sharding.make_allreduce_callback(dist, rank, world, "cpu")
    r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=60, allreduce=cb)
    pr.reset()
    r1 = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, iterations=60)  # same problem, alone
    q.put((rank, r.iterations, r.converged, r.ssr, r1.iterations, r1.ssr, float(np.max(np.abs(r.minimizer - r1.minimizer)))))
    pr.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_lm_with_exchange():
    """Both ranks leave the loop in the same outer iteration (the slower problem's count), each
    rank's own solution is unchanged by the exchange, and the reported ssr is the global sum."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_lm, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(world):
        rec = q.get(timeout=300)
        res[rec[0]] = rec[1:]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (it0, c0, ssr0, alone0, s0, d0), (it1, c1, ssr1, alone1, s1, d1) = res[0], res[1]
    assert c0 and c1
    assert it0 == it1 == max(alone0, alone1) + 1 or it0 == it1 == max(alone0, alone1)
    assert d0 == 0.0 and d1 == 0.0                      # local trajectories untouched
    assert ssr0 == pytest.approx(s0 + s1, rel=1e-12) and ssr1 == pytest.approx(ssr0, rel=1e-15)
