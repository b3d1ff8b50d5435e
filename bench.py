#!/usr/bin/env python3
"""bench.py -- LM+LSMR outer iterations / second on the C4 workload of BASELINE.json
(sparse CSC 10^6 x 10^4, 0.1 % nnz => nnz = 10^7, LevenbergMarquardt(LSMR())), one problem per GPU.

A "step" is ONE Levenberg-Marquardt outer iteration of the hot path (levenberg_marquardt.jl:72-140)
on the synthetic tanh model: g! (when the previous step was accepted), colsumabs2, the damped
preconditioned LSMR solve (k inner iterations of J*v / J'*u), J'f, f!, J*dx and the accept/reject
logic -- everything resident in HBM, f!/g! on the device.  The timed region is EXACTLY K steps of
the REFERENCE'S OWN SCHEDULE (round 6): solves from x0 = 0 with the reference's default tolerances
(x_tol = f_tol = g_tol = 1e-8, levenberg_marquardt.jl:41), each of which stops by itself at
convergence (6 iterations on this problem, levenberg_marquardt.jl:123-124), repeated until K steps
are done (the last solve is cut at K).  Every timed step is one the reference would run, and one at
which the HIP path and the oracle take the same decisions (`parity_ok`).  The zero-tolerance
8-iteration schedule of rounds 1-5 (two cheap steps past convergence per solve) is reported beside
it as `value_fixed8_schedule`, never as `value`.

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: spawns N ranks under torch.distributed.run itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, the J*v product inside LSMR
(k_sell_rows<EpiU>): algorithmic bytes per launch (12*nnz + 4*(m+1) + 8*n + 16*m
= 140.08 MB, plus 24*n for the fused damping rows) / average launch duration measured with HIP
events on the library's stream inside the timed region.  `cpu_baseline` is the oracle (scalar C
port, 1 thread) timed on this box's host on a bounded sample of the same workload, with an OpenMP
all-cores variant of the same path beside it (`cpu_baseline.cpu_all_cores_value`).  The driver's
record keeps scalars of `config`, `roofline` and `cpu_baseline` only, so every companion figure
(generic g!, fixed-8 schedule, C2 / C3 / wide-n legs, J'u, parity, tail speculation) is ALSO a
scalar inside one of those three objects (and at top level).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--iters-per-solve", type=int, default=8)
    ap.add_argument("--m", type=int, default=1_000_000)
    ap.add_argument("--n", type=int, default=10_000)
    ap.add_argument("--per-col", type=int, default=1000)
    ap.add_argument("--cpu-steps", type=int, default=96, help="outer iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the secondary dense-path timings (C2/C3 ldiv!)")
    ap.add_argument("--repeats", type=int, default=25,
                    help="how many times the K-step timed region is repeated (value = median region)")
    ap.add_argument("--exchange", choices=("rccl", "nccl", "gloo"), default=os.environ.get("LSQ_EXCHANGE_BACKEND", "rccl"),
                    help="the per-outer-iteration ||r|| exchange of sharded runs: rccl = served in C (liblsqrccl.so: a direct "
                         "ncclAllReduce over xGMI on a side stream, torch only carries the unique id); nccl / gloo = the "
                         "Python hook over that torch.distributed backend (A/B)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch-path check without a GPU: spawn the ranks, build the process group (gloo), run the "
                         "exchange protocol on made-up scalars and print the JSON line with value = null")
    # ranks started by spawn_ranks() get the original command line through the environment: torchrun's own argparse
    # would try to abbreviation-match script options such as --m against its own (--master-addr, --max-restarts, ...)
    argv = json.loads(os.environ["LSQ_BENCH_ARGV"]) if "LSQ_BENCH_ARGV" in os.environ else None
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a torchrun environment: re-execute under torch.distributed.run with one
    rank per GPU (what the driver's own multi-GPU command line does).  Does not return."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    os.environ["LSQ_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    # What RCCL / device-memory sharing between the ranks need on this driver must be in the environment BEFORE torch (and
    # with it the HIP runtime) is loaded -- also when a launcher (torch.distributed.run from the driver) started this rank
    # and spawn_ranks() below never ran: the host driver only supports dmabuf IPC.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                      # re-executes under torch.distributed.run; never returns
    # Exactly ONE line on stdout: native libraries (RCCL's version banner, rocm notices) also write to fd 1,
    # so everything but the final JSON line is sent to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if a.dry_run:
        return dry_run(a, rank, world, real_stdout)
    import torch  # plumbing only: device selection, barrier, RCCL all-reduce
    dist = None
    # LSQ_BENCH_FORCE_EXCHANGE=1 (diagnostic): run the sharded protocol with its RCCL exchange even on one rank
    force_x = world == 1 and os.environ.get("LSQ_BENCH_FORCE_EXCHANGE") == "1"
    if force_x:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    exchange_backend, rccl_ranks = None, 0
    if world > 1 or force_x:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()          # size of the nccl (= RCCL) group that carries the exchange
        # The per-outer-iteration exchange {sum ssr, max |g|, all-converged} is ONE RCCL all-reduce over xGMI
        # (north_star).  --exchange gloo / LSQ_EXCHANGE_BACKEND=gloo is an opt-in A/B (the payload is 80
        # bytes of host scalars, and an RCCL kernel has to find a CU next to the persistent product kernels).
        xgroup, xdev, exchange_backend = None, "cuda", "rccl-c" if a.exchange == "rccl" else "nccl"
        if a.exchange == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: the container hostname may not resolve
            xgroup, xdev, exchange_backend = dist.new_group(backend="gloo"), "cpu", "gloo"
        print("bench: rank %d/%d on cuda:%d, exchange backend %s" % (rank, world, local_rank, exchange_backend),
              file=sys.stderr)
    else:
        torch.cuda.set_device(local_rank)
    import numpy as np
    import lsq_amd as lsq
    L = lsq.lib()
    ctx = lsq.Context(local_rank)

    def probe(stage):      # diagnostics (LSQ_BENCH_PROBE=1): the dense Dogleg+QR outer iteration measured at this point of the run
        if not os.environ.get("LSQ_BENCH_PROBE"):
            return
        p3 = lsq.synthetic.TanhProblem(16384, 2048, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
        p3.reset()
        p3.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)
        p3.reset()
        t_ = time.perf_counter()
        r_ = p3.optimize(lsq._lib.DOGLEG, lsq._lib.QR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False)
        ctx.sync()
        print("probe[%s]: %.2f ms per Dogleg+QR outer iteration" % (stage, (time.perf_counter() - t_) / r_.iterations * 1e3), file=sys.stderr)
        p3.close()
    probe("context created")
    m, n, pc = a.m, a.n, a.per_col
    nnz = n * pc
    seed = lsq.synthetic.BASE_SEED + rank  # the 8 problems of C5 differ
    t0 = time.time()
    inputs = lsq.synthetic.sparse_inputs(m, n, pc, seed)
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=seed, ctx=ctx, inputs=inputs)
    t_setup = time.time() - t0

    # one scalar all-reduce per outer iteration for the sharded (C5) case (sharding.py): sum of ssr,
    # max of the gradient norms, all-converged -- ONE RCCL all-reduce of world+2 doubles.
    allreduce = xchg_c = None
    if world > 1 or force_x:
        from lsq_amd import sharding
        if exchange_backend == "rccl-c":
            # the exchange in C: its own RCCL communicator (unique id broadcast over the process group), one ncclAllReduce of
            # world + 3 doubles per outer iteration on a side stream, no Python on the thread that feeds the launches
            try:
                allreduce = xchg_c = sharding.RcclScalarExchange(rank, world, dist if world > 1 else None)
            except Exception as e:   # noqa: BLE001  (plumbing must not cost the run: the Python hook over the same RCCL)
                print("bench: rank %d: C exchange unavailable (%s); falling back to the Python hook over nccl" % (rank, e),
                      file=sys.stderr)
                exchange_backend = "nccl (fallback from rccl-c)"
            # every rank must take the same path: a collective mismatch would hang the job
            if world > 1:
                okt = torch.tensor([1.0 if xchg_c is not None else 0.0], dtype=torch.float64, device="cuda")
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if okt.item() < 0.5 and xchg_c is not None:
                    xchg_c.close()
                    allreduce = xchg_c = None
                    exchange_backend = "nccl (fallback from rccl-c)"
        if allreduce is None:
            allreduce = sharding.make_allreduce_callback(dist, rank, world, xdev, group=xgroup)

    LM, LSMR = lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR

    SOLVE_CAP = 50      # (a solve of the reference schedule that did not converge would end here; C4 converges in 6)

    def run(iters, fixed=0):
        """`iters` LM outer iterations.  fixed == 0: the reference's schedule -- solves from x0 = 0 with the DEFAULT tolerances, each
        stopping by itself at convergence, the last one cut at `iters`.  fixed == k > 0: the schedule of rounds 1-5 (solves of k
        iterations, zero tolerances).  Returns (iterations counted by the loops, iterations that did device work on this rank --
        a rank of a sharded run that has converged takes part in the exchange without working --, inner iterations, last result)."""
        done = real = inner = 0
        r = None
        while done < iters:
            if fixed:
                k = min(fixed, iters - done)
                pr.reset()
                r = pr.optimize(LM, LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=k, allreduce=allreduce, fetch_x=False)
                assert r.iterations == k, (r.iterations, k)
            else:
                k = min(SOLVE_CAP, iters - done)
                pr.reset()
                r = pr.optimize(LM, LSMR, iterations=k, allreduce=allreduce, fetch_x=False)     # x_tol = f_tol = g_tol = 1e-8
                assert 0 < r.iterations <= k, (r.iterations, k)
            done += r.iterations
            real += r.f_calls - 1                 # (one f! per working iteration + the one at x0)
            inner += r.lsmr_iterations
        return done, real, inner, r

    def barrier():
        if dist is not None:
            from lsq_amd import sharding as _sh
            _sh.drain_all()   # exchanges the active ranks left in flight
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    probe("problem created")
    if a.warmup > 0:
        run(a.warmup)
    # The timed region is EXACTLY K steps between barrier + synchronize; it is repeated --repeats times (each
    # repeat bracketed the same way) and `value` comes from the MEDIAN region, so the headline is not one sample
    # of a few milliseconds.  Every twelfth J*v launch of every region carries its own start/stop events (250 samples
    # per run; a timed launch costs ~9 us of pipeline gaps, so every third -- rounds 1-3 -- cost 1.5 % of the rate it measured).
    probe("warm-up done")
    stride = int(os.environ.get("LSQ_BENCH_PROF_STRIDE", "12"))
    L.lsq_prof_select(ctx.h, 1 | (stride << 8))     # bits 0-7: kernel mask; bits 8+: time every k-th launch
    L.lsq_prof_begin(ctx.h, 1 << 16)
    region_s, inner_local, real_local, r = [], 0, 0, None
    tail_before = ctx.tail_stats()
    reps = max(1, a.repeats)
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        steps_done, real_rep, inner_rep, r = run(a.steps)
        barrier()
        region_s.append(time.perf_counter() - t0)
        inner_local += inner_rep
        real_local += real_rep
        assert steps_done == a.steps, (steps_done, a.steps)
        assert world > 1 or real_rep == a.steps, (real_rep, a.steps)
    tail_timed = tuple(int(x - y) for x, y in zip(ctx.tail_stats(), tail_before))
    iters_per_ref_solve = None
    if r is not None:
        pr.reset()
        iters_per_ref_solve = pr.optimize(LM, LSMR, iterations=SOLVE_CAP, allreduce=allreduce, fetch_x=False).iterations
    probe("timed regions done")
    avg = (C.c_double * 2)()
    cnt = (C.c_int * 2)()
    L.lsq_prof_end(ctx.h, avg, cnt)
    probe("lsq_prof_end")
    # the J'u kernel is timed in a separate (untimed) pass of one solve: every instrumented launch
    # costs a few microseconds of pipeline gaps, which the timed region should not pay twice
    L.lsq_prof_select(ctx.h, 2)
    L.lsq_prof_begin(ctx.h, 8192)
    run(a.iters_per_solve, fixed=a.iters_per_solve)
    avg2 = (C.c_double * 2)()
    cnt2 = (C.c_int * 2)()
    L.lsq_prof_end(ctx.h, avg2, cnt2)
    avg[1], cnt[1] = avg2[1], cnt2[1]
    L.lsq_prof_select(ctx.h, 3)
    ev_ovh = C.c_double(0.0)   # what an EMPTY event pair measures on this stream (marker overhead)
    L.lsq_prof_overhead(ctx.h, 50, C.byref(ev_ovh))
    probe("J'u pass + overhead")
    region_s_local = list(region_s)
    own_srt = sorted(region_s_local)
    own_rate = a.steps / own_srt[len(own_srt) // 2]
    rank_rates = [own_rate]
    if dist is not None:
        tt = torch.tensor(region_s, dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)       # every region: the slowest rank's time
        region_s = [float(v) for v in tt.tolist()]
        rr = torch.zeros(world, dtype=torch.float64, device="cuda")
        rr[rank] = own_rate
        dist.all_reduce(rr)                             # every rank's own median rate, for the line (a straggler shows)
        rank_rates = [float(v) for v in rr.tolist()]
        it = torch.tensor([float(inner_local), float(real_local)], dtype=torch.float64, device="cuda")
        dist.all_reduce(it)
        inner_total, real_total = float(it[0].item()), float(it[1].item())
    else:
        inner_total, real_total = float(inner_local), float(real_local)
    # every rank's own rate on stderr: a straggler among N ranks is visible next to the max-over-ranks headline
    own = sorted(region_s_local)
    print("bench: rank %d/%d own median region %.3f ms = %.1f LM it/s (inner %d)"
          % (rank, world, own[len(own) // 2] * 1e3, a.steps / own[len(own) // 2], inner_local), file=sys.stderr)
    srt = sorted(region_s)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    inner_total /= reps                                  # per region
    real_total /= reps                                   # working LM iterations of all ranks per region (N = 1: exactly K)

    # generic J*v (y <- J x + y) timed back-to-back with HIP events, for reference
    xv = lsq.DeviceVector(ctx, n, np.random.default_rng(0).standard_normal(n))
    yv = lsq.DeviceVector(ctx, m, np.zeros(m))
    ms = C.c_float(0)
    lsq._lib.check(L.lsq_bench_mul(pr.J, 0, 50, xv.ptr, yv.ptr, 1.0, C.byref(ms)))
    ms_t = C.c_float(0)
    lsq._lib.check(L.lsq_bench_mul(pr.J, 1, 50, yv.ptr, xv.ptr, 1.0, C.byref(ms_t)))

    probe("lsq_bench_mul")

    def leave_group():
        if dist is not None:
            from lsq_amd import sharding as _sh
            _sh.drain_all()
            if xchg_c is not None:
                xchg_c.close()          # (the exchange's own communicator goes before the process group does)
            dist.barrier()
            dist.destroy_process_group()

    if rank != 0:
        leave_group()
        return

    bytes_jv = 12 * nnz + 4 * (m + 1) + 8 * n + 16 * m      # SURVEY 8d (CSR mirror, beta != 0)
    bytes_k1 = bytes_jv + 24 * n                            # + fused damping rows (t, dg, ux rw)
    bytes_jtu = 12 * nnz + 4 * (n + 1) + 8 * m + 16 * n
    # the two events are the launch's own start/stop events (hipExtLaunchKernelGGL): they carry the
    # dispatch's begin/end timestamps, i.e. the duration rocprofv3 reports -- no marker overhead
    k1_raw_ms = avg[0] if cnt[0] > 0 else float("nan")
    k1_ms = k1_raw_ms
    achieved = bytes_k1 / (k1_ms * 1e-3) / 1e9 if cnt[0] > 0 else None
    three = not os.environ.get("LSQ_LSMR_FOUR_LAUNCHES") and n <= 12160
    roof = {"bound": "hbm", "kernel": ("k_lsmr_fused (LSMR J*v: u <- (J w)/alpha - cu*u, + sum u^2; the n-vector updates and the stop test of "
                                        "the previous iteration ride in 3 extra workgroups)" if three else
                                        "k_sell_rows<EpiU> (LSMR J*v: u <- J t - cu*u, + sum u^2)"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
            "algorithmic_bytes_per_launch": bytes_k1, "avg_launch_ms": k1_ms, "launches_timed": int(cnt[0]),
            "timing": "HIP start/stop events of the launch itself (hipExtLaunchKernelGGL) on the library stream",
            "empty_event_pair_ms": ev_ovh.value,
            "jtu_kernel_avg_ms": avg[1],
            "jtu_GBps": (bytes_jtu + 16 * n) / (avg[1] * 1e-3) / 1e9 if cnt[1] else None,
            "generic_jv_ms": ms.value, "generic_jv_GBps": bytes_jv / (ms.value * 1e-3) / 1e9,
            "generic_jtu_ms": ms_t.value, "generic_jtu_GBps": bytes_jtu / (ms_t.value * 1e-3) / 1e9}

    # HBM traffic of that kernel comes from rocprofv3 PMC passes (it cannot be read inside this
    # process): the committed measurement for exactly this workload, newest round first
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")))
            if tr["config"] == {"m": m, "n": n, "nnz": nnz}:
                roof["traffic"] = tr["hbm_bytes_per_launch"]
                roof["traffic_source"] = tr["source"]
                break
        except Exception:
            pass

    # (the headline measurement is complete at this point: a failure in the reported-alongside legs must not lose it)
    # Order: the device-only secondary legs FIRST.  The OpenMP legs after them (oracle/lsq_oracle_omp.c under
    # OMP_PROC_BIND=close, then tools/hostg's producer threads) leave bound threads behind in this process, and the
    # host-synchronous Dogleg loop of dense_secondary measured 20.3 instead of 8.0 ms per outer iteration when it ran after both
    # (either one alone: 8.0; putting this thread's affinity mask back did not cure it).
    skip = set(filter(None, os.environ.get("LSQ_BENCH_SKIP", "").split(",")))     # diagnostics: legs to leave out (generic, dense, wide)
    dense = None
    if not a.no_cpu and world == 1 and not a.no_dense and "dense" not in skip:
        try:
            dense = dense_secondary(ctx, lsq, probe)
        except Exception as e:   # noqa: BLE001
            dense = {"error": repr(e)}

    wide = None
    if not a.no_cpu and world == 1 and "wide" not in skip:
        try:
            wide = sparse_secondary(ctx, lsq)
        except Exception as e:   # noqa: BLE001
            wide = {"error": repr(e)}
    cpu = parity = None
    if not a.no_cpu and a.cpu_steps > 0 and world == 1:   # reported at N = 1 only
        try:
            # the solve the timed region repeats, once more with its trace kept: the CPU leg's first solve is the same
            # problem on the same schedule, so the two trajectories are compared instead of thrown away; and the
            # zero-tolerance 8-iteration solve of rounds 1-5 (two iterations past convergence) beside it
            pr.reset()
            rgd = pr.optimize(LM, LSMR, iterations=SOLVE_CAP, trace=True)
            pr.reset()
            rgt = pr.optimize(LM, LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=a.iters_per_solve, trace=True)
            cpu, parity = cpu_baseline(a, pr, inputs, rgd, rgt)
        except Exception as e:   # noqa: BLE001
            cpu = {"value": None, "unit": "LM outer iterations/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    # the schedule of rounds 1-5 beside the headline: solves of --iters-per-solve (8) iterations with zero tolerances, i.e. two
    # cheap one-inner-iteration steps past convergence per solve (they flatter the rate: never `value`)
    fixed_sched = None
    if world == 1:
        try:
            run(a.iters_per_solve, fixed=a.iters_per_solve)
            ts = []
            for _ in range(5):
                ctx.sync()
                t0 = time.perf_counter()
                _, _, inn, _ = run(a.steps, fixed=a.iters_per_solve)
                ctx.sync()
                ts.append(time.perf_counter() - t0)
            dtm = sorted(ts)[len(ts) // 2]
            fixed_sched = {"value": a.steps / dtm, "unit": "LM outer iterations/s", "ms_per_step": dtm / a.steps * 1e3,
                           "iterations_per_solve": a.iters_per_solve, "lsmr_inner_per_outer": inn / a.steps,
                           "tolerances": "x_tol = f_tol = g_tol = 0 (rounds 1-5: runs two iterations past convergence)"}
        except Exception as e:   # noqa: BLE001
            fixed_sched = {"error": repr(e)}
    generic = None
    if not a.no_cpu and world == 1 and "generic" not in skip:
        try:
            generic = generic_g(a, ctx, lsq, inputs, pr.b)
        except Exception as e:   # noqa: BLE001
            generic = {"error": repr(e)}

    def g(d, *path):        # nested lookup that tolerates failed legs
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    value = real_total / dt         # N = 1: exactly K / dt; N > 1: the working iterations of all ranks / the slowest rank's time
    # ---- companions of the headline as SCALARS inside the three objects the driver's record keeps (and at top level) ----
    parity_ok = None if parity is None else bool(parity["ok"])
    flat = {"parity_ok": parity_ok,
            "schedule_iterations_per_solve": iters_per_ref_solve,
            "value_fixed8_schedule": g(fixed_sched, "value"),
            "value_generic_g_device": g(generic, "device_g_all_nnz", "value"),
            "value_generic_g_host_pinned_async": g(generic, "host_g_pinned_async", "value"),
            "c2_ldiv_ms": g(dense, "c2_cholesky_damped_4096x512", "ldiv_ms"),
            "c2_frac": g(dense, "c2_cholesky_damped_4096x512", "roofline", "frac"),
            "c2_lm_cholesky_outer_ms": g(dense, "c2_lm_cholesky_4096x512", "outer_iteration_ms"),
            "c3_ldiv_ms": g(dense, "c3_qr_16384x2048", "ldiv_ms"),
            "c3_frac": g(dense, "c3_qr_16384x2048", "roofline", "frac"),
            "c3_lm_stacked_ldiv_ms": g(dense, "c3_qr_lm_stacked_18432x2048", "ldiv_ms"),
            "c3_dogleg_qr_outer_ms": g(dense, "c3_dogleg_qr_16384x2048", "outer_iteration_ms"),
            "wide_jv_frac": g(wide, "jv", "roofline", "frac"), "wide_jtu_frac": g(wide, "jtu", "roofline", "frac"),
            "wide_lm_lsmr_outer_ms": g(wide, "lm_lsmr_outer_iteration_ms"),
            "tail_guesses": tail_timed[0], "tail_wrong": tail_timed[1],
            "tail_wrong_frac": (tail_timed[1] / tail_timed[0]) if tail_timed[0] else None}
    roof["jtu_frac"] = (roof["jtu_GBps"] / HBM_PEAK_GBS) if roof.get("jtu_GBps") else None
    # the PHYSICAL rate of the dominant kernel beside the algorithmic one: PMC bytes per launch / the launch's duration
    roof["physical_GBps"] = (roof["traffic"] / (k1_ms * 1e-3) / 1e9) if (roof.get("traffic") and cnt[0] > 0) else None
    roof["physical_frac"] = (roof["physical_GBps"] / HBM_PEAK_GBS) if roof["physical_GBps"] else None
    for k in ("c2_ldiv_ms", "c2_frac", "c3_ldiv_ms", "c3_frac", "wide_jv_frac", "wide_jtu_frac"):
        roof[k] = flat[k]
    config = {"workload": "C4: sparse CSC %dx%d, nnz=%d (%.3g%%), LevenbergMarquardt(LSMR()), tanh model, "
                          "1 problem per GPU" % (m, n, nnz, 100.0 * nnz / (m * n)),
              "schedule": "reference: solves from x0 = 0 with the default tolerances (1e-8), each stopping at convergence "
                          "(levenberg_marquardt.jl:41,123-124); K steps = consecutive solves, the last one cut at K",
              "m": m, "n": n, "nnz": nnz, "seed": lsq.synthetic.BASE_SEED, "problems": world,
              "exchange_backend": exchange_backend, "rccl_ranks": rccl_ranks,
              "exchange": (dict(xchg_c.stats(), served_by="liblsqrccl.so lsq_rccl_xchg_* (C, direct ncclAllReduce)")
                           if xchg_c is not None else None),
              "per_rank_it_per_s": rank_rates,
              "per_rank_it_per_s_min": min(rank_rates), "per_rank_it_per_s_max": max(rank_rates),
              "working_iterations_all_ranks_per_region": real_total,
              "lsmr_inner_iterations_total": inner_total,
              "lsmr_inner_per_outer": inner_total / max(real_total, 1e-9),
              "lsmr_inner_iterations_per_sec": inner_total / dt, "iters_per_solve_fixed8_leg": a.iters_per_solve,
              "jacobian": "column-scaled handle J = A diag(1 - tanh(x)^2) on the sliced layouts: g! writes n factors, "
                          "no Jacobian copy is multiplied out (lsq_mat_set_colscale; LSQ_NO_COLSCALE=1 restores the "
                          "multiplied-out copies of rounds 1-2)" if not os.environ.get("LSQ_NO_COLSCALE") else
                          "multiplied out into both sliced copies by g!",
              "lsmr_iteration": lsmr_iteration_text(three),
              "lm_tail": ("predicted residual and trial residual in ONE pass over A (k_sell_rows_pair, n <= 10200; LSQ_NO_PAIR_TAIL=1 "
                          "restores the two launches)" if n <= 10200 and not os.environ.get("LSQ_NO_PAIR_TAIL")
                          and not os.environ.get("LSQ_NO_COLSCALE") else "two passes over A (predicted residual, trial residual)"),
              "final_ssr": r.ssr, "setup_seconds": t_setup,
              # what the launch heuristics saw: 256 CUs / 8 XCDs = an unpartitioned MI355X (SPX); a partitioned device
              # (CPX: 32 CUs) takes other kernels in the dense solvers (no slab exchange) and fewer workgroups everywhere
              "device": ctx.device_info(), "debug_modes": dict(zip(("launch_jitter_us", "serial", "stalls"), lsq.debug_get()))}
    if xchg_c is not None:      # the exchange's counters as scalars too (a future SCALE line is checkable from the record)
        for k, v in xchg_c.stats().items():
            if isinstance(v, (int, float)):
                config["exchange_" + k] = v
    config.update(flat)
    if cpu is not None and cpu.get("value") is not None:
        ac = cpu.get("all_cores") or {}
        cpu["cpu_all_cores_value"] = ac.get("value")
        cpu["cpu_all_cores_threads"] = ac.get("cores")
        cpu["gpu_over_cpu_1_thread"] = value / cpu["value"] if cpu["value"] else None
        cpu["gpu_generic_g_over_cpu_1_thread"] = (flat["value_generic_g_device"] / cpu["value"]
                                                   if flat["value_generic_g_device"] and cpu["value"] else None)
        cpu["gpu_over_cpu_all_cores"] = value / ac["value"] if ac.get("value") else None
        cpu["parity_ok"] = parity_ok
    out = {"metric": "lm_lsmr_outer_iterations_per_sec", "value": value, "unit": "LM outer iterations/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "repeats": reps, "region_ms": {"median": dt * 1e3, "min": srt[0] * 1e3, "max": srt[-1] * 1e3},
           "value_min": real_total / srt[-1], "value_max": real_total / srt[0],
           "config": config,
           "roofline": roof, "cpu_baseline": cpu, "parity_vs_cpu": parity, "fixed8_schedule": fixed_sched, "generic_g": generic,
           "dense_secondary": dense, "sparse_secondary": wide,
           # bounded-wait give-ups of the fast paths that assume co-resident workgroups (include/lsqhip.h: lsq_solver_stats)
           "fallback_giveups": ctx.fallback_stats(),
           # LM+LSMR: solves whose follow-up kernels were queued behind a guessed last inner iteration, and wrong guesses --
           # inside the timed regions (`timed_regions`) and over the whole process (`whole_run`)
           "tail_speculation": {"timed_regions": dict(zip(("guesses", "wrong"), tail_timed)),
                                "whole_run": dict(zip(("guesses", "wrong"), ctx.tail_stats()))}}
    out.update(flat)
    leave_group()
    if parity is not None and not parity["ok"]:
        # a rate measured on a trajectory that is not the reference's is not a measurement of this path: the line fails
        out["invalid"] = "parity_vs_cpu failed: the HIP run of the timed solve does not follow the oracle's trajectory"
        out["value_unchecked"], out["value"] = out["value"], None
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())   # fd 1 stays on stderr: RCCL prints its banner at exit
    if out.get("invalid"):
        raise SystemExit(3)


def lsmr_iteration_text(three):
    if not three:
        return "four launches: k_sell_rows<EpiU> | k_sell_cols | k_combine<EpiV> | k_lsmr_update"
    return ("three launches: k_lsmr_fused | k_sell_cols | k_combine<EpiV> (LSQ_LSMR_FOUR_LAUNCHES=1 restores "
            "k_sell_rows<EpiU> | k_sell_cols | k_combine<EpiV> | k_lsmr_update)")


def dry_run(a, rank, world, real_stdout):
    """--dry-run: everything about the launch EXCEPT the hot path (there is no CPU implementation of it): rank
    bookkeeping, process group, the sharded runs' exchange protocol (sharding.py) on made-up scalars over gloo,
    max-over-ranks of a timed region, ONE JSON line from rank 0 with value = null.  What the CPU test suite uses to
    check that `bench.py --gpus N` really runs N ranks."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from lsq_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
        # the protocol the GPU runs use (liblsqrccl.so: lsq_rccl_xchg_*), here over a gloo transport
        cb = sharding.TorchTransportExchange(dist, rank, world)
        t0 = time.perf_counter()
        seen = []
        for it in range(a.steps):
            vals = (C.c_double * 3)(1.0 + rank, float(it), 1.0 if it == a.steps - 1 else 0.0)
            assert cb(vals, 3, None) == 0
            seen.append((vals[0], vals[1], vals[2]))
        sharding.drain_all()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ranks = dist.get_world_size()
        assert seen[-1] == (world * (world + 1) / 2.0, float(a.steps - 1), 1.0), seen[-1]
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = 1
    if rank == 0:
        out = {"metric": "lm_lsmr_outer_iterations_per_sec", "value": None, "unit": "LM outer iterations/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "dry_run": True,
               "config": {"workload": "dry run: launch path only, no hot-path work", "problems": world,
                          "exchange_backend": "c-protocol (liblsqrccl.so) over gloo", "rccl_ranks": 0,
                          "process_group_ranks": ranks}}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: dense fp64 matrix peak


def newest_profile(name):
    """profiles/rNN/<name> of the newest round that holds it (the committed rocprofv3 summaries the line's figures refer to)."""
    base = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(base) if d.startswith("r")), reverse=True):
        if os.path.exists(os.path.join(base, rnd, name)):
            return "profiles/%s/%s" % (rnd, name)
    return "profiles/ (none committed)"


def sparse_secondary(ctx, lsq):
    """The sparse products where x no longer fits in LDS (n > 12160, DESIGN 4.1: column-windowed sliced rows): the C4 entry
    count spread over n = 25000 columns -- J*v and J'u back to back (HIP events around 30 launches, lsq_bench_mul) against the
    HBM roofline with the algorithmic bytes of SURVEY 8d, and the LM+LSMR outer iteration on the tanh model.  Not part of
    `value`."""
    import ctypes as C
    import time
    L = lsq.lib()
    m, n, pc = 1_000_000, 25_000, 400
    pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
    nnz = pr.nnz
    pr.reset()
    pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)   # (J = A diag(s))
    best = None
    for _ in range(3):
        pr.reset()
        t0 = time.perf_counter()
        r = pr.optimize(lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=6, fetch_x=False)
        ctx.sync()
        ms = (time.perf_counter() - t0) / max(r.iterations, 1) * 1e3
        best = ms if best is None else min(best, ms)
    import numpy as np
    x = lsq.DeviceVector(ctx, n, np.ones(n))
    y = lsq.DeviceVector(ctx, m, np.ones(m))
    out = {"workload": "sparse CSC %dx%d, nnz=%d, tanh model" % (m, n, nnz),
           "lm_lsmr_outer_iteration_ms": best, "lsmr_inner_per_outer": r.lsmr_iterations / max(r.iterations, 1)}
    for trans, name in ((0, "jv"), (1, "jtu")):
        ms = C.c_float(0)
        if trans == 0:
            lsq._lib.check(L.lsq_bench_mul(pr.J, 0, 30, x.ptr, y.ptr, 1.0, C.byref(ms)))
        else:
            lsq._lib.check(L.lsq_bench_mul(pr.J, 1, 30, y.ptr, x.ptr, 1.0, C.byref(ms)))
        b = 12 * nnz + (4 * (m + 1) + 8 * n + 16 * m if trans == 0 else 4 * (n + 1) + 8 * m + 16 * n)
        out[name] = {"ms": ms.value, "roofline": {"bound": "hbm", "achieved": b / (ms.value * 1e-3) / 1e9, "peak": 8000.0,
                                                   "unit": "GB/s", "frac": b / (ms.value * 1e-3) / 1e9 / 8000.0,
                                                   "algorithmic_bytes_per_launch": b}}
    pr.close()
    return out


def generic_g(a, ctx, lsq, inputs, b):
    """The SAME C4 problem and schedule with a g! that does what the reference's general sparse g! does -- rewrite every stored
    value of J (test/nonlinearleastsquares.jl:47-86) -- instead of the headline's column-scaled handle, which only a model
    of the form r = V phi(x) - b can use.  Two legs, each as LM outer iterations / s and ms per step (median of 5 regions of
    --steps steps), reported NEXT TO the headline, never as `value`:
      device_g_all_nnz        g! on the device multiplies A diag(1 - tanh(x)^2) out into both sliced copies (the layouts
                              J*v and J'u read): 2 x (read 8 B + index 2 B, write 8 B) per stored entry per accepted step;
                              predicted and trial residual as two passes (no shared stream of A);
      host_g_pinned_async     g! on the HOST, in C + OpenMP (tools/hostg/hostg.c, a consumer of include/lsqhip.h): x comes
                              down (80 KB), nnz values are written into a page-locked buffer and go up through
                              lsq_mat_set_values_async (80 MB over PCIe per accepted step), the device re-sorts them into
                              its two layouts; f! stays on the device.  What a Julia g! over the shim would cost at best."""
    import ctypes as C
    import numpy as np
    L = lsq.lib()
    m, n, pc = a.m, a.n, a.per_col
    LM, LSMR = lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR
    colptr, rowval, nzval = inputs
    out = {"workload": "same C4 problem and schedule as the headline (solves from x0 = 0 with the reference's default tolerances, "
                       "each stopping at convergence)",
           "headline_g": "column-scaled handle: g! writes n factors (model-specific)"}
    saved = {k: os.environ.get(k) for k in ("LSQ_NO_COLSCALE", "LSQ_NO_PAIR_TAIL")}
    os.environ["LSQ_NO_COLSCALE"] = "1"     # (read when the model is created: it keeps A aside and J gets multiplied out)
    os.environ["LSQ_NO_PAIR_TAIL"] = "1"
    try:
        pr2 = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, seed=0, ctx=ctx, inputs=inputs, b=b)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def timed(optimize, regions=5):
        def run(iters):
            done = 0
            r = None
            while done < iters:
                k = min(50, iters - done)
                pr2.reset()
                r = optimize(k)
                assert 0 < r.iterations <= k, (r.iterations, k)
                done += r.iterations
            return r
        run(a.iters_per_solve)
        ts = []
        for _ in range(regions):
            ctx.sync()
            t0 = time.perf_counter()
            r = run(a.steps)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        return {"value": a.steps / dt, "unit": "LM outer iterations/s", "ms_per_step": dt / a.steps * 1e3,
                "final_ssr": r.ssr, "g_calls_per_solve": r.g_calls}

    legs = os.environ.get("LSQ_BENCH_GENERIC_LEGS", "device,host").split(",")      # (diagnostics)
    if "device" in legs:
        leg = timed(lambda k: pr2.optimize(LM, LSMR, iterations=k, fetch_x=False))
        leg["g"] = "device kernel, every stored value of both sliced copies rewritten (LSQ_NO_COLSCALE=1, two-pass tail)"
        out["device_g_all_nnz"] = leg

    try:
        if "host" not in legs:
            raise RuntimeError("leg switched off (LSQ_BENCH_GENERIC_LEGS)")
        here = os.path.join(ROOT, "tools", "hostg")
        so = os.path.join(here, "libhostg.so")
        if not os.path.exists(so):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", here])
        H = C.CDLL(so)

        class HostG(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("inner_f", C.c_void_p), ("inner_user", C.c_void_p), ("n", C.c_int),
                        ("nnz", C.c_longlong), ("colptr", C.c_void_p), ("A", C.c_void_p), ("stage", C.c_void_p),
                        ("xh", C.c_void_p), ("threads", C.c_int), ("g_calls", C.c_int), ("fill_seconds", C.c_double),
                        ("g_seconds", C.c_double)]
        assert H.hostg_sizeof() == C.sizeof(HostG), (H.hostg_sizeof(), C.sizeof(HostG))
        H.hostg_f_ptr.restype = C.c_void_p
        H.hostg_g_ptr.restype = C.c_void_p
        stage = C.c_void_p()
        lsq._lib.check(L.lsq_host_alloc(ctx.h, len(nzval) * 8, C.byref(stage)))
        xh = np.zeros(n)
        thr = max(1, min(32, H.hostg_max_threads(), (os.cpu_count() or 1)))
        hg = HostG(ctx.h, C.cast(L.lsq_model_f(), C.c_void_p), pr2.model, n, len(nzval), colptr.ctypes.data, nzval.ctypes.data,
                   stage, xh.ctypes.data, thr, 0, 0.0, 0.0)
        fcb = C.cast(H.hostg_f_ptr(), lsq._lib.F_CALLBACK)
        gcb = C.cast(H.hostg_g_ptr(), lsq._lib.G_CALLBACK)

        class _H:
            h = pr2.J

        def opt_host(k):
            from lsq_amd.api import LeastSquaresResult, _run_native
            st, res, _ = _run_native(ctx, LM, LSMR, _H, pr2.x, pr2.fcur, fcb, gcb, C.cast(C.pointer(hg), C.c_void_p),
                                     1e-8, 1e-8, 1e-8, k, None, None, None, False, n)
            lsq._lib.check(st)
            r = LeastSquaresResult()
            r.iterations, r.ssr, r.g_calls = res.iterations, float(res.ssr), res.g_calls
            return r
        leg = timed(opt_host, regions=3)
        lsq._lib.check(L.lsq_mat_upload_wait(pr2.J))
        leg.update({"g": "host C + OpenMP producer (tools/hostg/hostg.c) -> page-locked buffer -> lsq_mat_set_values_async",
                    "host_threads": thr, "host_fill_ms_per_g": hg.fill_seconds / max(hg.g_calls, 1) * 1e3,
                    "host_g_ms_per_g": hg.g_seconds / max(hg.g_calls, 1) * 1e3,
                    "upload_bytes_per_g": len(nzval) * 8})
        out["host_g_pinned_async"] = leg
        ctx.sync()
        L.lsq_host_free(ctx.h, stage)
    except Exception as e:   # noqa: BLE001
        out["host_g_pinned_async"] = {"error": repr(e)}
    pr2.close()
    return out


def dense_secondary(ctx, lsq, probe=lambda stage: None):
    """BASELINE.json's secondary figures: time per ldiv! of the dense solvers at the C2 / C3 sizes,
    measured after the timed region on fresh N(0,1)/sqrt(m) matrices, next to the host's LAPACK (numpy/scipy, all threads,
    warmed, median of 3) on the same operands, with the useful flops of SURVEY 8d against the fp64 MFMA peak.  Not part of
    `value`."""
    import numpy as np
    out = {}
    if os.environ.get("LSQ_BENCH_DENSE_SLEEP"):       # (diagnostics)
        time.sleep(float(os.environ["LSQ_BENCH_DENSE_SLEEP"]))
    probe("dense_secondary: start")
    rng = np.random.default_rng(lsq.synthetic.BASE_SEED)
    for name, m, n, solver, for_lm in (("c2_cholesky_damped_4096x512", 4096, 512, lsq.Cholesky(), True),
                                       ("c3_qr_16384x2048", 16384, 2048, lsq.QR(), False),
                                       # LM's own use of the QR solver: the stacked operand [J; sqrt(damp)] (18432 x 2048)
                                       ("c3_qr_lm_stacked_18432x2048", 16384, 2048, lsq.QR(), True)):
        is_qr = name.startswith("c3_qr")
        A = rng.standard_normal((m, n)) / np.sqrt(m)
        yh = rng.standard_normal(m)
        J = lsq.DeviceMatrix(ctx, A)
        y = lsq.DeviceVector(ctx, m, yh)
        x = lsq.DeviceVector(ctx, n)
        sv = lsq.AllocatedSolver(J, solver, for_lm=for_lm)
        dmp = lsq.DeviceVector(ctx, n, np.full(n, 0.1)) if for_lm else None

        def go():
            if for_lm:
                sv.ldiv_(x, y, dmp)
            else:
                sv.ldiv_(x, y)
            ctx.sync()

        def fresh():                # operands of the next solve, outside the timed part (the damped solvers may clobber damp)
            if for_lm:
                dmp.set(np.full(n, 0.1))
            ctx.sync()
        fresh(); go()
        fresh(); go()
        times = []
        for _ in range(9):          # median of 9: one stray host/runtime hiccup must not colour the figure
            fresh()
            t0 = time.perf_counter()
            go()
            times.append((time.perf_counter() - t0) * 1e3)
        gpu_ms = sorted(times)[len(times) // 2]

        def host():
            if for_lm:
                return np.linalg.solve(A.T @ A + 0.1 * np.eye(n), A.T @ yh)
            import scipy.linalg as sla   # the reference's algorithm: dgeqp3 + Q'b + triangular solve (full rank here)
            Q, R, piv = sla.qr(A, mode="economic", pivoting=True)
            r = np.empty(n)
            r[piv] = sla.solve_triangular(R, Q.T @ yh)
            return r
        reps = 3 if for_lm else 1      # (dgeqp3 at C3 takes seconds: one warm-up, one timed run)
        ref = host()
        ht = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ref = host()
            ht.append((time.perf_counter() - t0) * 1e3)
        cpu_ms = sorted(ht)[len(ht) // 2]
        err = float(np.linalg.norm(x.get() - ref) / np.linalg.norm(ref))
        if is_qr and for_lm:    # Householder QR of the (m + n) x n stacked operand
            flops, dom = 2 * (m + n) * n * n - 2 * n ** 3 / 3 + 4 * (m + n) * n, "k_qr1_vtb / k_qr1_update (block reflector), k_cqr_pass (panel; look-ahead)"
        elif for_lm:    # J'J m n (n+1) + Cholesky n^3/3 + J'y 2mn + two triangular solves 2 n^2
            flops, dom = m * n * (n + 1) + n ** 3 / 3 + 2 * m * n + 2 * n * n, "k_syrk_mfma (J'J), k_chol_chain"
        else:         # Householder QR 2mn^2 - 2n^3/3 (+ Q'b riding along)
            flops, dom = 2 * m * n * n - 2 * n ** 3 / 3 + 4 * m * n, "k_qr1_vtb / k_qr1_update (block reflector), k_cqr_pass (panel)"
        tf = flops / (gpu_ms * 1e-3) / 1e12
        info = sv.info()
        out[name] = {"ldiv_ms": gpu_ms, "ldiv_ms_min": min(times), "host_lapack_ms": cpu_ms, "host_lapack_runs": reps,
                     "rel_err_vs_host_lapack": err,
                     "roofline": {"bound": "mfma", "useful_flops": flops, "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TFLOPS, "dominant_kernels": dom,
                                  "note": "whole ldiv! (factorisation + solve) over the useful flops of SURVEY 8d; per-kernel "
                                          "split and MFMA counters: " + newest_profile("dense_kernel_summary.md")},
                     "path": {k: info.get(k) for k in ("qr_path", "qr_panel", "chol_path")}}
        J.free()
        probe("dense_secondary: after the ldiv! leg of " + name)
    # time per OUTER iteration of the dense tanh problems (f!, g! and the trust-region bookkeeping included)
    for name, m, n, opt, sol in (("c2_lm_cholesky_4096x512", 4096, 512, lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.CHOLESKY),
                                 ("c3_dogleg_qr_16384x2048", 16384, 2048, lsq._lib.DOGLEG, lsq._lib.QR)):
        pr = lsq.synthetic.TanhProblem(m, n, sparse=False, seed=lsq.synthetic.BASE_SEED, ctx=ctx)
        pr.reset()
        pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=2, fetch_x=False)   # warm-up
        k = 6
        best = None
        for _ in range(3):          # best of 3 identical runs (same trajectory every time)
            pr.reset()
            t0 = time.perf_counter()
            r = pr.optimize(opt, sol, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=k, fetch_x=False)
            ctx.sync()
            ms = (time.perf_counter() - t0) / max(r.iterations, 1) * 1e3
            best = ms if best is None else min(best, ms)
        out[name] = {"outer_iteration_ms": best, "iterations": r.iterations, "ssr": r.ssr}
        pr.close()
        probe("dense_secondary: after the outer loop of " + name)
    return out


def parity_past_convergence(rg, ro, ssr0):
    """The HIP run of rounds 1-5's fixed schedule (8 iterations, zero tolerances) against the oracle's run of the same solve:
    levenberg_marquardt.jl:72-140 / iterative_lsmr.jl:238-259.  That schedule runs on past convergence (the
    reference's own run stops at `useful_iterations`: the timed schedule), where a step changes the objective by ~1e-15
    relative and the gain ratio rho is a quotient of rounding errors of two sums over m squares: an iteration at which the two
    runs disagree about acceptance AND the accepting run changed ssr by less than 1e-12 relative is ROUND-OFF-DECIDED
    (tests/gpu_common.py::compare_until_roundoff, the same rule).  `ok` = up to the first such iteration identical accept
    decisions and LSMR inner counts, a disagreement only where it is excused, identical mul_calls, and at EVERY iteration
    iterates within 1e-8 max(1, |x|_inf) and ssr within 1e-9 relative."""
    import numpy as np
    same_len = rg.iterations == ro.iterations
    k_all = min(rg.iterations, ro.iterations)
    excused, bad_decision, inner_eq = None, None, True
    prev_g = prev_o = ssr0
    for k in range(k_all):
        ag, ao = int(rg.trace["accept"][k]), int(ro.trace["accept"][k])
        sg, so = float(rg.trace["ssr"][k]), float(ro.trace["ssr"][k])
        if excused is None and bad_decision is None:
            if ag != ao:
                s_prev, s_new = (prev_g, sg) if ag else (prev_o, so)
                if abs(s_prev - s_new) <= 1e-12 * s_prev:
                    excused = k
                else:
                    bad_decision = k
            elif int(rg.trace["inner"][k]) != int(ro.trace["inner"][k]):
                inner_eq = False
        prev_g, prev_o = sg, so
    xs = max(1.0, float(np.max(np.abs(ro.trace["x"][:k_all])))) if k_all else 1.0
    # iterates: 1e-8 max(1, |x|inf) up to the first round-off-decided iteration; from there on one run has taken a step the other
    # refused (|dx| of such a step is itself ~1e-8 here): 5e-8
    dxs = [float(np.max(np.abs(np.asarray(rg.trace["x"][i]) - ro.trace["x"][i]))) for i in range(k_all)]
    dx = max(dxs) if dxs else None
    dx_ok = all(d <= (1e-8 if (excused is None or i < excused) else 5e-8) * xs for i, d in enumerate(dxs))
    ssr_rel = float(np.max(np.abs(np.asarray(rg.trace["ssr"][:k_all]) - ro.trace["ssr"][:k_all]) / ro.trace["ssr"][:k_all])) if k_all else None
    ok = bool(same_len and inner_eq and bad_decision is None and rg.mul_calls == ro.mul_calls and dx is not None
              and dx_ok and ssr_rel <= 1e-9)
    return {"ok": ok, "inner_equal": bool(inner_eq), "accept_equal": bool(bad_decision is None),
            "first_roundoff_decided_iteration": None if excused is None else excused + 1,
            "decision_mismatch_at_a_step_that_moved_the_objective": None if bad_decision is None else bad_decision + 1,
            "mul_calls": [int(rg.mul_calls), int(ro.mul_calls)], "max_abs_dx": dx, "ssr_rel": ssr_rel,
            "iterations": [int(rg.iterations), int(ro.iterations)],
            "inner_per_outer": {"hip": [int(v) // 2 for v in rg.trace["inner"]], "cpu": [int(v) // 2 for v in ro.trace["inner"]]},
            "accept": {"hip": [int(v) for v in rg.trace["accept"]], "cpu": [int(v) for v in ro.trace["accept"]]},
            "checker": "oracle/lsq_oracle.c (CPU restatement of the reference), first solve of the cpu_baseline leg",
            "tolerances": {"max_abs_dx": "1e-8*max(1,|x|inf) (5e-8 from the first round-off-decided iteration on)", "ssr_rel": 1e-9,
                           "roundoff_decided": "accept decisions differ and the accepting run moved ssr by <= 1e-12 relative"}}


def cpu_baseline(a, pr, inputs, gpu_ref_run, gpu_fixed_run):
    """The oracle (scalar C port of the reference, 1 thread -- the reference's sparse products and vector loops ARE serial,
    SURVEY 8d) on the SAME inputs and the SAME schedule as the timed region (solves from x0 = 0 with the default tolerances,
    each stopping at convergence), bounded sample, one warm-up solve first -- whose trajectory is compared with the HIP run
    of the same solve (`parity_vs_cpu`: every count, flag, inner count and accept decision identical, iterates to 1e-8, ssr to
    1e-9).  The zero-tolerance 8-iteration solve of rounds 1-5 is compared as well (`past_convergence`, the round-off-decided
    rule of parity_past_convergence).  NB the CPU leg's g! multiplies J = A diag(1 - tanh(x)^2) out entry by entry (orc_tanh_g:
    what a generic g! of the reference does, test/nonlinearleastsquares.jl:47-86); the headline GPU leg keeps a column-scaled
    handle (n factors) -- `value_generic_g_device` in the same line is the GPU doing what the CPU leg does.  Julia itself is
    not in the image: both CPU figures are `"kind": "port"` (the C restatement; the OpenMP variant restructured for all host
    cores: oracle/lsq_oracle_omp.c, CSR mirror for J*v, both copies written by g!, parallel reductions)."""
    import numpy as np
    from oracle import oracle as O
    m, n = a.m, a.n
    colptr, rowval, nzval = inputs
    A = O.Mat(csc=(m, n, colptr, rowval, nzval))
    J = O.Mat(csc=(m, n, colptr, rowval, np.zeros_like(nzval)))
    f, g, ud, keep = O.tanh_model(A, pr.b)

    def solves(total):
        done = inner = 0
        while done < total:  # same schedule as the GPU: solves from x0 = 0 with the default tolerances, the last one cut
            k = min(50, total - done)
            ro = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=k, trace=True, trace_x=False)
            done += int(ro.iterations)
            inner += int(ro.trace["inner"].sum()) // 2
        return inner
    # warm-up (page faults, caches) = the solve whose trajectory is checked against the GPU's
    rod = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=50, trace=True, trace_x=True)
    rgd = gpu_ref_run
    k_all = min(int(rgd.iterations), int(rod.iterations))
    xs = max(1.0, float(np.max(np.abs(rod.trace["x"][:k_all])))) if k_all else 1.0
    dxs = [float(np.max(np.abs(np.asarray(rgd.trace["x"][i]) - rod.trace["x"][i]))) for i in range(k_all)]
    ssr_rel = (float(np.max(np.abs(np.asarray(rgd.trace["ssr"][:k_all]) - rod.trace["ssr"][:k_all]) / rod.trace["ssr"][:k_all]))
               if k_all else None)
    same = int(rgd.iterations) == int(rod.iterations)
    parity = {
        "schedule": "the timed schedule: one solve from x0 = 0 with the default tolerances (1e-8)",
        "iterations": [int(rgd.iterations), int(rod.iterations)], "converged": [bool(rgd.converged), bool(rod.converged)],
        "flags_xfg": [[bool(rgd.x_converged), bool(rgd.f_converged), bool(rgd.g_converged)],
                      [bool(rod.x_converged), bool(rod.f_converged), bool(rod.g_converged)]],
        "counts_f_g_mul": [[int(rgd.f_calls), int(rgd.g_calls), int(rgd.mul_calls)], [int(rod.f_calls), int(rod.g_calls), int(rod.mul_calls)]],
        "inner_equal": bool(same and np.array_equal(rgd.trace["inner"], rod.trace["inner"])),
        "accept_equal": bool(same and np.array_equal(rgd.trace["accept"], rod.trace["accept"])),
        "inner_per_outer": {"hip": [int(v) // 2 for v in rgd.trace["inner"]], "cpu": [int(v) // 2 for v in rod.trace["inner"]]},
        "max_abs_dx": max(dxs) if dxs else None, "ssr_rel": ssr_rel,
        "checker": "oracle/lsq_oracle.c (CPU restatement of the reference), warm-up solve of the cpu_baseline leg",
        "tolerances": {"max_abs_dx": "1e-8*max(1,|x|inf)", "ssr_rel": 1e-9, "counts, flags, inner counts, accept pattern": "identical"}}
    parity["ok"] = bool(same and parity["converged"][0] == parity["converged"][1] and parity["flags_xfg"][0] == parity["flags_xfg"][1]
                        and parity["counts_f_g_mul"][0] == parity["counts_f_g_mul"][1] and parity["inner_equal"]
                        and parity["accept_equal"] and dxs and max(dxs) <= 1e-8 * xs and ssr_rel <= 1e-9)
    parity["useful_iterations"] = int(rod.iterations)
    if gpu_fixed_run is not None:     # rounds 1-5's schedule, two iterations past convergence: reported, and still has to hold
        ro8 = O.optimize(O.LM, O.LSMR, J, np.zeros(n), f, g, ud=ud, iterations=a.iters_per_solve, x_tol=0.0, f_tol=0.0,
                         g_tol=0.0, trace=True, trace_x=True)
        parity["past_convergence"] = parity_past_convergence(gpu_fixed_run, ro8, float(np.sum(np.asarray(pr.b) ** 2)))
        parity["ok"] = bool(parity["ok"] and parity["past_convergence"]["ok"])
    t0 = time.perf_counter()
    inner = solves(a.cpu_steps)
    dt = time.perf_counter() - t0
    # CPU J*v bandwidth on the same matrix (algorithmic bytes, same formula)
    x = np.random.default_rng(0).standard_normal(n)
    t1 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        O.mul(A, x, 1.0, 1.0, np.zeros(m))
    t_mv = (time.perf_counter() - t1) / reps
    nnz = len(nzval)
    out = {"value": a.cpu_steps / dt, "unit": "LM outer iterations/s", "cores": 1, "kind": "port",
           "sample": "%d LM outer iterations (%d LSMR inner; solves from x0=0 with the default tolerances, %d iterations each: "
                     "the timed schedule) of the same C4 problem with oracle/lsq_oracle.c (C restatement of the reference; no "
                     "Julia in the image), 1 thread, after one warm-up solve, %.1f s; its g! multiplies J = A diag(1 - tanh(x)^2) "
                     "out entry by entry (like for like: value_generic_g_device)"
                     % (a.cpu_steps, inner, int(rod.iterations), dt),
           "lsmr_inner_iterations_per_sec": inner / dt, "host_cores_available": os.cpu_count(),
           "jv_GBps": (12 * nnz + 4 * (m + 1) + 8 * n + 16 * m) / t_mv / 1e9}
    try:   # the all-cores figure (a restructured, OpenMP-parallel port: labelled, not the reference's serial path)
        b = np.ascontiguousarray(pr.b, dtype=np.float64)
        k = int(rod.iterations)       # (the OpenMP port runs a fixed count with zero tolerances: the reference run's length)
        best = None
        ncpu = os.cpu_count() or 1
        for thr in sorted({t for t in (16, 32, 64, ncpu // 2, ncpu) if 1 < t <= ncpu}):
            op = O.OmpProblem(m, n, colptr, rowval, nzval, b, threads=thr)     # (layout build: not timed, as on the GPU)
            try:
                op.run(np.zeros(n), k)                                          # warm-up
                nsolve = 4
                t0 = time.perf_counter()
                inner_o = sum(op.run(np.zeros(n), k)[2] for _ in range(nsolve))
                dto = time.perf_counter() - t0
            finally:
                op.close()
            rec = {"value": nsolve * k / dto, "cores": thr, "lsmr_inner_iterations_per_sec": inner_o / dto,
                   "sample": "%d LM outer iterations (%d LSMR inner) of the same C4 problem with oracle/lsq_oracle_omp.c on %d "
                             "threads (OMP_PROC_BIND=close, first-touch placement), %.1f s" % (nsolve * k, inner_o, thr, dto)}
            if best is None or rec["value"] > best["value"]:
                best = rec
        out["all_cores"] = dict(best, unit="LM outer iterations/s", kind="port (OpenMP, best thread count of those tried)")
    except Exception as e:   # noqa: BLE001
        out["all_cores"] = {"value": None, "sample": "failed: %r" % (e,)}
    return out, parity


def guarded_main():
    """Every rank's failure ends with ONE line of reason on stderr and a non-zero exit code (under torch.distributed.run the
    launcher then stops the other ranks and reports which one failed first)."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:   # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        print("bench: rank %s FAILED: %s: %s" % (os.environ.get("RANK", "0"), type(e).__name__, str(e).splitlines()[0] if str(e) else ""),
              file=sys.stderr)
        sys.stderr.flush()
        os._exit(1)      # (not sys.exit: a rank stuck in a collective's teardown must not keep the job alive)


if __name__ == "__main__":
    guarded_main()
