#!/usr/bin/env python3
"""C4 (10^6 x 10^4, nnz 10^7) LM+LSMR with the row-sharded loop's collectives switched on at ONE rank (a GPU box has one
device): what the in-stream all-reduces cost per outer iteration -- direct RCCL (liblsqrccl.so) vs the host-staged hook vs
the unsharded loop.  The multi-rank version needs the 8-GPU node."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsq_amd as lsq  # noqa: E402
from lsq_amd import rowshard as RS  # noqa: E402

m, n, pc = 1_000_000, 10_000, 1000
ctx = lsq.Context(0)
inputs = lsq.synthetic.sparse_inputs(m, n, pc, lsq.synthetic.BASE_SEED)
pr = lsq.synthetic.TanhProblem(m, n, sparse=True, per_col=pc, ctx=ctx, inputs=inputs)
LM, LSMR = lsq._lib.LEVENBERG_MARQUARDT, lsq._lib.LSMR
hooks = {"unsharded": None, "rccl (direct ncclAllReduce on the stream)": RS.RcclRowAllreduce(0, 1, None),
         "host-staged": RS.HostStagedRowAllreduce(ctx, None)}
for name, hook in hooks.items():
    best = None
    for rep in range(6):
        pr.reset()
        ctx.sync()
        t0 = time.perf_counter()
        kw = {} if hook is None else dict(row_allreduce=hook.callback, row_allreduce_user=hook.user, global_rows=m)
        r = pr.optimize(LM, LSMR, x_tol=0.0, f_tol=0.0, g_tol=0.0, iterations=8, fetch_x=False, **kw)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("%-45s %.3f ms per outer iteration (%d inner), ssr %.12e%s" % (name, best / 8 * 1e3, r.lsmr_iterations, r.ssr,
          "" if hook is None else "  collectives: %d calls / %d doubles in the last solve x6" % hook.stats()))
