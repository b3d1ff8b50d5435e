"""The one global exchange of sharded runs: independent problems, one per rank, and a
single all-reduce per OUTER iteration carrying {sum of ssr, max of gradient norms, all-converged}.

A mixed sum/max/min reduction is packed into ONE SUM all-reduce of world+2 doubles: slot 0 = ssr,
slot 1 = converged count, slot 2+rank = this rank's gradient norm (zeros elsewhere), so every rank
recovers the max locally.  Backend `nccl` is RCCL over xGMI on MI355X; `gloo` is used by the CPU
tests.  The payload is 8*(world+2) bytes: latency-only, never inside the LSMR loop.
"""
import sys

from . import _lib


def exchange(dist, rank, world, ssr, gnorm, converged, buf, host=None):
    """Returns (sum ssr, max gnorm, all converged).  `buf` is a float64 tensor of world+2 elements on
    the backend's device; `host` an optional (pinned) staging tensor of the same shape."""
    import torch
    if host is None:
        host = torch.zeros(world + 2, dtype=torch.float64)
    host.zero_()
    host[0] = ssr
    host[1] = 1.0 if converged else 0.0
    host[2 + rank] = gnorm
    buf.copy_(host)
    dist.all_reduce(buf)
    host.copy_(buf)
    return float(host[0]), float(host[2:].max()), bool(float(host[1]) >= world - 0.5)


def make_allreduce_callback(dist, rank, world, device):
    """ctypes callback for lsq_options.allreduce (vals = {ssr, maxabs_gr, converged})."""
    import torch
    buf = torch.zeros(world + 2, dtype=torch.float64, device=device)
    host = torch.zeros(world + 2, dtype=torch.float64)
    if str(device).startswith("cuda"):
        host = host.pin_memory()

    def _cb(vals, count, _user):
        try:
            s, g, allc = exchange(dist, rank, world, vals[0], vals[1], vals[2] > 0.5, buf, host)
            vals[0], vals[1], vals[2] = s, g, 1.0 if allc else 0.0
            return 0
        except Exception as e:  # pragma: no cover
            print("allreduce callback failed:", e, file=sys.stderr)
            return 1

    return _lib.ALLREDUCE_CALLBACK(_cb)
