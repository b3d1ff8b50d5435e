"""Which runs of the MINPACK grid keep the oracle's counts on the FAST kernels (tree reductions)?  Prints one line per
run that differs, with its count-stable flag (tests/golden/count_stable.json).  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P  # noqa: E402
import lsq_amd as lsq  # noqa: E402
from gpu_common import OPT, SOL, gpu_run, oracle_run  # noqa: E402

cs = json.load(open(os.path.join(ROOT, "tests", "golden", "count_stable.json")))
probs = {P.label(p): p for p in P.minpack_all()}
lsq.set_exact(False)
bad = []
for rec in cs["runs"]:
    p = probs[rec["problem"]]
    rg = gpu_run(p, OPT[rec["optimizer"]][0], SOL[rec["solver"]][0](), rec["sparse"])
    ro = oracle_run(p, OPT[rec["optimizer"]][1], SOL[rec["solver"]][1], rec["sparse"])
    same = (rg.iterations, rg.f_calls, rg.g_calls, rg.mul_calls) == (ro.iterations, ro.f_calls, ro.g_calls, ro.mul_calls)
    if not same:
        bad.append((rec["problem"], rec["optimizer"], rec["solver"], rec["sparse"], rec["stable"], rec["robust"]))
        print(rec["problem"], rec["optimizer"], rec["solver"], rec["sparse"], "stable" if rec["stable"] else "unstable", "robust" if rec["robust"] else "-",
              (rg.iterations, rg.mul_calls), (ro.iterations, ro.mul_calls), flush=True)
print("differ:", len(bad), "of", len(cs["runs"]), "; inside the count-stable set:", sum(b[4] for b in bad),
      "; inside the robust set:", sum(b[5] for b in bad))
