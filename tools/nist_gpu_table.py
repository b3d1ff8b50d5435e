#!/usr/bin/env python3
"""The NIST classification table (tests/nist_cases.py) through the HIP path, reference-order kernels and fast kernels;
prints the runs whose class differs from tests/golden/nist_outcomes.json and the time per configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lsq_amd as lsq  # noqa: E402
import nist  # noqa: E402
import nist_cases as NC  # noqa: E402

fx = NC.load_outcomes()
run = NC.hip_runner(lsq)
probs = nist.problems()
out = {}
for exact in (True, False):
    lsq.set_exact(exact)
    for (opt, solver, storage) in NC.CONFIGS:
        for jac in ("analytic", "central"):
            key = NC.config_key(opt, solver, storage, jac)
            t0 = time.time()
            diffs = {}
            for p in probs:
                for si in range(len(p.starts)):
                    try:
                        cls, ev = NC.classify(p, si, opt, solver, storage, jac, run, fx["plateaus"])
                    except AssertionError as e:
                        cls, ev = "ASSERT", {"msg": str(e)[:300]}
                    want = fx["classes"][key]["%s/%d" % (p.name, si)]["class"]
                    if cls != want:
                        diffs["%s/%d" % (p.name, si)] = [want, cls, ev]
            out["%s exact=%d" % (key, exact)] = diffs
            print("%-34s exact=%d  %.1fs  diffs: %s" % (key, exact, time.time() - t0, diffs), flush=True)
lsq.set_exact(True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "nist_gpu_table.json"), "w"), indent=1)
