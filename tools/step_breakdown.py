#!/usr/bin/env python3
"""Per-kernel time per outer step inside bench.py's timed region, from a rocprofv3 kernel trace CSV."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows])
names = [e[2] for e in ev]
damp = [i for i, n in enumerate(names) if n.startswith('k_lm_damp') or n.startswith('k_lm_lsmr_setup')]
seg = ev[damp[warm]:damp[warm + steps]]
d = collections.defaultdict(list)
for a, b, n in seg:
    d[n.split('(')[0][:60]].append((b - a) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%8.1f us/step %5.2f x/step avg %7.2f min %7.2f max %7.2f  %s" % (sum(v) / steps, len(v) / steps, sum(v) / len(v), min(v), max(v), k))
print("busy %.1f us/step, span %.1f us/step" % (tot / steps, (seg[-1][1] - seg[0][0]) / 1e3 / steps))
