#!/usr/bin/env python3
"""One solve of the bench schedule as a timeline: every kernel of the k-th solve of a rocprofv3 kernel trace with its start
offset, duration and the gap in front of it (us).  usage: step_timeline.py kernel_trace.csv [solve index, default 3]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# a solve starts with the residual at x0 (k_sell_rows<EpiResidual...> or k_tanh in front of it)
starts = [i for i, e in enumerate(ev) if "k_first_nonfinite" in e[2]]
a, b = starts[which], starts[which + 1]
t0 = ev[a][0]
prev_end = ev[a][0]
busy = 0
for s, e, n in ev[a:b]:
    short = n.split("(")[0].replace("void ", "")[:48]
    print("%9.2f  dur %7.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    prev_end = max(prev_end, e)
    busy += e - s
print("solve span %.1f us, busy %.1f us, kernels %d" % ((ev[b][0] - t0) / 1e3, busy / 1e3, b - a))
