#!/usr/bin/env python3
"""Longest kernels and longest gaps between consecutive kernels in a rocprofv3 kernel trace (directory or csv)."""
import csv, glob, sys
f = sys.argv[1] if sys.argv[1].endswith(".csv") else glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows]
print("kernels:", len(ev))
print("longest kernels:")
for s, e, n in sorted(ev, key=lambda t: t[0] - t[1])[:6]:
    print("  %10.3f ms  at %.3f ms  %s" % ((e - s) / 1e6, (s - ev[0][0]) / 1e6, n))
gaps = [(ev[i + 1][0] - max(x[1] for x in ev[max(0, i - 3):i + 1]), i) for i in range(len(ev) - 1)]
print("longest gaps:")
for g, i in sorted(gaps, reverse=True)[:6]:
    print("  %10.3f ms  at %.3f ms  after %s  before %s" % (g / 1e6, (ev[i][1] - ev[0][0]) / 1e6, ev[i][2][:40], ev[i + 1][2][:40]))
