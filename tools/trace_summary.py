#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per kernel, all launches and the *working* launches.

LSMR runs at most LSQ_LOOKAHEAD iterations ahead of the device-side stop test; launches queued
behind a finished solve return on their first instruction (3-5 us).  `--stats` averages those
no-op launches together with the working ones, so this table separates them (a launch counts as
"working" when it lasts longer than 3x the kernel's shortest launch, or the kernel never exits
early)."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | launches | avg us (all) | working launches | avg us (working) | min us | max us | % of GPU time |")
print("|---|---|---|---|---|---|---|---|")
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    mn = min(v)
    early = any(t in name for t in ("k_seg_", "k_sell_", "k_combine", "k_lsmr_update"))
    work = [d for d in v if (d > 3 * mn and d > 8.0)] if early and mn < 8.0 else v
    work = work or v
    print("| `%s` | %d | %.2f | %d | %.2f | %.2f | %.2f | %.1f |" % (
        name[:100], len(v), sum(v) / len(v), len(work), sum(work) / len(work), mn, max(v), 100 * sum(v) / tot))
