#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per kernel, all launches and the *working* launches.

LSMR runs at most LSQ_LOOKAHEAD iterations ahead of the device-side stop test; launches queued
behind a finished solve return on their first instruction (3-5 us).  `--stats` averages those
no-op launches together with the working ones, so this table separates them (for the kernels that can
exit early, a launch counts as "working" when it lasts longer than 60 % of the kernel's 90th-percentile
duration, provided the shortest launch is below half of that percentile)."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"]
        if "k_lsmr_fused" in name:      # the launch comes whole, commit-only (a few workgroups) or product-only: keep them apart
            name += " grid=%s" % r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        rows[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | launches | avg us (all) | working launches | avg us (working) | min us | max us | % of GPU time |")
print("|---|---|---|---|---|---|---|---|")
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    mn = min(v)
    early = any(t in name for t in ("k_seg_", "k_sell_", "k_combine", "k_lsmr_update", "k_lsmr_fused"))
    p90 = sorted(v)[min(len(v) - 1, int(0.9 * len(v)))]
    work = [d for d in v if d > (0.8 if "k_lsmr_fused" in name else 0.6) * p90] if early and mn < 0.5 * p90 else v
    work = work or v
    print("| `%s` | %d | %.2f | %d | %.2f | %.2f | %.2f | %.1f |" % (
        name[:100], len(v), sum(v) / len(v), len(work), sum(work) / len(work), mn, max(v), 100 * sum(v) / tot))
