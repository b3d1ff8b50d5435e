#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB.  gfx950 correction: FETCH_SIZE
reports half of the bytes of wide coalesced reads (guide: 'double it'); the factor is calibrated
below on kernels of this run whose traffic is known (column scaling: 80 MB read + 80 MB written;
generic J*v)."""
import csv, sys, collections
def load(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return d
F = load(sys.argv[1], "FETCH_SIZE"); W = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | working launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB (raw) | 2*F+W MB | F+W MB |")
print("|---|---|---|---|---|---|")
for k in sorted(F, key=lambda k: -sum(v for v, _ in F[k])):
    # working launches: longer than 8 us; for k_lsmr_fused (round 5) longer than 18 us -- its cautious launches that commit a stop
    # (10-12 us) and the aborted streams of a finished solve move a fraction of the bytes and are not what the roofline prices
    tmin = 18.0 if "k_lsmr_fused" in k else 8.0
    fv = [v for v, t in F[k] if t > tmin] or [v for v, t in F[k]]
    wv = [v for v, t in W.get(k, []) if t > tmin] or [v for v, t in W.get(k, [(0, 0)])]
    f, w = sum(fv) / len(fv), sum(wv) / len(wv)
    print("| `%s` | %d | %.0f | %.0f | %.1f | %.1f |" % (k[:70], len(fv), f, w, (2 * f + w) * 1024 / 1e6, (f + w) * 1024 / 1e6))
