#!/bin/bash
# HBM traffic of the kernels of the C3 QR solve (two PMC passes, FETCH_SIZE and WRITE_SIZE, as MI355X_MICROARCH.md
# prescribes; run through gpurun from the repo root).  Writes gpurun_out/c3_traffic/c3_traffic.md: per kernel the average
# launch, 2*FETCH+WRITE bytes (the gfx950 correction calibrated in profiles/r03/pmc_traffic.md) and the rate they imply.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/c3_traffic
rm -rf $OUT && mkdir -p $OUT
CASE=${1:-qr:16384:2048:0}
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python tools/dense_bench.py $CASE > $OUT/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python tools/dense_bench.py $CASE > $OUT/w.log 2>&1
python - "$CASE" <<'PY'
import csv, glob, collections, sys
def load(pat, name):
    d = collections.defaultdict(list)
    for path in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                d[r["Kernel_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return d
F = load("gpurun_out/c3_traffic/f/**/*counter_collection.csv", "FETCH_SIZE")
W = load("gpurun_out/c3_traffic/w/**/*counter_collection.csv", "WRITE_SIZE")
rows = []
for k in F:
    n = len(F[k])
    f = sum(v for v, _ in F[k]) / n
    t = sum(t for _, t in F[k]) / n
    w = sum(v for v, _ in W.get(k, [(0.0, 0.0)])) / max(1, len(W.get(k, [])))
    mb = (2 * f + w) * 1024 / 1e6
    rows.append((n * t, k, n, t, f, w, mb))
rows.sort(reverse=True)
out = ["# HBM traffic of the kernels of `tools/dense_bench.py %s` (4 solves; `tools/c3_traffic.sh`)" % sys.argv[1], "",
       "FETCH_SIZE / WRITE_SIZE in KiB per launch (averages over all launches of the kernel); bytes = 2 x FETCH + WRITE (gfx950: FETCH_SIZE",
       "counts half of the bytes of wide coalesced reads, calibrated in pmc_traffic.md).", "",
       "| kernel | launches | avg us | FETCH KiB | WRITE KiB | MB per launch | TB/s |", "|---|---|---|---|---|---|---|"]
for tot, k, n, t, f, w, mb in rows[:14]:
    out.append("| `%s` | %d | %.1f | %.0f | %.0f | %.1f | %.2f |" % (k[:60], n, t, f, w, mb, mb / t if t else 0.0))
open("gpurun_out/c3_traffic/c3_traffic.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
