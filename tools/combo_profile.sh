cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/comboprof
rm -rf $OUT && mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python tools/dense_combos.py 1000000x20 > $OUT/run.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), ("%.2f ms" % (int(r["TotalDurationNs"])/1e6)).rjust(10), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(10), r["Percentage"])
PY
