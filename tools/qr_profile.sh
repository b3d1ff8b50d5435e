#!/bin/bash
# Kernel-time breakdown of one dense solve (run through gpurun): tools/qr_profile.sh qr:16384:2048:0
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${QRPROF_OUT:-gpurun_out/qrprof}
rm -rf $OUT && mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o qr --output-format csv -- python tools/dense_bench.py "$@" > $OUT/run.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.2f ms" % (int(r["TotalDurationNs"])/1e6)).rjust(10), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(10), r["Percentage"])
PY
tail -3 $OUT/run.log
