#!/bin/bash
# rocprofv3 --kernel-trace --stats of the dense solvers (run through gpurun from the repo root); writes
# gpurun_out/denseprof/dense_kernel_summary.md, which tools/publish_profiles.py copies into profiles/<round>/.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/denseprof
rm -rf $OUT && mkdir -p $OUT
run() {  # name, env assignment or "-", case
    local d=$OUT/$1
    mkdir -p $d
    if [ "$2" = "-" ]; then
        timeout 300 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python tools/dense_bench.py $3 > $d/run.log 2>&1
    else
        env $2 timeout 300 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python tools/dense_bench.py $3 > $d/run.log 2>&1
    fi
}
run c2 - chol:4096:512:1
run c3 - qr:16384:2048:0
run c3_lm - qr:16384:2048:1
run c3_pivot LSQ_QR_ALWAYS_PIVOT=1 qr:16384:2048:0
python - <<'PY'
import csv, glob, re
out = ["# rocprofv3 --kernel-trace --stats of the dense solvers (`tools/dense_bench.py`, 4 solves each: 1 warm-up + 3 timed)", ""]
cases = [("c2", "C2: damped Cholesky `ldiv!`, 4096x512 (`chol:4096:512:1`)"),
         ("c3", "C3: QR `ldiv!`, 16384x2048 (`qr:16384:2048:0`; full-rank certificate path)"),
         ("c3_lm", "C3, LM's stacked operand: QR `ldiv!` with damping, (16384+2048)x2048 (`qr:16384:2048:1`)"),
         ("c3_pivot", "C3 with the pivoted sweep forced (`LSQ_QR_ALWAYS_PIVOT=1 qr:16384:2048:0`)")]
for name, title in cases:
    d = "gpurun_out/denseprof/" + name
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    log = [l.strip() for l in open(d + "/run.log") if re.match(r"^(QR|Cholesky)\s", l)]
    out += ["## " + title, ""] + ["`%s`" % l for l in log] + [""]
    if not f:
        out += ["(no kernel stats)", ""]
        continue
    out += ["| kernel | launches per solve | time per solve | avg per launch |", "|---|---|---|---|"]
    for r in list(csv.DictReader(open(f[0])))[:14]:
        out.append("| `%s` | %.1f | %.3f ms | %.1f us |" % (r["Name"][:70], int(r["Calls"]) / 4, int(r["TotalDurationNs"]) / 4e6,
                                                        float(r["AverageNs"]) / 1e3))
    out.append("")
open("gpurun_out/denseprof/dense_kernel_summary.md", "w").write("\n".join(out))
print("\n".join(out[:40]))
PY
