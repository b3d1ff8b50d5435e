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
# MFMA counters of the same solves (their own pass: --pmc with --kernel-trace only)
run_pmc() {
    local d=$OUT/$1_pmc
    mkdir -p $d
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $d -o p --output-format csv -- python tools/dense_bench.py $2 > $d/run.log 2>&1
}
run c2 - chol:4096:512:1
run c3 - qr:16384:2048:0
run c3_lm - qr:16384:2048:1
run c3_pivot LSQ_QR_ALWAYS_PIVOT=1 qr:16384:2048:0
run c3_steps LSQ_QR1_NO_CHOLQR=1 qr:16384:2048:0
run_pmc c2 chol:4096:512:1
run_pmc c3 qr:16384:2048:0
python - <<'PY'
import csv, glob, re
out = ["# rocprofv3 --kernel-trace --stats of the dense solvers (`tools/dense_bench.py`, 4 solves each: 1 warm-up + 3 timed)", ""]
cases = [("c2", "C2: damped Cholesky `ldiv!`, 4096x512 (`chol:4096:512:1`)"),
         ("c3", "C3: QR `ldiv!`, 16384x2048 (`qr:16384:2048:0`; full-rank certificate path)"),
         ("c3_lm", "C3, LM's stacked operand: QR `ldiv!` with damping, (16384+2048)x2048 (`qr:16384:2048:1`)"),
         ("c3_pivot", "C3 with the pivoted sweep forced (`LSQ_QR_ALWAYS_PIVOT=1 qr:16384:2048:0`; Householder-step panels)"),
         ("c3_steps", "C3 with the column-by-column Householder panel of round 1 (`LSQ_QR1_NO_CHOLQR=1 qr:16384:2048:0`)")]
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
# ---- MFMA counters (north_star: "MFMA-util counters against CDNA4 peak") --------------------------------------------
import collections
NSIMD = 256 * 4
NXCD = 8             # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = GRBM_GUI_ACTIVE / 8 (the MEASURED clock)
out += ["## fp64 MFMA counters of the MFMA kernels (`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES "
        "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE`, own pass)", "",
        "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES summed over the device / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of "
        "all SIMD-cycles of the launch in which an MFMA was executing, with the launch's cycles MEASURED in the same pass "
        "(`clock` = GRBM_GUI_ACTIVE / 8 / duration: what the shader clock really was, no assumed frequency; for launches shorter "
        "than ~30 us the counter also covers the dispatch's ramp outside the kernel's timestamps, so the derived clock overstates "
        "-- 3-5 GHz -- and MFMA-busy understates there: read it for the 40+ us kernels).  `MOPS_F64` x 512 = fp64 MFMA flops issued (a 16x16x4 "
        "f64 MFMA = 2048 flops = 4 MOPS); `TFLOP/s` = that / duration, against the 78.6 TFLOP/s fp64 matrix peak.", ""]
for name, title in (("c2_pmc", "C2 (`chol:4096:512:1`)"), ("c3_pmc", "C3 (`qr:16384:2048:0`)")):
    f = glob.glob("gpurun_out/denseprof/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out += ["### " + title, "", "(no counters)", ""]
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"])
        if key not in cnt[k]:
            cnt[k].add(key)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out += ["### " + title, "", "| kernel | launches | avg us | clock GHz | MFMA busy | MOPS_F64 per launch | TFLOP/s (MFMA) | % of 78.6 |", "|---|---|---|---|---|---|---|---|"]
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)):
        mops, busy, n = agg[k].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0), agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), len(cnt[k])
        if mops <= 0 or dur[k] <= 0:
            continue
        tf = mops * 512 / dur[k] / 1e12
        cyc = agg[k].get("GRBM_GUI_ACTIVE", 0.0) / NXCD
        if cyc <= 0:
            continue
        out.append("| `%s` | %d | %.1f | %.2f | %.1f %% | %.3g | %.1f | %.1f %% |" % (k[:60], n, dur[k] / n * 1e6, cyc / dur[k] / 1e9,
                                                                           100 * busy / (cyc * NSIMD), mops / n, tf, 100 * tf / 78.6))
    out.append("")
open("gpurun_out/denseprof/dense_kernel_summary.md", "w").write("\n".join(out))
print("\n".join(out[-40:]))
PY
