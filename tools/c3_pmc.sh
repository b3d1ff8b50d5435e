#!/bin/bash
# where do the idle fp64-MFMA cycles of the C3 trailing update go?
#  (a) the pure-MFMA micro-kernel under PMC: MFMA-busy fraction and the MEASURED clock (GRBM_GUI_ACTIVE / 8 XCDs / duration)
#  (b) stall-side counters of the C3 QR kernels, in passes of <= 8 SQ counters
# Results: gpurun_out/c3pmc/summary.md (tools/publish_profiles.py copies it into profiles/<round>/c3_mfma_stalls.md).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/c3pmc
rm -rf $OUT && mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma64_bench.hip -o /tmp/mfma64_bench 2> $OUT/build.err
/tmp/mfma64_bench > $OUT/micro_plain.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $OUT/micro -o p --output-format csv -- /tmp/mfma64_bench > $OUT/micro_pmc.txt 2>&1
pass() {  # name, counters...
    local d=$OUT/$1; shift
    mkdir -p $d
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -o p --output-format csv -- python tools/dense_bench.py qr:16384:2048:0 > $d/run.log 2>&1
}
pass c3_a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass c3_b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass c3_c SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections
out = ["# fp64 MFMA issue rate and stall counters (tools/c3_pmc.sh)", ""]
def table(path, want=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.defaultdict(set)
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if want and not any(w in k for w in want):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg, dur, cnt
f = glob.glob("gpurun_out/c3pmc/micro/**/*counter_collection.csv", recursive=True)
out += ["## (a) pure-MFMA micro-kernel (`tools/micro/mfma64_bench.hip`: NACC independent v_mfma_f64_16x16x4 accumulators per wave, 4 waves per block, constant operands)", "",
        "```"] + [l.rstrip() for l in open("gpurun_out/c3pmc/micro_plain.txt")] + ["```", ""]
if f:
    # per dispatch: one row each
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        d = per[(r["Kernel_Name"], int(r["Dispatch_Id"]))]
        d[r["Counter_Name"]] = float(r["Counter_Value"]); d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        d["grid"] = int(r["Grid_Size"]) if "Grid_Size" in r else 0
    out += ["| kernel | dispatch | grid (threads) | ms | clock GHz (GRBM_GUI_ACTIVE / 8 / duration) | MFMA busy (busy cycles / (cycles x 1024 SIMDs)) | TFLOP/s (MOPS_F64 x 512 / duration) |", "|---|---|---|---|---|---|---|"]
    for (k, did), d in sorted(per.items(), key=lambda kv: kv[0][1]):
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc <= 0: continue
        out.append("| `%s` | %d | %d | %.3f | %.2f | %.1f %% | %.1f |" % (k[:40], did, d["grid"], d["dur"] * 1e3, cyc / d["dur"] / 1e9,
                   100 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512 / d["dur"] / 1e12))
    out.append("")
out += ["## (b) C3 QR `ldiv!` 16384x2048: SQ counters of the MFMA kernels (sums over all waves of all launches; quad-cycle units for SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*)", ""]
want = ("k_qr1_vtb", "k_qr1_update", "k_cqr_pass", "k_cqr_tw", "k_tri_level", "k_syrk")
for name in ("c3_a", "c3_b", "c3_c"):
    f = glob.glob("gpurun_out/c3pmc/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out += ["(%s: no counters)" % name, ""]; continue
    agg, dur, cnt = table(f[0], want)
    names = sorted({c for k in agg for c in agg[k]})
    out += ["| kernel | launches | avg us | " + " | ".join(names) + " |", "|---|---|---|" + "---|" * len(names)]
    for k in sorted(agg, key=lambda k: -dur[k]):
        out.append("| `%s` | %d | %.1f | " % (k[:46], len(cnt[k]), dur[k] / len(cnt[k]) * 1e6) + " | ".join("%.4g" % (agg[k][c] / len(cnt[k])) for c in names) + " |")
    out.append("")
    if name == "c3_a":
        out += ["Derived (per kernel): share of wave time by state, MFMA busy, clock:", "", "| kernel | WAIT_ANY (parked: s_waitcnt / barrier) | WAIT_INST_ANY (issue stall) | of which LDS issue stall | ACTIVE_INST_ANY | MFMA busy | clock GHz | waves per launch |", "|---|---|---|---|---|---|---|---|"]
        for k in sorted(agg, key=lambda k: -dur[k]):
            a = agg[k]; wc = a.get("SQ_WAVE_CYCLES", 0)
            if wc <= 0: continue
            cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
            out.append("| `%s` | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.2f | %.0f |" % (k[:46], 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc,
                       100 * a["SQ_WAIT_INST_LDS"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc, 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0, cyc / dur[k] / 1e9 if dur[k] else 0, a["SQ_WAVES"] / len(cnt[k])))
        out.append("")
open("gpurun_out/c3pmc/summary.md", "w").write("\n".join(out))
print("\n".join(out))
PY
