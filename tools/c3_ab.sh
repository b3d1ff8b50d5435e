cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c3
./build/micro/cumask_probe 32 > gpurun_out/c3/cumask_32.txt 2>&1; cat gpurun_out/c3/cumask_32.txt
python tools/c3_outer.py 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c3/outer -o o --output-format csv -- python tools/c3_outer.py > gpurun_out/c3/outer.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/c3/outer/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.2f ms" % (int(r["TotalDurationNs"])/1e6)).rjust(10), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(10), r["Percentage"])
PY
for w in 0 1; do echo "== LSQ_QR_UPDATE_W=$w"; LSQ_QR_UPDATE_W=$w python tools/dense_bench.py qr:16384:2048:0 qr:16384:2048:0; done
LSQ_QR_UPDATE_W=1 LSQ_QR_UPDATE_TPW=16 python tools/dense_bench.py qr:16384:2048:0
LSQ_QR_UPDATE_W=1 LSQ_QR_UPDATE_TPW=4 python tools/dense_bench.py qr:16384:2048:0
LSQ_QR_UPDATE_W=1 bash tools/qr_profile.sh qr:16384:2048:0 2>&1 | tail -22
LSQ_QR_UPDATE_W=1 timeout 900 python -m pytest tests/test_b_gpu_kernels.py -m gpu -q -x -k "qr" -p no:cacheprovider 2>&1 | tail -5
