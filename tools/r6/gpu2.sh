#!/bin/bash
# round 6, call 2: Q1 form + fused Gram + vtb LDS reservation: A/B, kernel traces, new tests
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_2; mkdir -p $O
( timeout 900 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "qr or cholqr or serial or partitioned" 2>&1 | tail -n 8 ) > $O/pytest_b.log
( timeout 600 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c3 or kat or factor_model or bounds or finite_difference" 2>&1 | tail -n 12 ) > $O/pytest_a.log
( timeout 300 python -m pytest tests/test_zz_gpu_stress.py -x -q 2>&1 | tail -n 8 ) > $O/pytest_zz.log
( timeout 900 python -m pytest tests/test_sharding.py -x -q -k "eight or world8" 2>&1 | tail -n 12 ) > $O/pytest_shard.log
C="qr:16384:2048:0 qr:16384:2048:1 qr:4096:512:0 qr:3000:700:1"
for r in 1 2 3; do
  python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/DEFAULT     /'
  LSQ_QR_VTB_LDS=0 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/NO_VTB_LDS  /'
  LSQ_QR_NO_FUSED_GRAM=1 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/NO_GRAM     /'
  LSQ_QR_LOOKAHEAD=0 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/NO_LOOKAHEAD/'
  LSQ_QR_CQR_PASS2=1 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/PASS2       /'
  LSQ_QR_CQR_PASS2=1 LSQ_QR_VTB_LDS=0 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/PASS2_R5    /'
done > $O/ab.txt 2>&1
QRPROF_OUT=$O/prof_default bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_default.txt 2>&1
LSQ_QR_LOOKAHEAD=0 QRPROF_OUT=$O/prof_nola bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_nola.txt 2>&1
LSQ_QR_CQR_PASS2=1 LSQ_QR_VTB_LDS=0 QRPROF_OUT=$O/prof_r5 bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_r5.txt 2>&1
rm -rf $O/prof_default $O/prof_nola $O/prof_r5
for f in pytest_b pytest_a pytest_zz pytest_shard; do echo "== $f"; tail -n 5 $O/$f.log; done
sort $O/ab.txt | awk '{k=$1" "$3" "$4; s[k]+=$5; n[k]++} END {for (k in s) printf "%-45s %.3f ms\n", k, s[k]/n[k]}' | sort
cat $O/prof_default.txt; echo; cat $O/prof_nola.txt; echo; cat $O/prof_r5.txt
