#!/bin/bash
# round 6, call 3: Neumann k_cqr_top + 16-column k_cqr_tw_q1: tests, A/B, kernel trace
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_3; mkdir -p $O
( timeout 900 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "qr or cholqr or serial or partitioned" 2>&1 | tail -n 8 ) > $O/pytest_b.log
( timeout 600 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c3 or minpack_trajectories or operator_level or golden" 2>&1 | tail -n 12 ) > $O/pytest_a.log
( timeout 300 python -m pytest tests/test_zz_gpu_stress.py -x -q 2>&1 | tail -n 8 ) > $O/pytest_zz.log
C="qr:16384:2048:0 qr:16384:2048:1 qr:4096:512:0 qr:3000:700:1 qr:20000:1000:0"
for r in 1 2 3; do
  python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/DEFAULT     /'
  LSQ_QR_TOP_LU=1 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/TOP_LU      /'
  LSQ_QR_LOOKAHEAD=0 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/NO_LOOKAHEAD/'
  LSQ_QR_CQR_PASS2=1 python tools/dense_bench.py $C 2>&1 | grep -v amdgpu.ids | sed 's/^/PASS2       /'
done > $O/ab.txt 2>&1
QRPROF_OUT=$O/prof_default bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_default.txt 2>&1
LSQ_QR_LOOKAHEAD=0 QRPROF_OUT=$O/prof_nola bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_nola.txt 2>&1
rm -rf $O/prof_default $O/prof_nola
for f in pytest_b pytest_a pytest_zz; do echo "== $f"; tail -n 5 $O/$f.log; done
awk '{k=$1" "$3" "$4; s[k]+=$5; n[k]++} END {for (k in s) printf "%-45s %.3f ms\n", k, s[k]/n[k]}' $O/ab.txt | sort
grep -v "^[EW]2026" $O/prof_default.txt | head -20; echo; grep -v "^[EW]2026" $O/prof_nola.txt | head -20
