#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_18; mkdir -p $O
( timeout 900 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "qr or cholqr or serial or partitioned" 2>&1 | tail -n 8 ) > $O/pytest_b.log
( timeout 600 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c3 or minpack_trajectories or operator_level or golden" 2>&1 | tail -n 12 ) > $O/pytest_a.log
( timeout 300 python -m pytest tests/test_zz_gpu_stress.py -x -q 2>&1 | tail -n 8 ) > $O/pytest_zz.log
C="16384:2048:0 16384:2048:1 4096:512:0 3000:700:1 20000:1000:0"
for r in 1 2 3; do TAG=TRSM python tools/r6/probe_seq.py $C; done 2>&1 | grep -v amdgpu.ids | cut -c1-70 > $O/timing.txt
QRPROF_OUT=$O/prof bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof.txt 2>&1; rm -rf $O/prof
for f in pytest_b pytest_a pytest_zz; do echo "== $f"; tail -n 4 $O/$f.log; done
cat $O/timing.txt; grep -v "^[EW]2026" $O/prof.txt | head -9 | cut -c1-40,70-130
