#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_44; mkdir -p $O
for seed in 61 62 63; do timeout 900 python tools/dense_fuzz.py $seed 300 2>&1 | grep -v amdgpu.ids | tail -4; echo "seed $seed rc=$?"; done > $O/dense_fuzz.txt 2>&1
timeout 900 python tools/dense_fuzz_rank.py 64 150 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/dense_fuzz.txt
cat $O/dense_fuzz.txt
