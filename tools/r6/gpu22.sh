#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_22; mkdir -p $O
timeout 600 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "wave_private" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log
export LSQ_QR_LOOKAHEAD=0
QRPROF_OUT=gpurun_out/r6_22/p16384 bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_16384.txt 2>&1
QRPROF_OUT=gpurun_out/r6_22/p18432 bash tools/qr_profile.sh qr:18432:2048:0 > $O/prof_18432.txt 2>&1
QRPROF_OUT=gpurun_out/r6_22/p17000 bash tools/qr_profile.sh qr:17000:2048:0 > $O/prof_17000.txt 2>&1
rm -rf $O/p16384 $O/p18432 $O/p17000
tail -n 3 $O/t_b.log; cat $O/prof_16384.txt $O/prof_17000.txt $O/prof_18432.txt
