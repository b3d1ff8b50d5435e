#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_5; mkdir -p $O
C="16384:2048:0 16384:2048:1 4096:512:0 3000:700:1 20000:1000:0"
for r in 1 2 3 4; do
  TAG=DEFAULT python tools/r6/probe_seq.py $C
  TAG=PASS2 LSQ_QR_CQR_PASS2=1 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | tee $O/probe_seq.txt
