#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_4; mkdir -p $O
for r in 1 2 3; do
  TAG=DEFAULT python tools/r6/probe_outlier.py 3000 700 1 40
  TAG=NO_GRAM LSQ_QR_NO_FUSED_GRAM=1 python tools/r6/probe_outlier.py 3000 700 1 40
  TAG=NO_VTB_LDS LSQ_QR_VTB_LDS=0 python tools/r6/probe_outlier.py 3000 700 1 40
  TAG=PASS2 LSQ_QR_CQR_PASS2=1 python tools/r6/probe_outlier.py 3000 700 1 40
  TAG=PASS2_R5 LSQ_QR_CQR_PASS2=1 LSQ_QR_VTB_LDS=0 LSQ_QR_NO_FUSED_GRAM=1 python tools/r6/probe_outlier.py 3000 700 1 40
  TAG=NOCOOP LSQ_QR1_COOP=0 python tools/r6/probe_outlier.py 3000 700 1 40
done 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
