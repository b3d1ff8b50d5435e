#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_48; mkdir -p $O
QRPROF_OUT=$O/p bash tools/qr_profile.sh qr:16384:2048:0 2>&1 | grep -v "rocprofv3\|output_stream" | head -9
LSQ_QR_NO_SWIZZLE=1 QRPROF_OUT=$O/p bash tools/qr_profile.sh qr:16384:2048:0 2>&1 | grep -v "rocprofv3\|output_stream" | head -9
rm -rf $O/p
