#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_15
./tools/micro/bin/mfma4_stream_bench 2>&1 | tee gpurun_out/r6_15/mfma4_stream.txt
