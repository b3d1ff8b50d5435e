#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_10
./tools/micro/bin/mfma64_4x4_bench 2>&1 | tee gpurun_out/r6_10/mfma64_4x4.txt
