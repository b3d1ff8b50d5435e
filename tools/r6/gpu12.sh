#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_12
./tools/micro/bin/mfma4_equiv 2>&1 | tee gpurun_out/r6_12/mfma4_equiv.txt
