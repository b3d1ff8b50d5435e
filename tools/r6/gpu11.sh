#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_11
./tools/micro/bin/mfma4_probe > gpurun_out/r6_11/mfma4_probe.txt 2>&1
head -70 gpurun_out/r6_11/mfma4_probe.txt
