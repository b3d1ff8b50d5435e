#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_16
./tools/micro/bin/mfma16_rule_bench 2>&1 | tee gpurun_out/r6_16/mfma16_rule.txt
