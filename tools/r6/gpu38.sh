#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LSQ_QR_CLK=1 LSQ_QR_VTB_W=6 TAG=clk timeout 120 python tools/r6/probe_seq.py 16384:2048:0 2>&1 | grep -v amdgpu.ids | tail -15
