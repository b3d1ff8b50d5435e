#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in 4 6 7 8 9; do echo "== LSQ_QR_VTB_W=$m"; LSQ_QR_CLK=1 LSQ_QR_VTB_W=$m TAG=clk timeout 120 python tools/r6/probe_seq.py 16384:2048:0 2>&1 | grep "qr clk" | tail -3; done
