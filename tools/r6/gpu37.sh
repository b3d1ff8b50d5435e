#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in 4 6 2 1; do echo "== LSQ_QR_VTB_W=$m"; LSQ_QR_CLK=1 LSQ_QR_VTB_W=$m TAG=clk python tools/r6/probe_seq.py 16384:2048:0 2>&1 | grep "qr clk\|clk 16384" | tail -4; done
