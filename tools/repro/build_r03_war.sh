#!/bin/bash
# Builds tools/repro/liblsqhip_r03war.so: TODAY's library (with the LSQ_DEBUG_* hooks) but with round 3's CholeskyQR2 panel
# (git c5845a2: k_cqr_top on the side stream stores S*R into A's panel triangle while k_cqr_pass<2> on the main stream may
# still read Q1's top rows from the same elements).  Used by tools/repro/qr_race.py to show that the round-3 GPU-suite
# failure (2049 x 129, QR, for_lm) is that write-after-read: the old panel fails under launch jitter, the new one does not.
# Needs the git history (run in the build container, not on the GPU box; the .so travels with the snapshot).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d /tmp/r03war.XXXX)
mkdir -p "$T/pkg" "$T/include"
cp -r "$ROOT/leastsquaresoptim.jl_amd/csrc" "$T/pkg/csrc"      # (csrc includes "../../include/lsqhip.h": same relative layout)
cp "$ROOT"/include/*.h "$T/include/"
rm -f "$T"/pkg/csrc/*.o
for f in lsq_qr_cholqr.hip lsq_qr_cholqr.h; do
  git -C "$ROOT" show c5845a2:leastsquaresoptim.jl_amd/csrc/$f | sed 's/hipLaunchKernelGGL(/LSQ_LAUNCH(/g' > "$T/pkg/csrc/$f"
done
make -s -C "$T/pkg/csrc" -j8 ../liblsqhip.so
cp "$T/pkg/liblsqhip.so" "$ROOT/tools/repro/liblsqhip_r03war.so"
rm -rf "$T"
echo built tools/repro/liblsqhip_r03war.so
