#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_45; mkdir -p $O
{
for seed in 71 72; do
  timeout 1200 python tools/r6/qr_fuzz_wide.py $seed 40 /tmp/a_$seed.npz 2>&1 | grep -v amdgpu.ids | tail -5
  LSQ_QR_UPDATE_FLAT=0 LSQ_QR_LOOKAHEAD=0 timeout 1200 python tools/r6/qr_fuzz_wide.py $seed 40 /tmp/b_$seed.npz 2>&1 | grep -v amdgpu.ids | tail -5
  LSQ_QR_UPDATE_FLAT=0 timeout 1200 python tools/r6/qr_fuzz_wide.py $seed 40 /tmp/c_$seed.npz 2>&1 | grep -v amdgpu.ids | tail -5
  python3 - <<PY
import numpy as np
a, b, c = np.load("/tmp/a_$seed.npz"), np.load("/tmp/b_$seed.npz"), np.load("/tmp/c_$seed.npz")
same_c = sum(np.array_equal(a[k], c[k]) for k in a.files)
same_b = sum(np.array_equal(a[k], b[k]) for k in a.files)
worst = max(np.max(np.abs(a[k] - b[k])) / np.max(np.abs(a[k])) for k in a.files)
print("seed $seed: default vs round-5 update grid: %d of %d bit-identical; vs round-5 grid without look-ahead: %d bit-identical, worst rel diff %.2g" % (same_c, len(a.files), same_b, worst))
PY
done
} > $O/qr_fuzz_wide.txt 2>&1
cat $O/qr_fuzz_wide.txt
