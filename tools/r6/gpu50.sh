#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_50; mkdir -p $O
timeout 900 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "qr or cholqr or serial or partitioned" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log
timeout 900 python -m pytest tests/test_zz_gpu_stress.py -q -x > $O/t_zz.log 2>&1; echo "zz rc=$?" >> $O/t_zz.log
timeout 600 python -m pytest tests/test_a_gpu_contract.py -q -x -k "c3 or minpack_trajectories or operator_level or golden" > $O/t_a.log 2>&1; echo "a rc=$?" >> $O/t_a.log
tail -n 3 $O/t_b.log $O/t_zz.log $O/t_a.log
for seed in 81; do
  timeout 1200 python tools/r6/qr_fuzz_wide.py $seed 40 /tmp/a_$seed.npz 2>&1 | grep -v amdgpu.ids | tail -3
  LSQ_QR_NO_SWIZZLE=1 timeout 1200 python tools/r6/qr_fuzz_wide.py $seed 40 /tmp/b_$seed.npz 2>&1 | grep -v amdgpu.ids | tail -3
  python3 - <<PY
import numpy as np
a, b = np.load("/tmp/a_$seed.npz"), np.load("/tmp/b_$seed.npz")
print("fragment-order operands vs matrix order: %d of %d bit-identical" % (sum(np.array_equal(a[k], b[k]) for k in a.files), len(a.files)))
PY
done
