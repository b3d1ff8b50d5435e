#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_51; mkdir -p $O
python tools/r6/bits_ab.py /tmp/new.npz 2>&1 | grep -v amdgpu.ids
LSQ_LIB_PATH=$PWD/tools/ab/before.so python tools/r6/bits_ab.py /tmp/old.npz 2>&1 | grep -v amdgpu.ids
python3 - <<'PY'
import numpy as np
a, b = np.load("/tmp/old.npz"), np.load("/tmp/new.npz")
for k in a.files: print(k, "bit-identical" if np.array_equal(a[k], b[k]) else "DIFFERENT max rel %.3g" % (np.max(np.abs(a[k]-b[k]))/np.max(np.abs(a[k]))))
PY
for v in new old; do
  if [ $v = old ]; then export LSQ_LIB_PATH=$PWD/tools/ab/before.so; else unset LSQ_LIB_PATH; fi
  QRPROF_OUT=$O/p bash tools/qr_profile.sh chol:4096:512:1 2>&1 | grep "k_syrk_mfma\|k_chol_chain\|^Cholesky"
  QRPROF_OUT=$O/p bash tools/qr_profile.sh chol:16384:2048:1 2>&1 | grep "k_syrk_mfma\|^Cholesky"
done
unset LSQ_LIB_PATH
timeout 600 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "chol or syrk or dense" 2>&1 | tail -2
