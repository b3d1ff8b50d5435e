#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_41; mkdir -p $O
LSQ_LIB_PATH=$PWD/tools/ab/before_chol_la.so python tools/r6/bits_ab.py /tmp/old.npz 2>&1 | grep -v amdgpu.ids
python tools/r6/bits_ab.py /tmp/new.npz 2>&1 | grep -v amdgpu.ids
python3 - <<'PY'
import numpy as np
a, b = np.load("/tmp/old.npz"), np.load("/tmp/new.npz")
for k in a.files: print(k, "bit-identical" if np.array_equal(a[k], b[k]) else "DIFFERENT max rel %.3g" % (np.max(np.abs(a[k]-b[k]))/np.max(np.abs(a[k]))))
PY
timeout 900 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "chol or qr or serial" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log; tail -n 3 $O/t_b.log
timeout 300 python tools/chol_stress.py 3000 2>&1 | tail -1
for r in 1 2 3; do
  TAG=NEW python tools/r6/probe_seq.py 16384:2048:0 4096:512:0
  TAG=OLD LSQ_LIB_PATH=$PWD/tools/ab/before_chol_la.so python tools/r6/probe_seq.py 16384:2048:0 4096:512:0
done 2>&1 | grep -v amdgpu.ids | cut -c1-64
for r in 1 2; do
python tools/dense_bench.py chol:4096:512:1 2>&1 | grep Cholesky
LSQ_LIB_PATH=$PWD/tools/ab/before_chol_la.so python tools/dense_bench.py chol:4096:512:1 2>&1 | grep Cholesky
done
