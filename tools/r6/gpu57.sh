#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_57; mkdir -p $O
python tools/r6/bits_ab.py /tmp/new.npz 2>&1 | grep -v amdgpu.ids
LSQ_LIB_PATH=$PWD/tools/ab/before.so python tools/r6/bits_ab.py /tmp/old.npz 2>&1 | grep -v amdgpu.ids
python3 - <<'PY'
import numpy as np
a, b = np.load("/tmp/old.npz"), np.load("/tmp/new.npz")
for k in a.files: print(k, "bit-identical" if np.array_equal(a[k], b[k]) else "differs, max rel %.3g" % (np.max(np.abs(a[k]-b[k]))/np.max(np.abs(a[k]))))
PY
C="16384:2048:0 16384:2048:1 8192:1024:0 6000:2000:0 20000:1000:0 4096:512:0 4099:700:1"
for r in 1 2 3; do
  TAG=D2 python tools/r6/probe_seq.py $C
  TAG=OLD LSQ_LIB_PATH=$PWD/tools/ab/before.so python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_d2.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_57/ab_d2.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
timeout 900 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "qr or cholqr" 2>&1 | tail -2
