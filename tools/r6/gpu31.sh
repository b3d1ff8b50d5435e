#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_31; mkdir -p $O
C="16384:2048:0 16384:2048:1 8192:1024:0 4096:512:0 20000:1000:0 6000:2000:0"
for r in 1 2 3; do
  for g in 3 4 5 6; do TAG=RING$g LSQ_QR_VTB_RING=$g python tools/r6/probe_seq.py $C; done
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_ring.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_31/ab_ring.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
LSQ_QR_VTB_RING=5 timeout 600 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "wave_private" 2>&1 | tail -2
