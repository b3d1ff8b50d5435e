#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_49; mkdir -p $O
C="16384:2048:0 16384:2048:1 20000:1000:0 4096:512:0 8192:1024:0 6000:2000:0 3000:700:1"
for r in 1 2 3; do
  TAG=H2 LSQ_QR_HIER2=1 python tools/r6/probe_seq.py $C
  TAG=BASE LSQ_QR_HIER2=0 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_h2.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_49/ab_h2.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
LSQ_QR_HIER2=1 timeout 600 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "wave_private or cholqr" 2>&1 | tail -2
