#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_54; mkdir -p $O
C="16384:2048:0 16384:2048:1 8192:1024:0 6000:2000:0"
for r in 1 2 3; do
  TAG=BASE python tools/r6/probe_seq.py $C
  TAG=VTB2 LSQ_QR_VTB_WGS=2 python tools/r6/probe_seq.py $C
  TAG=FLAT2 LSQ_QR_UPDATE_FLAT=2 python tools/r6/probe_seq.py $C
  TAG=FLAT1 LSQ_QR_UPDATE_FLAT=1 python tools/r6/probe_seq.py $C
  TAG=LA1 LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=0 python tools/r6/probe_seq.py $C
  TAG=LA1_1024 LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=1024 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_retune.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_54/ab_retune.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
