#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_20; mkdir -p $O
C="16384:2048:0 16384:2048:1 20000:1000:0 4096:512:0 3000:700:1 8192:1024:0 40000:512:0 6000:2000:0"
for r in 1 2 3; do
  TAG=DEFAULT python tools/r6/probe_seq.py $C
  TAG=LA0 LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=0 python tools/r6/probe_seq.py $C
  TAG=LA0_FLDS LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=0 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
  TAG=LA128_FLDS LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=128 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_la2.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_20/ab_la2.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
