#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_28; mkdir -p $O
C="16384:2048:1 20000:1000:0 40000:512:0 24000:2048:0 17000:1500:0"
for r in 1 2 3; do
  TAG=BASE python tools/r6/probe_seq.py $C
  TAG=SPARE2 LSQ_QR_UPDATE_SPARE=2 python tools/r6/probe_seq.py $C
  TAG=SPARE2_FLDS LSQ_QR_UPDATE_SPARE=2 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
  TAG=SPARE4_FLDS LSQ_QR_UPDATE_SPARE=4 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
  TAG=REDUNDANT LSQ_QR_AHEAD_REDUNDANT=1 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_spare.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_28/ab_spare.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
