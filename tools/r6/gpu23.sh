#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_24; mkdir -p $O
timeout 900 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "qr or cholqr or serial or partitioned" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log
C="16384:2048:0 16384:2048:1 20000:1000:0 4096:512:0 3000:700:1 8192:1024:0 40000:512:0 6000:2000:0 17000:1500:0 24000:2048:0 12000:2048:0"
for r in 1 2 3; do
  TAG=RULE python tools/r6/probe_seq.py $C
  TAG=LEGACY LSQ_QR_UPDATE_FLAT=0 python tools/r6/probe_seq.py $C
  TAG=FLAT1 LSQ_QR_UPDATE_FLAT=1 python tools/r6/probe_seq.py $C

  TAG=RULE_LA_OFF LSQ_QR_LOOKAHEAD=0 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_flat.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_24/ab_flat.txt"):
    m = re.match(r"(\S+) (\d+x\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(4).split()][1:]
        d[(m.group(2), m.group(3), m.group(1))] += v
for k in sorted(d):
    print(k, "median %.3f  min %.3f" % (sorted(d[k])[len(d[k]) // 2], min(d[k])))
PY
tail -n 3 $O/t_b.log
