#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_25; mkdir -p $O
C="16384:2048:0 16400:2048:0 16416:2048:0 16448:2048:0 16512:2048:0 16640:2048:0 16128:2048:0 16320:2048:0 8192:1024:0 8208:1024:0 8256:1024:0 4096:512:0 4112:512:0 4160:512:0"
for r in 1 2 3; do
  TAG=LD python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_ld.txt
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r6_25/ab_ld.txt"):
    m = re.match(r"(\S+) (\d+)x(\d+) f=(\d): ([\d. ]+)", l)
    if m:
        v = [float(x) for x in m.group(5).split()][1:]
        d[(int(m.group(3)), int(m.group(2)))] += v
for k in sorted(d):
    md = sorted(d[k])[len(d[k]) // 2]
    fl = 2.0 * k[1] * k[0] ** 2 - 2.0 * k[0] ** 3 / 3
    print(k, "median %.3f  min %.3f   %.2f TF" % (md, min(d[k]), fl / md / 1e9))
PY
