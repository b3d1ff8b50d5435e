#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_19; mkdir -p $O
C="16384:2048:0 16384:2048:1 20000:1000:0"
for r in 1 2 3; do
  TAG=DEFAULT python tools/r6/probe_seq.py $C
  TAG=LA1024 LSQ_QR_LOOKAHEAD=1 python tools/r6/probe_seq.py $C
  TAG=LA1024_FLDS LSQ_QR_LOOKAHEAD=1 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
  TAG=LA512_FLDS LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=512 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
  TAG=LA0_FLDS LSQ_QR_LOOKAHEAD=1 LSQ_QR_LOOKAHEAD_MINCOLS=0 LSQ_QR_FACTOR_LDS=102400 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $O/ab_la.txt
LSQ_QR_LOOKAHEAD=1 LSQ_QR_FACTOR_LDS=102400 QRPROF_OUT=$O/prof bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_la_flds.txt 2>&1; rm -rf $O/prof
sort $O/ab_la.txt; grep -v "^[EW]2026" $O/prof_la_flds.txt | head -14 | cut -c1-40,70-130
