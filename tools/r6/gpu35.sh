#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_35; mkdir -p $O
for cfg in "4 1" "4 2"; do
  set -- $cfg
  LSQ_QR_VTB_W=$1 LSQ_QR_VTB_WGS=$2 QRPROF_OUT=$O/p bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_$1_$2.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/p/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('void k_qr1_vtb_w')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows][-32:]
print("VTB_W=$1 WGS=$2 grid", rows[-32]['Grid_Size_X'] if 'Grid_Size_X' in rows[0] else '?', "per panel us:", ' '.join('%.0f'%x for x in d), ' sum %.0f' % sum(d))
PY
  rm -rf $O/p
done
