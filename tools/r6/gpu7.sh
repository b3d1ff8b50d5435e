#!/bin/bash
# round 6, call 7: group-level Gram sums (hier): tests + A/B + trace; k_combine prefetch + tail-oracle A/B on the bench
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_7; mkdir -p $O
( timeout 900 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "qr or cholqr or serial or partitioned or lsmr or combine or sparse_products" 2>&1 | tail -n 8 ) > $O/pytest_b.log
( timeout 600 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c3 or c4 or minpack_trajectories or operator_level or golden or three_launch" 2>&1 | tail -n 12 ) > $O/pytest_a.log
( timeout 300 python -m pytest tests/test_zz_gpu_stress.py -x -q 2>&1 | tail -n 8 ) > $O/pytest_zz.log
C="16384:2048:0 16384:2048:1 4096:512:0 3000:700:1 20000:1000:0 40000:512:0"
for r in 1 2 3; do
  TAG=DEFAULT python tools/r6/probe_seq.py $C
  TAG=NO_HIER LSQ_QR_NO_HIER=1 python tools/r6/probe_seq.py $C
  TAG=PASS2 LSQ_QR_CQR_PASS2=1 python tools/r6/probe_seq.py $C
done 2>&1 | grep -v amdgpu.ids > $O/ab_hier.txt
QRPROF_OUT=$O/prof_default bash tools/qr_profile.sh qr:16384:2048:0 > $O/prof_default.txt 2>&1
rm -rf $O/prof_default
for r in 1 2 3; do
  for mode in default LSQ_TAIL_ORACLE_SEQ=1,1,6,5,3,1 LSQ_NO_TAIL_SPECULATION=1; do
    if [ "$mode" = default ]; then pre=""; else pre="env $mode"; fi
    $pre python bench.py --no-cpu --no-dense 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('%-40s round $r: %.1f it/s (min %.1f max %.1f)  Jv %.2f us  Jtu %.2f us  tail %s' % ('$mode', j['value'], j['value_min'], j['value_max'], r['avg_launch_ms']*1e3, r['jtu_kernel_avg_ms']*1e3, j['tail_speculation']['timed_regions']))"
  done
done > $O/ab_tail_oracle.txt 2>&1
for f in pytest_b pytest_a pytest_zz; do echo "== $f"; tail -n 5 $O/$f.log; done
cat $O/ab_hier.txt | cut -c1-90; grep -v "^[EW]2026" $O/prof_default.txt | head -20; cat $O/ab_tail_oracle.txt
