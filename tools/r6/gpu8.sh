#!/bin/bash
# round 6, call 8: react-instead-of-preplace tail policy: correctness + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_8; mkdir -p $O
( LSQ_TAIL_REACT=1 timeout 900 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c4 or tanh or three_launch or pair_tail or speculative or column_scaled or device_g" 2>&1 | tail -n 8 ) > $O/pytest_a_react.log
( LSQ_TAIL_REACT=1 timeout 600 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "serial_mode" 2>&1 | tail -n 6 ) > $O/pytest_b_react.log
for r in 1 2 3; do
  for mode in default LSQ_TAIL_REACT=1 LSQ_TAIL_ORACLE_SEQ=1,1,6,5,3,1; do
    if [ "$mode" = default ]; then pre=""; else pre="env $mode"; fi
    $pre python bench.py --no-cpu --no-dense 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('%-40s round $r: %.1f it/s (min %.1f max %.1f)  fixed8 %.1f  Jv %.2f us  Jtu %.2f us  tail %s' % ('$mode', j['value'], j['value_min'], j['value_max'], j['value_fixed8_schedule'], r['avg_launch_ms']*1e3, r['jtu_kernel_avg_ms']*1e3, j['tail_speculation']['timed_regions']))"
  done
done > $O/ab_tail_react.txt 2>&1
TAG=DEFAULT python tools/r6/probe_seq.py 16384:2048:0 16384:2048:1 2>&1 | grep -v amdgpu.ids | cut -c1-80
for f in pytest_a_react pytest_b_react; do echo "== $f"; tail -n 5 $O/$f.log; done
cat $O/ab_tail_react.txt
