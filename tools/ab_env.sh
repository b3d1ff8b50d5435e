#!/bin/bash
# Same-box A/B of environment switches: alternates `python bench.py --no-cpu` with and without the given VAR=VALUE, 3 rounds.
#   tools/ab_env.sh LSQ_NO_TAIL_SPECULATION=1     (run through gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
  for mode in default "$1"; do
    if [ "$mode" = default ]; then pre=""; else pre="env $mode"; fi
    $pre python bench.py --no-cpu ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('%-32s round $round: %.1f it/s (min %.1f max %.1f)  Jv %.2f us  Jtu %.2f us  tail %s' % ('$mode', j['value'], j['value_min'], j['value_max'], r['avg_launch_ms']*1e3, r['jtu_kernel_avg_ms']*1e3, j.get('tail_speculation')))"
  done
done
