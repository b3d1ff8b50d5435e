#!/bin/bash
# Same-box A/B of library builds: alternates `python bench.py --no-cpu` between the given .so files (LSQ_LIB_PATH), 3 rounds,
# prints value / J*v / J'u kernel times per run.   tools/ab_bench.sh tools/ab/A.so tools/ab/B.so ...   (run through gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
  for lib in "$@"; do
    LSQ_LIB_PATH=$PWD/$lib python bench.py --no-cpu ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('$lib round $round: %.1f it/s (min %.1f max %.1f)  Jv %.2f us  Jtu %.2f us  gen Jv %.2f  gen Jtu %.2f' % (j['value'], j['value_min'], j['value_max'], r['avg_launch_ms']*1e3, r['jtu_kernel_avg_ms']*1e3, r['generic_jv_ms']*1e3, r['generic_jtu_ms']*1e3))"
  done
done
