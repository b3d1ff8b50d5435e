#!/bin/bash
# round 6, call 6: full GPU suite + smoke + bench on the current tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_6; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 ) > $O/gputest_full.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 ) > $O/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err )
cat $O/gputest_full.log $O/smoke.log
python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/r6_6/bench.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step","parity_ok","value_fixed8_schedule","value_generic_g_device","c2_ldiv_ms","c3_ldiv_ms","c3_frac","c3_dogleg_qr_outer_ms","tail_wrong_frac","wide_jv_frac","wide_jtu_frac")})
    r=j["roofline"]; print(r["frac"], r["avg_launch_ms"], r["jtu_frac"], r["jtu_kernel_avg_ms"], r.get("physical_GBps"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r6_6/bench.err").read()[-3000:])
PY
