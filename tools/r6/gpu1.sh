#!/bin/bash
# round 6, call 1: Q1-form panel (no pass 2) A/B + targeted parity tests + the new bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_1; mkdir -p $O
( timeout 900 python -m pytest tests/test_b_gpu_kernels.py -x -q -k "qr or cholqr or lsmr or serial or partitioned" 2>&1 | tail -15 ) > $O/pytest_b.log
( timeout 600 python -m pytest tests/test_a_gpu_contract.py -x -q -k "c3 or three_launch or minpack_trajectories or operator_level" 2>&1 | tail -15 ) > $O/pytest_a.log
( timeout 300 python -m pytest tests/test_zz_gpu_stress.py -x -q 2>&1 | tail -15 ) > $O/pytest_zz.log
for r in 1 2 3; do
  python tools/dense_bench.py qr:16384:2048:0 qr:16384:2048:1 qr:4096:512:0 qr:3000:700:1 2>&1 | grep -v amdgpu.ids
  LSQ_QR_CQR_PASS2=1 python tools/dense_bench.py qr:16384:2048:0 qr:16384:2048:1 qr:4096:512:0 qr:3000:700:1 2>&1 | grep -v amdgpu.ids | sed 's/^/PASS2 /'
  LSQ_QR_LOOKAHEAD=0 python tools/dense_bench.py qr:16384:2048:0 qr:16384:2048:1 2>&1 | grep -v amdgpu.ids | sed 's/^/NOLA /'
done > $O/ab_q1form.txt 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err )
tail -5 $O/pytest_b.log $O/pytest_a.log $O/pytest_zz.log; cat $O/ab_q1form.txt; python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/r6_1/bench.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step","parity_ok","value_fixed8_schedule","value_generic_g_device","c2_ldiv_ms","c3_ldiv_ms","tail_wrong_frac","lsmr_inner_iteration_us")})
    print(j["roofline"]["frac"], j["roofline"]["jtu_frac"], j["cpu_baseline"]["value"], j["cpu_baseline"].get("cpu_all_cores_value"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r6_1/bench.err").read()[-3000:])
PY
