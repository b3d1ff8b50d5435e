#!/bin/bash
# The GPU suite's contract / kernel / sharding tiers run twice more with the library's debug modes switched on from the
# environment (include/lsqhip.h: lsq_debug_set): under launch jitter every oracle comparison must still hold (a hand-off
# between kernels that depends on queue timing would break one), and serialised (every launch waited for) likewise.
# Run through gpurun from the repo root; the record goes to gpurun_out/<dir>/suite_under_debug_modes.txt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/debug_modes}
mkdir -p $OUT
{
  echo "== LSQ_DEBUG_LAUNCH_JITTER=100: contract + kernel + sharding tiers"
  LSQ_DEBUG_LAUNCH_JITTER=100 timeout 1500 python -m pytest tests/test_a_gpu_contract.py tests/test_b_gpu_kernels.py tests/test_rowshard.py tests/test_sharding.py -m gpu -q 2>&1 | tail -15
  echo "== LSQ_DEBUG_SERIAL=1: contract + kernel tiers"
  LSQ_DEBUG_SERIAL=1 timeout 1500 python -m pytest tests/test_a_gpu_contract.py tests/test_b_gpu_kernels.py -m gpu -q 2>&1 | tail -15
} > $OUT/suite_under_debug_modes.txt 2>&1
cat $OUT/suite_under_debug_modes.txt
