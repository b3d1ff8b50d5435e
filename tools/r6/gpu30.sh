#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_30; mkdir -p $O
timeout 120 bash tools/asan_run.sh --canary > $O/asan_canary.log 2>&1; echo "rc=$?" >> $O/asan_canary.log
timeout 2400 bash tools/asan_run.sh tests/test_b_gpu_kernels.py -m gpu -q -x > $O/asan_tier_b.log 2>&1; echo "rc=$?" >> $O/asan_tier_b.log
timeout 2400 bash tools/asan_run.sh tests/test_a_gpu_contract.py tests/test_zz_gpu_stress.py -m gpu -q -x > $O/asan_tier_a_zz.log 2>&1; echo "rc=$?" >> $O/asan_tier_a_zz.log
timeout 1200 bash tools/asan_run.sh tests/test_rowshard.py tests/test_sharding.py -m gpu -q -x > $O/asan_tier_shard.log 2>&1; echo "rc=$?" >> $O/asan_tier_shard.log
tail -n 4 $O/asan_canary.log $O/asan_tier_b.log $O/asan_tier_a_zz.log $O/asan_tier_shard.log
