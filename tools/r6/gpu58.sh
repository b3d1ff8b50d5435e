#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_58; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputest_full.log 2>&1; echo "rc=$?" >> $O/gputest_full.log
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
bash tools/make_dense_profiles.sh > $O/make_dense.log 2>&1
tail -n 4 $O/gputest_full.log; tail -c 400 gpurun_out/prof/bench.json
