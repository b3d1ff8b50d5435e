#!/bin/bash
# round 6, call 9: full GPU suite on the tree with the round's defaults (Q1 form, fused Gram, Neumann top, react tail), then the
# round's profile collection (bench + rocprofv3 kernel trace + PMC passes) and the dense kernel summaries
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_9; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 ) > $O/gputest_full.log
git rev-parse --short HEAD >> $O/gputest_full.log 2>/dev/null
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
bash tools/make_dense_profiles.sh > $O/make_dense_profiles.log 2>&1
cat $O/gputest_full.log; tail -c 1500 gpurun_out/prof/bench.json; ls gpurun_out/prof gpurun_out/denseprof
