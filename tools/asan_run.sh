#!/bin/bash
# The GPU tiers of the suite on the AddressSanitizer build (tools/asan_build.sh): host-side heap / stack / use-after-free errors of
# the C-ABI library abort the run with a report; leaks are not collected (the Python interpreter's own would drown them).
# usage (on the GPU box, from the repo root): tools/asan_run.sh [pytest arguments]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
RT=$(ls /usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/gcc/x86_64-linux-gnu/*/libasan.so 2>/dev/null | head -1)   # (gcc's: see asan_build.sh)
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export LSQ_LIB_PATH=$PWD/build/asan/liblsqhip.so
if [ "${1:-}" = "--canary" ]; then exec python tools/asan_canary.py; fi
exec python -m pytest "$@"
