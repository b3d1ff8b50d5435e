#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LSQ_CHOL_TRACE=1 timeout 300 python tools/chol_stress.py 70 2>&1 | grep -v amdgpu.ids | tail -40
