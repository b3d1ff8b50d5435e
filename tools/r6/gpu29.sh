#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_29; mkdir -p $O
timeout 900 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "qr or cholqr or serial or partitioned" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log
timeout 900 python -m pytest tests/test_zz_gpu_stress.py -q -x > $O/t_zz.log 2>&1; echo "zz rc=$?" >> $O/t_zz.log
timeout 600 python -m pytest tests/test_a_gpu_contract.py -q -x -k "c3" > $O/t_a.log 2>&1; echo "a rc=$?" >> $O/t_a.log
tail -n 3 $O/t_b.log $O/t_zz.log $O/t_a.log
