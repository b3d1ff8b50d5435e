#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_26; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" QRPROF_OUT=$O/p bash tools/qr_profile.sh chol:4096:512:1 > $O/p_$tag.txt 2>&1
  echo "== $tag: $(grep -h 'k_syrk_mfma\|k_syrk_reduce' $O/p_$tag.txt | awk '{print $(NF-3), $(NF-2)}' | tr '\n' ' ') | $(grep -h '^Cholesky' $O/p_$tag.txt | head -1)"
  rm -rf $O/p
}
run base LSQ_SYRK_PREFETCH=1 LSQ_SYRK_XCD_MAP=0
run pf2 LSQ_SYRK_PREFETCH=2 LSQ_SYRK_XCD_MAP=0
run pf3 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=0
run pf1_xcd LSQ_SYRK_PREFETCH=1 LSQ_SYRK_XCD_MAP=1
run pf3_xcd LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=1
run pf3_xcd_k7 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=1 LSQ_SYRK_KSLICES=7
run pf3_xcd_k8 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=1 LSQ_SYRK_KSLICES=8
run pf3_xcd_k16 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=1 LSQ_SYRK_KSLICES=16
run pf3_xcd_k21 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=1 LSQ_SYRK_KSLICES=21
run pf3_k7 LSQ_SYRK_PREFETCH=3 LSQ_SYRK_XCD_MAP=0 LSQ_SYRK_KSLICES=7
timeout 600 python -m pytest tests/test_b_gpu_kernels.py -q -x -k "chol" > $O/t_b.log 2>&1; echo "b rc=$?" >> $O/t_b.log; tail -n 3 $O/t_b.log
