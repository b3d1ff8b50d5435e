#!/bin/bash
# round 6, call 14: where the 4x4x4 form of the trailing kernels loses: MFMA side alone / memory side alone, both instruction shapes
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_14; mkdir -p $O
prof() {  # tag, env...
  local tag=$1; shift
  env "$@" QRPROF_OUT=$O/p_$tag bash tools/qr_profile.sh qr:16384:2048:0 > $O/p_$tag.txt 2>&1
  rm -rf $O/p_$tag
  echo "== $tag"; grep -E "k_qr1_update_w|k_qr1_vtb_w" $O/p_$tag.txt | cut -c1-40,70-130; tail -n 1 $O/p_$tag.txt | cut -c1-60
}
prof m4_full X=1
prof m4_upd_mem LSQ_QR_UPDATE_W=2
prof m4_upd_mfma LSQ_QR_UPDATE_W=3
prof m4_vtb_nomfma LSQ_QR_VTB_W=2
L=$PWD/tools/ab/mfma16.so
prof m16_full LSQ_LIB_PATH=$L
prof m16_upd_mfma LSQ_LIB_PATH=$L LSQ_QR_UPDATE_W=3
prof m16_vtb_nomfma LSQ_LIB_PATH=$L LSQ_QR_VTB_W=2
