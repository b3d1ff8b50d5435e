#!/bin/bash
# Host-side AddressSanitizer build of liblsqhip.so: the HOST code of every translation unit is
# instrumented (-fsanitize=address -fno-gpu-sanitize: device code as in the product build), into build/asan/.  Run the GPU suite
# on it with tools/asan_run.sh (LD_PRELOAD of the ASan runtime, LSQ_LIB_PATH pointing at this build).
set -eu
cd "$(dirname "$0")/../leastsquaresoptim.jl_amd/csrc"
OUT=../../build/asan
mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -ffp-contract=off -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in lsq_vec lsq_sparse lsq_lsmr lsq_lsmr_general lsq_dense lsq_qr lsq_qr_stage1 lsq_optimize lsq_exact lsq_dense_mfma lsq_qr_cholqr; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -c lsq_synth.cpp -o $OUT/lsq_synth.o
# linked WITHOUT a sanitizer runtime: the __asan_* symbols come from the runtime that tools/asan_run.sh preloads -- gcc's libasan,
# because ROCm's own (libclang_rt.asan) intercepts hsa_amd_memory_pool_allocate and aborts under the non-ASan HIP runtime of this
# image ("out-of-memory" in the interceptor; SEGV with allocator_may_return_null=1), measured on the GPU box
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/liblsqhip.so $OUT/*.o -ldl
ls -la $OUT/liblsqhip.so
