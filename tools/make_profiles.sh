#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repo root):
#   bench.json, bench_under_rocprof.json, kernel trace + --stats CSVs, two PMC passes.
# Results land in gpurun_out/prof/; tools/publish_profiles.py turns them into profiles/<round>/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench --output-format csv -- python bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python bench.py --no-cpu --steps 16 --warmup 8 > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python bench.py --no-cpu --steps 16 --warmup 8 > /dev/null 2> $OUT/pmc_write.err
# calibration of the FETCH_SIZE correction on a kernel of KNOWN traffic: the multiplied-out g! (k_scale_lds: reads 10 B, writes
# 8 B per stored entry) only runs with LSQ_NO_COLSCALE=1 since round 3, so it gets two short passes of its own
LSQ_NO_COLSCALE=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_cal -o f --output-format csv -- python bench.py --no-cpu --steps 8 --warmup 0 --repeats 2 > /dev/null 2> $OUT/pmc_fetch_cal.err
LSQ_NO_COLSCALE=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_cal -o w --output-format csv -- python bench.py --no-cpu --steps 8 --warmup 0 --repeats 2 > /dev/null 2> $OUT/pmc_write_cal.err
find $OUT -name "*.csv" | head -20
tail -c 600 $OUT/bench.json
