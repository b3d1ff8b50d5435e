#!/bin/bash
# Build a VARIANT of liblsqhip.so for a same-box A/B: copies csrc to a scratch directory, applies the given sed script to it and
# links tools/ab/<name>.so (git-ignored, travels with gpurun).   tools/ab_build.sh <name> '<sed script>' [file ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SED=$2; shift 2
W=/tmp/ab_build_$NAME
rm -rf $W && mkdir -p $W/pkg/csrc $W/include $ROOT/tools/ab
cp $ROOT/leastsquaresoptim.jl_amd/csrc/*.hip $ROOT/leastsquaresoptim.jl_amd/csrc/*.h $ROOT/leastsquaresoptim.jl_amd/csrc/*.cpp $ROOT/leastsquaresoptim.jl_amd/csrc/Makefile $W/pkg/csrc/
cp $ROOT/include/*.h $W/include/
sed -i 's|../../include/lsqhip.h|../../include/lsqhip.h|' $W/pkg/csrc/Makefile
for f in ${@:-lsq_sell.h}; do sed -i -E "$SED" $W/pkg/csrc/$f; done
make -s -C $W/pkg/csrc -j4 ../liblsqhip.so
cp $W/pkg/liblsqhip.so $ROOT/tools/ab/$NAME.so
echo built tools/ab/$NAME.so
