#!/bin/bash
# Developer aid: builds csrc/libgcnpt_<name>.so from the current sources with extra compile-time switches, in a scratch copy
# (for tools/ab_libs.sh).   usage: tools/build_variant.sh <name> "-DGCNPT_W_STAGGER=4 ..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2
S=/tmp/gcnpt_variant_$name
rm -rf $S && mkdir -p $S/pkg $S/include
cp -r "$R/gcn-over-pruned-trees_amd/csrc" $S/pkg/csrc && cp "$R/include/gcnpt.h" $S/include/
cd $S/pkg/csrc && rm -rf *.o *.so _stamps
make -s -j8 libgcnpt.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $flags"
cp libgcnpt.so "$R/gcn-over-pruned-trees_amd/csrc/libgcnpt_$name.so"
rm -rf $S
