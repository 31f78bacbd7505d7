#!/bin/bash
# Build a variant of libmg_hip.so for tools/ab.py:   tools/build_variant.sh NAME [-DMACRO=VALUE ...]   ->  build/lib_NAME.so
# Only mg_backproject.hip is recompiled with the extra flags (-DMG_ONLY_KK10: just the 'walk' instantiations, to keep it quick).
set -e
cd "$(dirname "$0")/../morphablegraphs_amd/csrc"
name=$1; shift
make -s libmg_hip.so
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result -DMG_ONLY_KK10 "$@" \
    -c -o ../../build/bp_$name.o mg_backproject.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$name.so mg_host.o ../../build/bp_$name.o mg_gmm.o mg_score.o mg_placement.o mg_trajectory.o
echo build/lib_$name.so
