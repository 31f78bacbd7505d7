#!/bin/bash
# Build a variant of libmg_hip.so for tools/ab.py:   tools/build_variant.sh NAME [-DMACRO=VALUE ...]   ->  build/lib_NAME.so
# FILE=mg_options.hip tools/build_variant.sh ... recompiles that translation unit with the extra flags (default: mg_frames_cs.hip
# with -DMG_ONLY_KK10: just the 'walk' instantiations, to keep it quick).
set -e
cd "$(dirname "$0")/../morphablegraphs_amd/csrc"
name=$1; shift
file=${FILE:-mg_frames_cs.hip}
make -s libmg_hip.so
mkdir -p ../../build
extra=""; case "$file" in mg_frames_cs.hip|mg_frames_ws.hip) extra=-DMG_ONLY_KK10;; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result $extra "$@" \
    -c -o ../../build/v_$name.o $file
objs=""
for f in mg_host mg_frames mg_frames_cs mg_frames_ws mg_frames_direct mg_gmm mg_score mg_placement mg_trajectory mg_options mg_frame_constraints mg_timewarp; do
    if [ "$f.hip" = "$file" ]; then objs="$objs ../../build/v_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$name.so $objs
echo build/lib_$name.so
