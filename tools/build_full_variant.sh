#!/bin/bash
# Build a whole-library variant for tools/ab.py:   tools/build_full_variant.sh NAME [-DMACRO=VALUE ...]   ->  build/lib_NAME.so
# Every translation unit is recompiled with the extra flags (macros that several of them read), the frames kernels with
# -DMG_ONLY_KK10 (just the 'walk' instantiations) to keep it quick.  Objects go to build/NAME/.
set -e
cd "$(dirname "$0")/../morphablegraphs_amd/csrc"
name=$1; shift
mkdir -p ../../build/$name
pids=""
for f in mg_host mg_frames mg_frames_cs mg_frames_ws mg_frames_direct mg_gmm mg_score mg_placement mg_trajectory mg_options mg_frame_constraints mg_timewarp; do
    extra=""; case "$f" in mg_frames_cs|mg_frames_ws) extra=-DMG_ONLY_KK10;; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result $extra "$@" \
        -c -o ../../build/$name/$f.o $f.hip &
    pids="$pids $!"
    if [ $(echo $pids | wc -w) -ge 6 ]; then wait $pids; pids=""; fi
done
wait $pids
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/lib_$name.so ../../build/$name/*.o
echo build/lib_$name.so
