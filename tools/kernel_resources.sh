#!/bin/bash
# Registers, spills, scratch and LDS of every kernel in an object file or the library (code-object metadata):
#   tools/kernel_resources.sh morphablegraphs_amd/csrc/mg_trajectory.o [name filter]
set -e
OBJ=${1:-morphablegraphs_amd/csrc/libmg_hip.so}
FILTER=${2:-.}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
LLVM=/opt/rocm/lib/llvm/bin
$LLVM/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$OBJ"
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
$LLVM/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
flt = re.compile(sys.argv[1])
for blk in txt.split("  - .agpr_count:")[1:]:
    def get(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = get("name")
    if not flt.search(name): continue
    print("%-110s vgpr %4s agpr %4s sgpr %4s spill %4s scratch %6s lds %7s" % (name[:110], get("vgpr_count"), blk.split()[0], get("sgpr_count"), get("vgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
' "$FILTER"
