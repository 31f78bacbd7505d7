#!/bin/bash
# Which LDS access class conflicts (VERDICT r4 "What's weak" 2 ii)?  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the fused step
# with one access class removed at a time (the diagnostic build's MG_DEBUG_FLAGS) and with the sweep's lane map of round 3.
#   gpurun -- bash tools/probes/lds_conflicts.sh        -> gpurun_out/r05_lds_conflicts.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_lds_conflicts.txt
: > $O
cd /tmp && export TMPDIR=/tmp
run() {   # name lib flags
  d=/tmp/ldsc_$(echo "$1" | tr -c 'a-zA-Z0-9' '_'); rm -rf $d
  MG_DEBUG_FLAGS=$3 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d "$d" -- python3 $R/tools/probes/run_step.py $2 200 > "$d.log" 2>&1 || { echo "$1 failed"; tail -n 3 "$d.log"; return; }
  python3 - "$1" "$d" >> $O <<'PY'
import csv, glob, sys
from collections import defaultdict
name, d = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frames_cs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
print("%-34s conflict cycles %10.0f  idx active %11.0f  = %5.1f %%   LDS instructions %9.0f  addr conflicts %s" % (
    name, c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
    c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_ADDR_CONFLICT", "-")))
PY
}
DBG=$R/morphablegraphs_amd/csrc/libmg_hip_dbg.so
run "product build" $R/morphablegraphs_amd/csrc/libmg_hip.so 0
run "diagnostic build, nothing removed" $DBG 16
run "no tap reads (sweep)" $DBG $((16+16384))
run "tap reads, no FMAs" $DBG $((16+8192))
run "stores only (no taps, no FMAs)" $DBG $((16+4))
run "no row production" $DBG $((16+1))
run "no root stage" $DBG $((16+1024))
[ -f $R/build/lib_lanemap0.so ] && run "third sample in lanes 40..59 (r3 map)" $R/build/lib_lanemap0.so 0
[ -f $R/build/lib_ropad0.so ] && run "root outputs unpadded (MG_RO_PAD 0)" $R/build/lib_ropad0.so 0
cat $O
