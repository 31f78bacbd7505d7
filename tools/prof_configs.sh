#!/bin/bash
# Kernel traces of the other bench configurations (BASELINE.json configs[2] and configs[4] on one GPU; the per-frame constraints), and the
# HBM-traffic counters of the per-frame constraints' kernels (new route and the chain it replaces run in the same command).
# usage: tools/prof_configs.sh <tag>
set -o pipefail
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
# The library sets this itself when it is loaded -- but under rocprofv3 the profiler's own library has initialised the HIP runtime before
# python starts, and the runtime reads the variable then: without the export the profiled kernels fetch their arguments from host memory
# (+1.6 us on the frames kernel) and the trace is not the product's.
export HIP_FORCE_DEV_KERNARG=1
for cfg in graph optimizer frame_constraints; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$cfg
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline > $out/trace.log 2>&1 || { echo $cfg trace failed; tail -5 $out/trace.log; exit 1; }
  grep -h "^{\"metric\"" $out/trace.log | tail -1 > $out/bench.json
  find $out -name '*stats*.csv' | head -3
done
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_frame_constraints
B="$GRAFT_REPO_ROOT/bench.py --config frame_constraints --steps 200 --warmup 20 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1 || { echo write failed; tail -5 $out/pmc_write.log; exit 1; }
