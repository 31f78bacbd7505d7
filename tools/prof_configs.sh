#!/bin/bash
# Kernel traces of the other two bench configurations (BASELINE.json configs[2] and configs[4] on one GPU).
# usage: tools/prof_configs.sh <tag>
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for cfg in graph optimizer; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$cfg
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline > $out/trace.log 2>&1 || { echo $cfg trace failed; tail -5 $out/trace.log; exit 1; }
  grep -h "^{\"metric\"" $out/trace.log | tail -1 > $out/bench.json
  find $out -name '*stats*.csv' | head -3
done
