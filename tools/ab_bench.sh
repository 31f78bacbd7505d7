#!/bin/bash
# A/B two builds of libmg_hip.so inside one gpurun call (boxes differ by +-10% between calls).
# usage: tools/ab_bench.sh ab/lib_old.so ab/lib_new.so [rounds]
set -e
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    MG_HIP_LIB=$PWD/$L timeout -k 10 120 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', d['roofline'].get('avg_kernel_ms'), 'frac', round(d['roofline']['frac'],4))"
  done
done
