#!/bin/bash
# fused step kernel vs two launches, alternating inside one gpurun call
for i in $(seq 1 ${1:-3}); do
for m in "" "--two-launch"; do
  timeout -k 10 120 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline $m 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('mode', '$m' or 'fused', 'step_ms', round(d['ms_per_step'],4), 'value', round(d['value']/1e6,2), 'kernel_ms', round(r['avg_kernel_ms'],4), 'frac', round(r['frac'],4))"
done; done
