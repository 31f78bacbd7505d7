#!/bin/bash
# sweep MG_DEBUG_FLAGS values on the current library: tools/flag_sweep.sh "0 64 128" [rounds]
for i in $(seq 1 ${2:-2}); do
for f in $1; do
  MG_DEBUG_FLAGS=$f timeout -k 10 120 python3 bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags', $f, 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline'].get('avg_kernel_ms'),4), 'frac', round(d['roofline']['frac'],4))"
done; done
