#!/bin/bash
# chunk geometry of the frames kernel: tools/chunk_sweep.sh "8 7 6 5" [rounds]   (MG_CHUNK_W values)
for i in $(seq 1 ${2:-2}); do
for w in $1; do
  MG_CHUNK_W=$w timeout -k 10 120 python3 bench.py --steps ${STEPS:-1000} --warmup 50 --no-cpu-baseline 2>/dev/null \
   | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window', $w, 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline'].get('avg_kernel_ms'),4), 'frac', round(d['roofline']['frac'],4))"
done; done
