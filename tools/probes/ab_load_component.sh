# Before / after builds (build/lib_lc0.so, build/lib_lc1.so) in alternating processes: the optimiser's iteration, the planner step, the two-launch step
cd $GRAFT_REPO_ROOT
L=morphablegraphs_amd/csrc/libmg_hip.so
cp $L /tmp/lib_keep.so
for v in lc0 lc1 lc0 lc1; do
  cp build/lib_$v.so $L
  python bench.py --config optimizer --no-cpu-baseline > /tmp/o.json 2>/dev/null
  python bench.py --config graph --no-cpu-baseline > /tmp/g.json 2>/dev/null
  python bench.py --two-launch --no-cpu-baseline --single-window --no-extra-configs --no-placement-compare > /tmp/t.json 2>/dev/null
  python -c "
import json
o=json.load(open('/tmp/o.json')); g=json.load(open('/tmp/g.json')); t=json.load(open('/tmp/t.json'))
print('$v', 'optimizer %.2f us' % (1e3*o['ms_per_step']), '| planner step %.2f us' % (1e3*g['ms_per_step']), '| two-launch step %.2f us, mixture kernel %.2f us' % (1e3*t['ms_per_step'], 1e3*t['roofline'].get('gmm_kernel_avg_ms')))"
done
cp /tmp/lib_keep.so $L
