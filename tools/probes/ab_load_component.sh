cd $GRAFT_REPO_ROOT
L=morphablegraphs_amd/csrc/libmg_hip.so
cp $L /tmp/lib_keep.so
for v in lc0 lc1 lc0 lc1; do
  cp build/lib_$v.so $L
  python bench.py --config optimizer --no-cpu-baseline > /tmp/o.json 2>/dev/null
  python bench.py --two-launch --no-cpu-baseline --single-window --no-extra-configs --no-placement-compare > /tmp/t.json 2>/dev/null
  python bench.py --frames-kernel 1 --no-cpu-baseline --single-window --no-extra-configs --no-placement-compare > /tmp/w.json 2>/dev/null
  python -c "
import json
o=json.load(open('/tmp/o.json')); t=json.load(open('/tmp/t.json')); w=json.load(open('/tmp/w.json'))
print('$v', 'optimizer %.2f us' % (1e3*o['ms_per_step']), '| two-launch step %.2f us, mixture kernel %s' % (1e3*t['ms_per_step'], t['roofline'].get('gmm_kernel_avg_ms')), '| tile-major fused step %.2f us' % (1e3*w['ms_per_step']))"
done
cp /tmp/lib_keep.so $L
