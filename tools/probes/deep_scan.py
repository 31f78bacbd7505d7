"""How deep would a placement scan have to look?  Plain hipMalloc candidates of the bench's output size, HELD together, up to N of them
(default 400 = 162 GB of the 288), each probed once: pattern TB/s per candidate, as a histogram over the allocation order -- on a box
whose first 32 candidates are all slow-class, is there fast memory further in?   usage: python tools/probes/deep_scan.py [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = _capi.Context(0)
nbytes = 8192 * 156 * 79 * 4
ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
bufs, rates = [], []
try:
    for i in range(N):
        b = ctx.malloc(nbytes)
        bufs.append(b)
        a = ctx.probe_placement(b)
        rates.append(nbytes / a["pattern_us"] * 1e-6)
        if (i + 1) % 25 == 0:
            chunk = rates[-25:]
            print("candidates %3d .. %3d: pattern TB/s min %.2f  median %.2f  max %.2f   (>= 6.0: %d)" % (
                i - 24, i, min(chunk), sorted(chunk)[12], max(chunk), sum(r >= 6.0 for r in chunk)), flush=True)
except _capi.MGError as e:
    print("stopped at", len(bufs), e)
print("all %d: fast (>= 6.0 TB/s) %d, first fast at %s" % (len(rates), sum(r >= 6.0 for r in rates), next((i for i, r in enumerate(rates) if r >= 6.0), None)))
for b in bufs:
    b.free()
