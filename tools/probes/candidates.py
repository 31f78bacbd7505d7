"""What a scan has to choose from: plain hipMalloc candidates of the bench's output size, held together, each probed twice
(pattern us, pattern / fill ratio) -- and what the library's scan then picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi
ctx = _capi.Context(0)
nbytes = 8192 * 156 * 79 * 4
ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
bufs = [ctx.malloc(nbytes) for _ in range(24)]
for i, b in enumerate(bufs):
    a, c = ctx.probe_placement(b), ctx.probe_placement(b)
    print("%2d  pattern %.1f / %.1f us   ratio %.3f / %.3f   %.2f TB/s" % (i, a["pattern_us"], c["pattern_us"], a["ratio"], c["ratio"], nbytes / min(a["pattern_us"], c["pattern_us"]) * 1e-6), flush=True)
for b in bufs:
    b.free()
ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
b = ctx.malloc_placed(nbytes)
print("scan:", b.placement)
