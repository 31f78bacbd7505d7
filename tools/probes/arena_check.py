"""What the scan saw against what a later probe of the same buffer sees."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi
ctx = _capi.Context(0)
nbytes = 8192 * 156 * 79 * 4
bufs = []
for i in range(4):
    b = ctx.malloc_placed(nbytes)
    bufs.append(b)
    print(i, "scan:", b.placement, "| later probe:", ctx.probe_placement(b), "|", ctx.probe_placement(b), flush=True)
print(ctx.output_bytes())
