"""Does the step time depend on WHERE the 404 MB output lives?  One process, several output buffers, each timed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
bufs = [ctx.malloc(B * 156 * 79 * 4) for k in range(8)]
def run(buf, n=1500):
    for _ in range(100): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
for rep in range(2):
    print(" ".join("buf%d@%x: %.1f" % (k, b.ptr.value & 0xffffffffff, run(b)) for k, b in enumerate(bufs)), flush=True)
