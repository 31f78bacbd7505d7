"""Frames kernels (1 = tile-major, 2 = chunk-stationary) with 0 / 4 / 16 / 28 CUs left free (another units-per-workgroup
count, another unit -> workgroup map) on one fast-class and one slow-class buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
B, L = 8192, 40
NB = B * 156 * 79 * 4
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
bufs = [ctx.malloc(NB) for _ in range(12)]
cls = [ctx.probe_placement(b) for b in bufs]
fast = min(range(12), key=lambda i: cls[i]["ratio"]); slow = max(range(12), key=lambda i: cls[i]["ratio"])
print("fast", cls[fast], "slow", cls[slow], flush=True)
def run(prim, buf, n=300):
    for _ in range(40): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prim.step_frames_and_logp_dev(S, np.float32, B, L, buf, lp)
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
prim = _capi.Primitive(ctx, data)
run(prim, bufs[fast], 1500)
for kern in (1, 2):
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, kern)
    for res in (0, 4, 8, 16, 28, 32, 64):
        ctx.set_reserved_cus(res)
        try:
            plan = prim.step_plan(B)
            print("kernel %d, %2d CUs free, plan %s: fast %.1f us, slow %.1f us" % (kern, res, plan, run(prim, bufs[fast]), run(prim, bufs[slow])), flush=True)
        except _capi.MGError as e:
            print("kernel %d, %d CUs free: %s" % (kern, res, e), flush=True)
