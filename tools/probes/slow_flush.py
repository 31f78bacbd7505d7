"""On a slow-class output buffer: does what separates launches matter?  Step time (wall) of the fused step with plain back-to-back
launches, with timing events attached to every launch (each carries a release to system scope), and the kernel's own duration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic
B, L, F, D = 8192, 40, 156, 79
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
lp = ctx.malloc(B * 4)
ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
bufs = {}
held = []
for _ in range(40):
    b = ctx.malloc(B * F * D * 4)
    info = ctx.probe_placement(b)
    k = "slow" if info["ratio"] > 1.15 else "fast"
    if k not in bufs:
        bufs[k] = (b, info)
    else:
        held.append(b)
    if len(bufs) == 2:
        break
for b in held:
    b.free()
for which in (2, 1):
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, which)
    for k, (out, info) in sorted(bufs.items()):
        def run(n):
            for _ in range(n):
                prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)
            ctx.synchronize()
        run(300)
        t0 = time.perf_counter(); run(1000); plain = (time.perf_counter() - t0) / 1000 * 1e6
        ctx.profile_reset(); ctx.profile_enable(1)
        run(100)
        t0 = time.perf_counter(); run(1000); evt = (time.perf_counter() - t0) / 1000 * 1e6
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("frames")
        print("kernel %d, %s buffer (pattern %.1f us): back to back %.1f us per step; events on every launch %.1f us per step, kernel itself %.1f us"
              % (which, k, info["pattern_us"], plain, evt, 1e3 * ms / n), flush=True)
