#!/usr/bin/env python3
"""What the fused mixture costs the frames kernel: kernel time (the dispatch's own events) of the stand-alone frames kernel and of
the fused step kernel on the same placed buffer, rounds interleaved."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic   # noqa: E402

B, L, F, D = 8192, 40, 156, 79
ctx = _capi.Context(0)
ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 2)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
out = ctx.malloc_placed(B * F * D * 4)
lp = ctx.malloc(B * 4)
print(out.placement)


def run(fused, n):
    for _ in range(n):
        if fused:
            prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)
        else:
            prim.back_project_frames_dev(S, np.float32, B, L, out, path=_capi.MG_PATH_MFMA)
    ctx.synchronize()


for fused in (True, False):
    run(fused, 300)
res = {True: [], False: []}
for r in range(6):
    for fused in (True, False):
        run(fused, 50)
        ctx.profile_reset()
        ctx.profile_enable(1)
        run(fused, 200)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("frames")
        res[fused].append(1e3 * ms / n)
for fused in (True, False):
    print("fused" if fused else "frames only", "median %.2f us  min %.2f  max %.2f" % (np.median(res[fused]), min(res[fused]), max(res[fused])))
