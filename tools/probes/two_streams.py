#!/usr/bin/env python3
"""Do independent steps overlap when they are launched from two contexts (two streams) in turn?  Step time per batch with one
context and with two (each with its own primitive, latents, output and log p buffers)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic   # noqa: E402

B, L, F, D = 8192, 40, 156, 79
data = synthetic.make_walk_primitive(seed=0)
lib = _capi.load_library()
sets = []
for i in range(int(os.environ.get("NCTX", "2"))):
    ctx = _capi.Context(0, lib=lib)
    prim = _capi.Primitive(ctx, data)
    S = ctx.upload(np.random.default_rng(i).standard_normal((B, L)).astype(np.float32))
    out = ctx.malloc_placed(B * F * D * 4)
    lp = ctx.malloc(B * 4)
    sets.append((ctx, prim, S, out, lp))
    print("context", i, out.placement, flush=True)


def run(n, k):
    for i in range(n):
        ctx, prim, S, out, lp = sets[i % k]
        prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)
    for s in sets:
        s[0].synchronize()


for k in (1, len(sets), 1, len(sets)):
    run(400, k)
    t0 = time.perf_counter()
    run(2000, k)
    dt = time.perf_counter() - t0
    print("%d context(s): %.2f us per step" % (k, 1e6 * dt / 2000), flush=True)
