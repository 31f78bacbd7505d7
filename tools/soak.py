#!/usr/bin/env python3
"""Soak of the fused step on the library's choice of frames kernel: many back-to-back launches per batch size (ragged ones
included), the last result of each compared bit for bit with the tile-major kernel's.   python3 tools/soak.py [launches]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
F, D, L = 156, 79, 40
rng = np.random.default_rng(5)
for B in (8192, 8197, 4099, 12000, 2048, 2049, 16384):
    S = rng.standard_normal((B, L)).astype(np.float32)
    d_S, d_f, d_l = ctx.upload(S), ctx.malloc(B * F * D * 4), ctx.malloc(B * 4)
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 1)
    prim.step_frames_and_logp_dev(d_S, np.float32, B, L, d_f, d_l)
    ref = (ctx.download(d_f, (B * F * D,), np.float32).view(np.uint32), ctx.download(d_l, (B,), np.float32).view(np.uint32))
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 0)
    t0 = time.perf_counter()
    for i in range(n):
        prim.step_frames_and_logp_dev(d_S, np.float32, B, L, d_f, d_l)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    got = (ctx.download(d_f, (B * F * D,), np.float32).view(np.uint32), ctx.download(d_l, (B,), np.float32).view(np.uint32))
    ok = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    print("B = %5d  %s  %d launches, %.2f us each  %s" % (B, prim.step_plan(B)["kernel"], n, 1e6 * dt / n, "identical" if ok else "DIFFER"), flush=True)
    for b in (d_S, d_f, d_l):
        b.free()
    if not ok:
        sys.exit(1)
