"""Probe ("What's weak" 2): does anything amortise when the batch grows?  The fused step at B = 4096 .. 32768 on output
buffers whose PLACEMENT CLASS is recorded beside every line (round 4's soak log did not say which class its larger buffers were):
kernel time by the dispatch's own events, the same per unit of a workgroup (a unit = 16 candidates x one time chunk; 256
workgroups), and what is left over the units' pace.  usage: python tools/probes/batch_scaling.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402

ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
F, D, L = 156, 79, 40
rows = []
for B in (4096, 8192, 12288, 16384, 24576, 32768):
    S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
    out = ctx.malloc_placed(B * F * D * 4)
    lp = ctx.malloc(B * 4)
    info = ctx.placement_info(out)
    probe = ctx.probe_placement(out)
    plan = prim.step_plan(B, out)
    for _ in range(600):
        prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(1)
    for _ in range(300):
        prim.step_frames_and_logp_dev(S, np.float32, B, L, out, lp)
    ctx.synchronize()
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("frames")
    us = 1e3 * ms / n
    units = (B // 16) * prim.n_chunks / 256.0
    rows.append((B, us, units))
    print("B = %6d  %-22s fast_class %-5s pattern %.2f TB/s fill %.1f us | kernel %7.2f us = %.3f of 8 TB/s | %5.1f units per workgroup, %.2f us per unit overall" % (
        B, plan["kernel"], info["fast"], info["pattern_TBps"], probe["pattern_us"] / probe["ratio"], us, B * 49460 / (us * 1e-6) / 8e12, units, us / units), flush=True)
    for b in (S, out, lp):
        b.free()
    ctx.trim_outputs() if hasattr(ctx, "trim_outputs") else None
(b0, u0, n0), (b1, u1, n1) = rows[1], rows[3]
pace = (u1 - u0) / (n1 - n0)
print("pace between B = %d and B = %d: %.2f us per unit; fixed part at B = %d: %.1f us" % (b0, b1, pace, b0, u0 - pace * n0))
