"""Probe: the time-warp batch -- mg_time_function_sample (the canonical time function's inversion per candidate)
and mg_back_project_frames_at (every candidate at its own times) for a walk-sized primitive with a time model (40 spatial + 3 time
latents, 156 frames), B = 16 (a walk's steps) and B = 4096: us per launch on resident buffers.  usage: python tools/probes/timewarp_bench.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402

ctx = _capi.Context(0)
data = synthetic.make_primitive(seed=3, n_components=40, n_frames=156, n_dim=79, n_gmm=8, name="walk_t", n_time_components=3, n_basis_time=8)
prim = _capi.Primitive(ctx, data)
F, D, L, Lt = 156, 79, 40, 3


def timed(fn, reps):
    for _ in range(5):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


for B in (16, 256, 4096):
    rng = np.random.default_rng(B)
    S = np.concatenate([rng.standard_normal((B, L)), 0.3 * rng.standard_normal((B, Lt))], axis=1)
    times, lens = prim.time_function_sample(S[:, L:], 1.0)
    cap = times.shape[1]
    d_g, d_s = ctx.upload(np.ascontiguousarray(S[:, L:])), ctx.upload(np.ascontiguousarray(S[:, :L]))
    d_t, d_l = ctx.upload(np.where(np.isnan(times), 0.0, times)), ctx.upload(lens.astype(np.int32))
    d_t2, d_l2 = ctx.malloc(B * cap * 8), ctx.malloc(B * 4)
    us_tf = timed(lambda: _capi._check(prim.lib.mg_time_function_sample(prim.handle, d_g.ptr, _capi.MG_F64, B, Lt, 1.0, d_t2.ptr, d_l2.ptr, cap, None)), 50)
    row = "B = %5d (rows of %d samples, mean length %.0f): mg_time_function_sample %8.1f us" % (B, cap, lens.mean(), us_tf)
    for odt, name in ((_capi.MG_F64, "float64"), (_capi.MG_F32, "float32")):
        d_o = ctx.malloc(B * cap * D * (8 if odt == _capi.MG_F64 else 4))
        us = timed(lambda: _capi._check(prim.lib.mg_back_project_frames_at(prim.handle, d_s.ptr, _capi.MG_F64, B, L, d_t.ptr, d_l.ptr, cap, d_o.ptr, odt)), 20)
        out_bytes = float(lens.sum()) * D * (8 if odt == _capi.MG_F64 else 4)
        row += " | mg_back_project_frames_at -> %s %8.1f us (%.2f TB/s of frames written)" % (name, us, out_bytes / (us * 1e-6) / 1e12)
        d_o.free()
    print(row, flush=True)
    for b in (d_g, d_s, d_t, d_l, d_t2, d_l2):
        b.free()
