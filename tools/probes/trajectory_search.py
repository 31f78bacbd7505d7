"""Probe: mg_score_trajectory / mg_score_trajectories with the reference's search (MG_OPT_TRAJECTORY_SEARCH 0: L-BFGS-B restated, one
lane per candidate) against the monotone walk (1: eight / four / one lanes by batch size): microseconds per launch and how far the
two searches' errors are apart on a smooth path-following batch.  usage: python tools/probes/trajectory_search.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402

ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_path_following_primitive(seed=0))
S0 = np.random.default_rng(0).standard_normal((1, 40)).astype(np.float32)
frames0 = prim.back_project_frames_f64(S0)[0]
traj = _capi.Trajectory(prim, frames0[::26, :3] + 0.25, 1000)


def timed(fn, reps):
    for _ in range(3):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for B in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    S = ctx.upload(np.random.default_rng(1).standard_normal((B, 40)).astype(np.float32))
    e = ctx.malloc(B * 8)
    out = {}
    for search in (0, 1):
        ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, search)
        out[search] = (timed(lambda: prim.score_trajectory_dev(traj, S, np.float32, B, 40, e), 10), ctx.download(e, (B,), np.float64))
    dev = np.abs(out[0][1] - out[1][1])
    print("B = %6d: reference search %8.1f us, monotone walk %8.1f us; errors apart: max %.2e, candidates beyond 1e-6: %d" % (
        B, out[0][0], out[1][0], dev.max(), int((dev > 1e-6 * np.maximum(1.0, out[1][1])).sum())), flush=True)
    S.free(); e.free()
n, B = 16, 4096
xs = [ctx.upload(np.random.default_rng(2 + i).standard_normal((B, 40)).astype(np.float32)) for i in range(n)]
es = [ctx.malloc(B * 8) for _ in range(n)]
for search in (0, 1):
    ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, search)
    t = timed(lambda: _capi.Primitive.score_trajectories_dev([prim] * n, [traj] * n, xs, np.float32, B, [40] * n, es, [0.0] * n, [1.0] * n), 10)
    print("16 scorers x 4096 side by side, MG_OPT_TRAJECTORY_SEARCH %d: %8.1f us" % (search, t), flush=True)
ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, 0)
