"""Probe: mg_score_trajectory at batch sizes from 2048 to 131072 candidates with eight lanes per candidate (MG_OPT_TRAJECTORY_LANES 8)
and with one (the streaming kernel, 1), and sixteen scorers of 4096 side by side (mg_score_trajectories) both ways: where does the
eight-lane walk stop paying?  usage: python tools/probes/trajectory_lanes.py"""
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


for B in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 131072):
    S = ctx.upload(np.random.default_rng(1).standard_normal((B, 40)).astype(np.float32))
    e = ctx.malloc(B * 8)
    out = {}
    for lanes in (8, 4, 1):
        if lanes != 1 and B > 65536:
            continue
        ctx.set_option(_capi.MG_OPT_TRAJECTORY_LANES, lanes)
        out[lanes] = (timed(lambda: prim.score_trajectory_dev(traj, S, np.float32, B, 40, e), 20), ctx.download(e, (B,), np.float64))
    same = all(np.array_equal(out[k][1], out[1][1]) for k in out)
    print("B = %6d: eight lanes %s us, four lanes %s us, one lane (streaming) %8.1f us, same bits %s" % (
        B, "%8.1f" % out[8][0] if 8 in out else "       -", "%8.1f" % out[4][0] if 4 in out else "       -", out[1][0], same), flush=True)
    S.free(); e.free()
n, B = 16, 4096
xs = [ctx.upload(np.random.default_rng(2 + i).standard_normal((B, 40)).astype(np.float32)) for i in range(n)]
es = [ctx.malloc(B * 8) for _ in range(n)]
for lanes in (8, 4, 1, 0):
    ctx.set_option(_capi.MG_OPT_TRAJECTORY_LANES, lanes)
    t = timed(lambda: _capi.Primitive.score_trajectories_dev([prim] * n, [traj] * n, xs, np.float32, B, [40] * n, es, [0.0] * n, [1.0] * n), 20)
    print("16 scorers x 4096 side by side, MG_OPT_TRAJECTORY_LANES %d: %8.1f us" % (lanes, t), flush=True)
ctx.set_option(_capi.MG_OPT_TRAJECTORY_LANES, 0)
