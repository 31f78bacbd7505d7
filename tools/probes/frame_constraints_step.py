"""Probe: a candidate batch scored against a root trajectory + a hand collision-avoidance constraint, three ways:
 chain   mg_score_trajectory + (mg_back_project_frames_f64 -> mg_joint_positions -> mg_score_frame_constraint)
 tracks  mg_score_trajectory + (mg_joint_tracks -> mg_score_frame_constraints)
 two     mg_joint_tracks (root on the canonical grid, hand on the integer frames) -> mg_score_frame_constraints (both constraints)
usage: python tools/probes/frame_constraints_step.py [B] [steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402
from morphablegraphs_amd import frame_constraints as fc  # noqa: E402
from morphablegraphs_amd.candidate_scoring import cached_trajectory  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = _capi.Context(0)
data = synthetic.make_path_following_primitive()
prim = _capi.Primitive(ctx, data)
joints, animated = synthetic.make_skeleton()
sk = _capi.Skeleton(joints, animated)
F = prim.n_canonical_frames
S = np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32)
frames0 = prim.back_project_frames_f64(S[:1])[0]
root_traj = {"type": "trajectory", "control_points": (frames0[::26, :3] + 0.25).tolist(), "min_u": 0.0, "weight": 1.0, "granularity": 1000}
hand0 = prim.joint_tracks(sk, ["LeftHand"], S[:1])[0, :, 0]
ca = {"type": "frame_ca_position", "joint": "LeftHand", "target": [float(hand0[60, 0]) + 3.0, None, float(hand0[60, 2]) - 2.0], "n_frames": F, "weight": 2.0}
root_as_track = {"type": "frame_joint_trajectory", "joint": "Hips", "control_points": root_traj["control_points"], "min_u": 0.0, "weight": 1.0, "granularity": 1000}
traj = cached_trajectory(prim, root_traj)
d_S, d_err = ctx.upload(S), ctx.malloc(B * 8)


def chain():
    prim.score_trajectory_dev(traj, d_S, np.float32, B, 40, d_err)
    fc.FUSED = False
    fc.add_frame_constraints_dev(prim, S, [ca], sk, None, d_err, accumulate=True)


def tracks():
    prim.score_trajectory_dev(traj, d_S, np.float32, B, 40, d_err)
    fc.FUSED = True
    fc.add_frame_constraints_dev(prim, S, [ca], sk, None, d_err, accumulate=True)


def two():
    fc.FUSED = True
    fc.add_frame_constraints_dev(prim, S, [root_as_track, ca], sk, None, d_err, accumulate=False)


out = {}
for name, fn in (("chain", chain), ("tracks", tracks), ("two", two)):
    for _ in range(5):
        fn()
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.profile_enable(False)
    out[name] = ctx.download(d_err, (B,), np.float64)
    prof = {k: ctx.profile_get(k) for k in ("score_constraints", "joint_tracks", "frame_constraints", "frames")}
    print("%-7s %8.1f us/step  " % (name, dt * 1e6) + "  ".join("%s %.1f us x%d" % (k, 1e3 * ms / max(n, 1), n) for k, (ms, n) in prof.items() if n))
print("chain == tracks bits:", np.array_equal(out["chain"], out["tracks"]), " |chain - two| max:", float(np.abs(out["chain"] - out["two"]).max()))
