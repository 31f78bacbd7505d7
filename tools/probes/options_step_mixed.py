"""Probe: a planner step (16 options x 4096 candidates) in which EVERY option carries a root trajectory constraint beside its keyframe
constraint (path following), and one in which every option also carries a collision-avoidance position of the hand: the mixed step
(one launch for sampling + keyframes, then the extras per option) against the general chain option by option.
usage: python tools/probes/options_step_mixed.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic  # noqa: E402
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
search = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # MG_OPT_TRAJECTORY_SEARCH: 0 the reference's search (default), 1 the monotone walk
n = 4096
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
joints, animated = synthetic.make_skeleton()
pset = HipPrimitiveSet(prims, separate_streams=False)
pset.ctx.set_option(_capi.MG_OPT_TRAJECTORY_SEARCH, search)
print("MG_OPT_TRAJECTORY_SEARCH = %d (%s)" % (search, "the reference's search" if search == 0 else "the monotone walk"))
sk = _capi.Skeleton(joints, animated)
traj = {"type": "trajectory", "control_points": [[0.0, 0.0, 0.0], [5.0, 0.0, 2.0], [12.0, 0.0, 3.0], [20.0, 0.0, 3.0]], "min_u": 0.0, "weight": 0.5, "granularity": 1000}
base = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]}] for nm, p in zip(names, prims)}
cases = {"keyframes only": base,
         "+ root trajectory": {nm: base[nm] + [traj] for nm in names}}
same_shape = [nm for nm, p in zip(names, prims) if p["n_dim_spatial"] == 79]
if same_shape:
    cases["+ root trajectory + hand position over all frames (options of the skeleton's shape)"] = {
        nm: base[nm] + [traj] + ([{"type": "frame_ca_position", "joint": "LeftHand", "target": [3.0, None, -2.0], "n_frames": int(p["n_canonical_frames"]), "weight": 2.0}]
                                 if nm in same_shape else []) for nm, p in zip(names, prims)}
for label, cons in cases.items():
    out = {}
    for mode in ("mixed", "option by option"):
        saved = HipPrimitiveSet._mixed_step
        if mode != "mixed":
            HipPrimitiveSet._mixed_step = lambda self, *a, **k: None
        try:
            for i in range(5):
                pset.evaluate_options_on_device(names, cons, n, seed=i, skeleton=sk if "hand" in label else None)
            t0 = time.perf_counter()
            for i in range(steps):
                np.random.seed(i)
                r = pset.evaluate_options_on_device(names, cons, n, seed=i, skeleton=sk if "hand" in label else None)
            out[mode] = ((time.perf_counter() - t0) / steps * 1e6, r)
        finally:
            HipPrimitiveSet._mixed_step = saved
    same = out["mixed"][1][0] == out["option by option"][1][0] and all(
        np.array_equal(out["mixed"][1][1][nm][0], out["option by option"][1][1][nm][0]) and out["mixed"][1][1][nm][1] == out["option by option"][1][1][nm][1] for nm in names)
    print("%-90s mixed %8.1f us/step   option by option %8.1f us/step   same winners: %s" % (label, out["mixed"][0], out["option by option"][0], same))
