"""What does building a device constraint set cost next to scoring with it? (planner: new targets every step)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
from morphablegraphs_amd.candidate_scoring import cached_constraint_set
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
joints, animated = synthetic.make_skeleton()
sk = _capi.Skeleton(joints, animated)
S = np.random.default_rng(0).standard_normal((1024, 40)).astype(np.float32)
d_S = ctx.upload(S)
prev = np.zeros(79); prev[3::4][:19] = 1.0
def cons(i):
    return [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0 + i, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0 + 0.01 * i]}]
n = 200
for mode in ("fresh set per step", "cached_constraint_set, new targets per step", "same set"):
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(n):
            if mode == "fresh set per step":
                cs = _capi.ConstraintSet(prim, cons(i), sk, alignment=sk.alignment_to(prev + 0.001 * i, 0))
            elif mode.startswith("cached"):
                cs = cached_constraint_set(prim, cons(i), sk, sk.alignment_to(prev + 0.001 * i, 0))
            elif i == 0:
                cs = _capi.ConstraintSet(prim, cons(0), sk, alignment=sk.alignment_to(prev, 0))
            prim.best_candidate_dev(cs, d_S, np.float32, 1024, 40)
            if mode == "fresh set per step":
                cs.close()
        dt = (time.perf_counter() - t0) / n
    print("%-48s %.1f us per step (set + score + argmin + readback)" % (mode, 1e6 * dt))
