"""Host-visible latency of one graph-walk scoring step (BASELINE.json configs[2]: 4096 candidates per option):
sample on the host -> score on the GPU -> first-minimum index back.  Usage: python3 tools/step_latency.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import HipMotionPrimitive, synthetic
from morphablegraphs_amd.candidate_scoring import evaluate_samples_using_constraints, sample_and_evaluate_on_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mp = HipMotionPrimitive(None)
mp._initialize_from_json(synthetic.make_walk_primitive(seed=0))
np.random.seed(0)
S = mp.sample_low_dimensional_vector(n)
cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
        {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
for _ in range(5):
    evaluate_samples_using_constraints(S, mp, cons, None)
t0 = time.perf_counter()
R = 200
for _ in range(R):
    best, err = evaluate_samples_using_constraints(S, mp, cons, None)
dt = (time.perf_counter() - t0) / R
print("evaluate_samples_using_constraints(%d candidates, 2 constraints): %.1f us per call, min error %.6g" % (n, 1e6 * dt, err))
t0 = time.perf_counter()
for _ in range(20):
    np.random.seed(1)
    mp.sample_low_dimensional_vector(n)
print("host sampling of %d latents (sklearn-compatible stream): %.1f us" % (n, 1e6 * (time.perf_counter() - t0) / 20))
for _ in range(5):
    sample_and_evaluate_on_device(mp, cons, n, 1)
t0 = time.perf_counter()
for i in range(R):
    best, err = sample_and_evaluate_on_device(mp, cons, n, i)
print("sample_and_evaluate_on_device(%d candidates): %.1f us per call (device sampler, scoring, argmin, winner back), min error %.6g"
      % (n, 1e6 * (time.perf_counter() - t0) / R, err))

from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
pcons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]}] for nm, p in zip(names, prims)}
for separate in (False, True):
    pset = HipPrimitiveSet(prims, separate_streams=separate)
    for i in range(3):
        pset.evaluate_options_on_device(names, pcons, n, seed=i)
    t0 = time.perf_counter()
    for i in range(50):
        pset.evaluate_options_on_device(names, pcons, n, seed=i)
    print("graph-walk step, 16 options x %d device-sampled candidates, %s: %.1f us per step"
          % (n, "one stream per option" if separate else "one stream", 1e6 * (time.perf_counter() - t0) / 50))
