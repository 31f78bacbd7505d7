import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from morphablegraphs_amd import synthetic
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
             {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
pset = HipPrimitiveSet(prims, separate_streams=False)
for i in range(50): pset.evaluate_options_on_device(names, cons, 4096, seed=i)
t0=time.perf_counter()
for i in range(500): pset.evaluate_options_on_device(names, cons, 4096, seed=i)
print("ms/step", (time.perf_counter()-t0)/500*1e3)
ctx = pset.ctx
ctx.profile_reset(); ctx.profile_enable(1)
for i in range(200): pset.evaluate_options_on_device(names, cons, 4096, seed=i)
ctx.profile_enable(0)
ms, cnt = ctx.profile_get("options_step")
print("fused kernel: %.1f us average over %d launches" % (1e3 * ms / max(cnt, 1), cnt))
pr = cProfile.Profile(); pr.enable()
for i in range(300): pset.evaluate_options_on_device(names, cons, 4096, seed=i)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
