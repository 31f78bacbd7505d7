"""Soak of the one-launch planner step's publication protocol (write-through partials, no fences): N steps with different seeds, after
each the record the host polled out of pinned memory is checked against the step's own device buffers -- winner = first minimum of
the errors, latent = that row of the candidates.  A stale or torn partial shows as a mismatch.  usage: options_step_soak.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import synthetic  # noqa: E402
from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = 4096
prims = synthetic.make_graph_primitives(16)
names = [p["name"] for p in prims]
cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
             {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
pset = HipPrimitiveSet(prims, separate_streams=False)
bad = 0
for i in range(steps):
    dev = i < steps // 2 or (i % 2) == 0     # first half: every step draws its counts on the device (the step before drew them ahead)
    best, results = pset.evaluate_options_on_device(names, cons, n, seed=1000 + i, device_counts=dev)
    if i % 10 == 0 or i < 20:       # the check reads 16 x (4096 errors + candidates) back: every tenth step
        for nm in names:
            prim = pset.nodes[nm]._prim
            d_x, d_e, d_r = pset._buffers[(nm, n, np.dtype(np.float32).str)]
            e = prim.ctx.download(d_e, (n,), np.float64)
            x = prim.ctx.download(d_x, (n, prim.n_gmm_dims), np.float32)
            w = int(np.argmin(e))
            lat, err = results[nm][0], results[nm][1]
            if err != e[w] or not np.array_equal(np.asarray(lat, dtype=np.float64), x[w].astype(np.float64)):
                bad += 1
                print("step %d option %s: record (%r) != buffers (row %d, %r)" % (i, nm, err, w, e[w]))
print("options_step_soak: %d steps, %d mismatches" % (steps, bad))
sys.exit(1 if bad else 0)
