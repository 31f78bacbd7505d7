"""mg_score_constraints at large batches: the tile kernel (a wave per 16 candidates) against the wide one (a wave per 64), per
constraint mix, by the dispatch's own events."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
pos = {"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]}
dire = {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}
for name, cons in (("position", [pos]), ("direction", [dire]), ("position + direction", [pos, dire]), ("4 x position", [pos] * 4)):
    cset = _capi.ConstraintSet(prim, cons)
    for B in (8192, 32768, 65536, 131072):
        S = ctx.upload(np.random.default_rng(0).standard_normal((B, 40)).astype(np.float32))
        err = ctx.malloc(B * 8)
        row = []
        for mode in (1, 2):
            ctx.set_option(_capi.MG_OPT_SCORE_KERNEL, mode)
            for _ in range(50):
                prim.score_constraints_dev(cset, S, np.float32, B, 40, err, np.float64)
            ctx.synchronize()
            ctx.profile_reset(); ctx.profile_enable(1)
            for _ in range(200):
                prim.score_constraints_dev(cset, S, np.float32, B, 40, err, np.float64)
            ctx.synchronize()
            ctx.profile_enable(False)
            ms, n = ctx.profile_get("score_constraints")
            row.append(1e3 * ms / n)
        print("%-22s B = %6d: tile %.1f us   wide %.1f us" % (name, B, row[0], row[1]), flush=True)
        S.free(); err.free()
    cset.close()
