import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
from oracle import c_oracle
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
prim = _capi.Primitive(ctx, data)
cp = c_oracle.COraclePrimitive(data)
for B in (1000, 1000, 255, 17, 8192):
    rng = np.random.default_rng(B)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    model = cp.frames_f32model(S.astype(np.float64)) if B <= 1000 else None
    for rep in range(3):
        got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
        if model is None:
            model = got.copy(); continue
        bad = np.argwhere(got.view(np.uint32) != model.view(np.uint32))
        print("B", B, "rep", rep, "mismatches", len(bad))
        if len(bad):
            print("  cands", np.unique(bad[:, 0])[:20], "frames", np.unique(bad[:, 1])[:40], "chans", np.unique(bad[:, 2])[:20])
            b, f, d = bad[0]; print("  first", bad[0], got[b, f, d], model[b, f, d])
