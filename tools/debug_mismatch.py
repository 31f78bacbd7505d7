"""Diagnostic: repeat small launches and report WHERE a mismatch against the f32-contract oracle lands."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
from oracle import c_oracle
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
cp = c_oracle.COraclePrimitive(data)
total_bad = 0
for B, reps in ((16, 400), (1000, 60), (17, 200)):
    rng = np.random.default_rng(B)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    model = cp.frames_f32model(S.astype(np.float64))
    for path, pname in ((_capi.MG_PATH_MFMA, "mfma"), (_capi.MG_PATH_DIRECT, "direct")):
        nbad = 0
        for rep in range(reps if pname == "mfma" else max(10, reps // 10)):
            prim = _capi.Primitive(ctx, data) if rep % 20 == 0 else prim
            got = prim.back_project_frames(S, path=path)
            bad = np.argwhere(got.view(np.uint32) != model.view(np.uint32))
            if len(bad):
                nbad += 1
                total_bad += 1
                if nbad <= 3:
                    print("B", B, pname, "rep", rep, "mismatches", len(bad), "cands", np.unique(bad[:, 0])[:20],
                          "frames", np.unique(bad[:, 1])[:40], "chans", np.unique(bad[:, 2])[:30])
                    b, f, d = bad[0]
                    print("   first", bad[0], got[b, f, d], model[b, f, d], "nan" if np.isnan(got[b, f, d]) else "")
        print("B", B, pname, "bad launches", nbad)
print("TOTAL BAD", total_bad)
