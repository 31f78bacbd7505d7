import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
from oracle import c_oracle
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
prim = _capi.Primitive(ctx, data)
cp = c_oracle.COraclePrimitive(data)
for B in (8, 16, 255, 1000):
    rng = np.random.default_rng(1000 + B)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    model = cp.frames_f32model(S.astype(np.float64))
    d_S = ctx.upload(S); d_f = ctx.malloc(B * 156 * 79 * 4); d_l = ctx.malloc(B * 4)
    for rep in range(2):
        prim.step_frames_and_logp_dev(d_S, np.float32, B, 40, d_f, d_l)
        ctx.synchronize()
        fr = ctx.download(d_f, (B, 156, 79), np.float32)
        bad = fr.view(np.uint32) != model.view(np.uint32)
        print("B", B, "rep", rep, "bad", int(bad.sum()))
        if bad.any():
            b, f, d = np.nonzero(bad)
            print("  cands", np.unique(b)[:20], "frames", np.unique(f), "chans", np.unique(d)[:40])
            print("  sample got/model", fr[b[0], f[0], d[0]], model[b[0], f[0], d[0]])
