"""Score-only mode: no frames are written -- log p(x) + fused keyframe scoring per candidate,
~170 bytes of traffic per candidate, so the bound is arithmetic (float64 MFMA), not HBM.  Reports candidates/s and
the fraction of the float64 matrix peak (78.6 TFLOP/s, AMD's published MI355X figure; the guide has no f64 row)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
ctx = _capi.Context(0)
data = synthetic.make_walk_primitive(seed=0)
prim = _capi.Primitive(ctx, data)
L, K = 40, 8
cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
        {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
cset = _capi.ConstraintSet(prim, cons)
rows = 14
flop = K * (2 * L * L + 2 * L) + 2 * rows * L          # mixture (dense x P_k, as sklearn) + the keyframe channel rows
for B in (8192, 131072):
    S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
    lp, err = ctx.malloc(B * 4), ctx.malloc(B * 8)
    def step():
        prim.gmm_log_prob_dev(S, np.float32, B, L, lp, np.float32)
        prim.score_constraints_dev(cset, S, np.float32, B, L, err, np.float64)
    for _ in range(20): step()
    ctx.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B = %6d: %.1f us per step, %.1f M candidates/s, %.2f TFLOP/s float64 = %.1f %% of 78.6" %
          (B, 1e6 * dt, B / dt / 1e6, B * flop / dt / 1e12, 100 * B * flop / dt / 78.6e12))
